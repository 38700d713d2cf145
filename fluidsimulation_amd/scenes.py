"""Seeded synthetic scenes with the shapes of the reference's main.cpp set-up.

The reference builds its scene with OpenVDB (container box ``main.cpp:122-133``, cube fluid source
``main.cpp:83-87``, floating sphere ``main.cpp:60-61``) and spawns particles from a
``std::random_device``-seeded generator (``src/util.cpp:5``, ``src/FluidSource.cpp:52-60``), so no two
runs of it agree.  These generators produce inputs of the same shape from a fixed seed; they are the
input side of every parity test and of the benchmark.

All dense arrays are C-ordered ``[nz, ny, nx]`` (x fastest), i.e. ``id = i + nx*(j + ny*k)``.
Positions are index-space doubles: cell ``(i,j,k)`` of the box is centred on ``(ox+i, oy+j, oz+k)``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

REST_DENSITY = 1000.0  # FluidDomain::density, src/FluidDomain.cpp:12
SOLID_HALF_WIDTH = 5   # createLevelSetBox(..., 5), main.cpp:125
FLUID_HALF_WIDTH = 3   # openvdb::createLevelSet default half width, src/FluidDomain.cpp:18


@dataclass
class Scene:
    n: int                      # nominal resolution (nx = ny = nz of main.cpp:45-47)
    dims: tuple                 # (nx, ny, nz) of the dense box
    origin: tuple               # index-space coordinate of local cell (0,0,0)
    dx: float                   # float32 voxel size
    phi_solid: np.ndarray       # float32 [nz,ny,nx]; > 0 inside the open container
    phi_solid_bg: float
    phi_fluid_bg: float
    pos: np.ndarray = field(default=None)    # float64 [n,3]
    vel: np.ndarray = field(default=None)    # float64 [n,3]
    mass: np.ndarray = field(default=None)   # float32 [n]
    phi_fluid: np.ndarray = field(default=None)  # float32 [nz,ny,nx]
    box_lo: tuple = None        # container walls in index space (continuous coordinates)
    box_hi: tuple = None
    phi_object: np.ndarray = None
    phi_object_bg: float = 0.0
    object_com: np.ndarray = None   # index space
    object_minv: np.ndarray = None
    object_vel: np.ndarray = None

    @property
    def n_cells(self) -> int:
        return int(self.dims[0]) * int(self.dims[1]) * int(self.dims[2])

    def desc(self) -> dict:
        return dict(nx=self.dims[0], ny=self.dims[1], nz=self.dims[2], ox=self.origin[0], oy=self.origin[1],
                    oz=self.origin[2], dx=self.dx, phi_fluid_bg=self.phi_fluid_bg, phi_solid_bg=self.phi_solid_bg,
                    rest_density=REST_DENSITY)

    def cell_coords(self):
        """Index-space coordinates of the cell centres as three broadcastable float64 arrays."""
        nx, ny, nz = self.dims
        x = (self.origin[0] + np.arange(nx, dtype=np.float64))[None, None, :]
        y = (self.origin[1] + np.arange(ny, dtype=np.float64))[None, :, None]
        z = (self.origin[2] + np.arange(nz, dtype=np.float64))[:, None, None]
        return x, y, z


def _box_sdf(x, y, z, lo, hi):
    """Signed distance (index units) to an axis-aligned box, negative inside."""
    cx, cy, cz = [(l + h) * 0.5 for l, h in zip(lo, hi)]
    hx, hy, hz = [(h - l) * 0.5 for l, h in zip(lo, hi)]
    qx, qy, qz = np.abs(x - cx) - hx, np.abs(y - cy) - hy, np.abs(z - cz) - hz
    outside = np.sqrt(np.maximum(qx, 0) ** 2 + np.maximum(qy, 0) ** 2 + np.maximum(qz, 0) ** 2)
    inside = np.minimum(np.maximum(qx, np.maximum(qy, qz)), 0)
    return outside + inside


def container(n: int, full_domain: bool = False, z_repeat: int = 1, z_planes=None) -> Scene:
    """The inverted box level set of main.cpp:122-133: walls at index (n/4,1,n/4)..(3n/4,n-2,3n/4)
    (``full_domain``: (1,1,1)..(n-2,n-2,n-2)), narrow band of 5 voxels, positive inside the open box.
    The dense box is the band's bounding box.  ``z_repeat`` stretches the open box to that many times its
    length along z (the multi-GPU benchmark scene); ``z_planes = (e0, e1)`` builds only those planes of the
    dense box (one rank's slab)."""
    dx = np.float32(1.0) / np.float32(n)   # width / max(nx,ny,nz), main.cpp:106
    if full_domain:
        lo, hi = (1.0, 1.0, 1.0), (n - 2.0, n - 2.0, n - 2.0)
    else:
        lo, hi = (n // 4 * 1.0, 1.0, n // 4 * 1.0), (3 * n // 4 * 1.0, n - 2.0, 3 * n // 4 * 1.0)
    hi = (hi[0], hi[1], lo[2] + z_repeat * (hi[2] - lo[2]))
    hw = SOLID_HALF_WIDTH
    origin = tuple(int(l) - hw for l in lo)
    dims = tuple(int(h) + hw - o + 1 for h, o in zip(hi, origin))
    if z_planes is not None:
        e0, e1 = z_planes
        assert 0 <= e0 < e1 <= dims[2]
        origin = (origin[0], origin[1], origin[2] + e0)
        dims = (dims[0], dims[1], e1 - e0)
    sc = Scene(n=n, dims=dims, origin=origin, dx=float(dx), phi_solid=None, phi_solid_bg=float(-hw * dx),
               phi_fluid_bg=float(FLUID_HALF_WIDTH * dx), box_lo=lo, box_hi=hi)
    x, y, z = sc.cell_coords()
    sdf = _box_sdf(x, y, z, lo, hi)
    phi = np.clip(-sdf, -hw, hw) * float(dx)          # invert(): positive inside the container
    sc.phi_solid = np.ascontiguousarray(phi, dtype=np.float32)
    return sc


def _fill_cells(sc: Scene, cell_lo, cell_hi, ppc: int, rng: np.random.Generator, margin: float = 0.05):
    """Jittered-uniform particles, ``ppc`` per cell of [cell_lo, cell_hi) (index coordinates of the cell
    centres), kept only strictly inside the container (the reference spawns only where the source region
    intersected with the solid is negative, src/FluidSource.cpp:17-18,63)."""
    ii = np.arange(cell_lo[0], cell_hi[0])
    jj = np.arange(cell_lo[1], cell_hi[1])
    kk = np.arange(cell_lo[2], cell_hi[2])
    K, J, I = np.meshgrid(kk, jj, ii, indexing="ij")
    centres = np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1).astype(np.float64)
    centres = np.repeat(centres, ppc, axis=0)
    pos = centres + rng.uniform(-0.5, 0.5, size=centres.shape)
    lo = np.asarray(sc.box_lo) + margin
    hi = np.asarray(sc.box_hi) - margin
    keep = np.all((pos > lo) & (pos < hi), axis=1)
    pos = np.ascontiguousarray(pos[keep])
    n = pos.shape[0]
    vel = np.zeros((n, 3), dtype=np.float64)
    # mass = density * dx^3 / spawn_counter (float arithmetic of src/FluidSource.cpp:68, pow -> double)
    mass = np.full(n, np.float32(REST_DENSITY * float(sc.dx) ** 3 / ppc), dtype=np.float32)
    return pos, vel, mass


def _fill_cells_by_plane(sc: Scene, cell_lo, cell_hi, ppc: int, seed: int, margin: float = 0.05):
    """As _fill_cells, but every z plane of cells draws from its own generator seeded by (seed, plane), and only
    the planes inside the scene's dense box are produced: two ranks that both hold a plane (one as ghosts) get
    identical particles in it, and no rank ever builds the whole set."""
    k_lo = max(cell_lo[2], sc.origin[2])
    k_hi = min(cell_hi[2], sc.origin[2] + sc.dims[2])
    ii = np.arange(cell_lo[0], cell_hi[0])
    jj = np.arange(cell_lo[1], cell_hi[1])
    J, I = np.meshgrid(jj, ii, indexing="ij")
    base = np.stack([I.ravel(), J.ravel()], axis=1).astype(np.float64)
    lo = np.asarray(sc.box_lo) + margin
    hi = np.asarray(sc.box_hi) - margin
    chunks = []
    for k in range(k_lo, k_hi):
        rng = np.random.default_rng([seed, k])
        centres = np.concatenate([base, np.full((base.shape[0], 1), float(k))], axis=1)
        centres = np.repeat(centres, ppc, axis=0)
        pos = centres + rng.uniform(-0.5, 0.5, size=centres.shape)
        chunks.append(pos[np.all((pos > lo) & (pos < hi), axis=1)])
    pos = np.ascontiguousarray(np.concatenate(chunks)) if chunks else np.zeros((0, 3))
    n = pos.shape[0]
    vel = np.zeros((n, 3), dtype=np.float64)
    mass = np.full(n, np.float32(REST_DENSITY * float(sc.dx) ** 3 / ppc), dtype=np.float32)
    return pos, vel, mass


def fluid_levelset(sc: Scene, pos: np.ndarray) -> np.ndarray:
    """Union of balls of radius 1.01*dx*sqrt(3)/2 (src/FluidSource.cpp:67) around the particles, clamped to
    the 3-voxel narrow band, then merged with the solid as IntersectSolidFluidLevelSets::intersect does
    (headers/LevelSet.h:83-96).  Stands in for OpenVDB's ParticlesToLevelSet (src/LevelSet.cpp:17-23),
    which is not available; it is an *input* of the hot path, not part of it."""
    from scipy.spatial import cKDTree

    nx, ny, nz = sc.dims
    radius = 1.01 * np.sqrt(3.0) / 2.0       # index units
    band = float(FLUID_HALF_WIDTH)
    phi = np.full((nz, ny, nx), band, dtype=np.float64)
    if pos.shape[0] > 0:
        x, y, z = sc.cell_coords()
        pts = np.stack(np.broadcast_arrays(x, y, z), axis=-1).reshape(-1, 3)
        tree = cKDTree(pos)
        dist, _ = tree.query(pts, k=1, distance_upper_bound=band + radius + 1.0, workers=-1)
        d = np.where(np.isfinite(dist), dist - radius, band)
        phi = np.clip(d, -band, band).reshape(nz, ny, nx)
    phi = phi * float(sc.dx)
    phi = phi.astype(np.float32)
    ps = sc.phi_solid
    merged = np.where(ps <= 0, np.maximum(phi, -ps), phi)
    return np.ascontiguousarray(merged, dtype=np.float32)


def fluid_levelset_box(sc: Scene, cell_lo, cell_hi) -> np.ndarray:
    """Cheap stand-in for large benchmark scenes: signed distance to the filled block of cells grown by
    0.4 voxel (roughly where the union of jittered balls ends), narrow band 3, merged with the solid."""
    x, y, z = sc.cell_coords()
    lo = [max(float(c) - 0.5, b) for c, b in zip(cell_lo, sc.box_lo)]
    hi = [min(float(c) - 0.5, b) for c, b in zip(cell_hi, sc.box_hi)]
    sdf = _box_sdf(x, y, z, lo, hi) - 0.4
    band = float(FLUID_HALF_WIDTH)
    phi = (np.clip(sdf, -band, band) * float(sc.dx)).astype(np.float32)
    ps = sc.phi_solid
    return np.ascontiguousarray(np.where(ps <= 0, np.maximum(phi, -ps), phi), dtype=np.float32)


def cube_drop(n: int, ppc: int = 8, seed: int = 12345, exact_levelset: bool = True) -> Scene:
    """Configs 1-2: the cube source of main.cpp:83-87 (scale 0.5, centred (0.5,0.25,0.5)) filled with
    ``ppc`` jittered particles per cell, velocity 0."""
    sc = container(n)
    rng = np.random.default_rng(seed)
    lo = (n // 4, 1, n // 4)
    hi = (3 * n // 4 + 1, n // 2, 3 * n // 4 + 1)
    sc.pos, sc.vel, sc.mass = _fill_cells(sc, lo, hi, ppc, rng)
    sc.phi_fluid = fluid_levelset(sc, sc.pos) if exact_levelset else fluid_levelset_box(sc, lo, hi)
    return sc


def dam_break(n: int, ppc: int = 8, seed: int = 12345, exact_levelset: bool = True, full_domain: bool = False) -> Scene:
    """Configs 4-5: a column against the -x wall, [n/4, n/2) x [1, 3n/4) x [n/4, 3n/4]."""
    sc = container(n, full_domain=full_domain)
    rng = np.random.default_rng(seed)
    if full_domain:
        lo, hi = (1, 1, 1), (n // 2, 3 * n // 4, n - 1)
    else:
        lo, hi = (n // 4, 1, n // 4), (n // 2, 3 * n // 4, 3 * n // 4 + 1)
    sc.pos, sc.vel, sc.mass = _fill_cells(sc, lo, hi, ppc, rng)
    sc.phi_fluid = fluid_levelset(sc, sc.pos) if exact_levelset else fluid_levelset_box(sc, lo, hi)
    return sc


def dam_break_slab(n: int, ppc: int = 8, seed: int = 12345, z_repeat: int = 1, z_planes=None, full_domain: bool = False) -> Scene:
    """The dam-break column in a container ``z_repeat`` times as long along z, restricted to the planes
    ``z_planes = (e0, e1)`` of its dense box (None: all of it).  Particles come plane by plane from
    (seed, plane)-seeded generators and the level set is the analytic stand-in, so a rank can build its slab
    alone and neighbouring ranks agree on the planes they share.  ``full_domain``: the container spans the whole
    nominal grid (walls at 1 and n-2 on every axis) and the column fills its lower-x half to 3/4 of the height:
    at n = 512 and 16 particles per cell that is the ~1e9-particle case."""
    sc = container(n, full_domain=full_domain, z_repeat=z_repeat, z_planes=z_planes)
    if full_domain:
        lo = (1, 1, 1)
        hi = (n // 2, 3 * n // 4, 1 + z_repeat * (n - 3) + 1)
    else:
        lo = (n // 4, 1, n // 4)
        hi = (n // 2, 3 * n // 4, n // 4 + z_repeat * (n // 2) + 1)
    sc.pos, sc.vel, sc.mass = _fill_cells_by_plane(sc, lo, hi, ppc, seed)
    sc.phi_fluid = fluid_levelset_box(sc, lo, hi)
    return sc


def slab_box_planes(n: int, z_repeat: int = 1, full_domain: bool = False) -> int:
    """Number of z planes of the dense box of ``dam_break_slab(n, z_repeat=z_repeat, full_domain=full_domain)``."""
    open_planes = (n - 3) if full_domain else (n // 2)
    return z_repeat * open_planes + 2 * SOLID_HALF_WIDTH + 1


def add_sphere(sc: Scene, radius_world: float = 0.1, centre_world=(0.5, 0.5, 0.5), density: float = 250.0,
               vel=(0.0, 0.0, 0.0, 0.0, 0.0, 0.0)) -> Scene:
    """The rigid sphere of main.cpp:60-61 / src/SolidObject.cpp:12-40: narrow-band (5 voxels) sphere SDF in
    world units, centre of mass in index space, inverse mass matrix with the reference's integer 4/3."""
    dx = float(sc.dx)
    hw = float(SOLID_HALF_WIDTH)
    # world = dx * (index + 0.5)  (main.cpp:107-108)  ->  index = world/dx - 0.5
    x, y, z = sc.cell_coords()
    wx, wy, wz = (x + 0.5) * dx, (y + 0.5) * dx, (z + 0.5) * dx
    d = np.sqrt((wx - centre_world[0]) ** 2 + (wy - centre_world[1]) ** 2 + (wz - centre_world[2]) ** 2) - radius_world
    sc.phi_object = np.ascontiguousarray(np.clip(d, -hw * dx, hw * dx), dtype=np.float32)
    sc.phi_object_bg = float(hw * dx)
    sc.object_com = np.asarray(centre_world, dtype=np.float64) / dx      # SolidObject.cpp:14 (center / voxel_size)
    r32 = np.float32(radius_world)
    mass = np.float32(1 * 3.14 * float(r32) ** 3 * density)              # 4/3 == 1 in integer arithmetic (:24)
    inertia = np.float64(np.float32(2) * mass * (r32 * r32) / np.float32(5))
    sc.object_minv = np.array([1.0 / float(mass)] * 3 + [1.0 / float(inertia)] * 3, dtype=np.float64)
    sc.object_vel = np.asarray(vel, dtype=np.float64)
    return sc
