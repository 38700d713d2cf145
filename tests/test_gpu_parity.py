"""GPU parity tests: every stage of the hot path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bars: bit-exact for integer/byte work (binning, bands, labels, flags) and for the float/double
stages whose operation order is fixed (P2G, face fractions, density, assembly, G2P, projections); solver
outputs to the tolerances the task states: post-projection velocity <= 1e-4 relative L-inf, residual at or
below the reference tolerance (1e-15 relative 2-norm), positions <= 1e-5 cell."""
import numpy as np
import pytest

from conftest import desc_from_golden, fresh_state
from fluidsimulation_amd import FlipContext, SOLVE_DENSITY, SOLVE_DIVERGENCE, scenes
from oracle import Oracle

pytestmark = pytest.mark.gpu

VEL_RTOL = 1e-4      # post-projection velocity, relative L-inf
POS_ATOL = 1e-5      # particle positions, cells per step
RES_TOL = 1e-15      # reference CG tolerance (src/FluidSimulator.cpp:660)


def rel_linf(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.fixture(scope="module")
def case():
    """16^3 cube drop with random particle velocities, first substep."""
    sc = scenes.cube_drop(16, ppc=6, seed=5)
    rng = np.random.default_rng(17)
    sc.vel = rng.normal(0.0, 0.25, sc.pos.shape)
    o = Oracle(sc.desc())
    ref = o.substep(fresh_state(sc, o), 0.004)
    ctx = FlipContext(sc.desc())
    ctx.upload_scene(sc)
    yield sc, o, ref, ctx
    ctx.close()


def test_binning_bit_exact(case):
    sc, o, ref, ctx = case
    n_out = ctx.bin_particles()
    cell_id, cell_start, perm = ctx.download_binning()
    assert n_out == 0
    assert np.array_equal(cell_id, ref["cell_id"])
    assert np.array_equal(cell_start, ref["cell_start"])
    assert np.array_equal(perm, ref["perm"])
    band = ctx.download("band")
    assert np.array_equal(band & 1, ref["pmask1"]) and np.array_equal((band >> 1) & 1, ref["band2"])


def test_p2g_bit_exact(case):
    sc, o, ref, ctx = case
    ctx.bin_particles()
    ctx.p2g()
    for name, want in zip(("vel_u", "vel_v", "vel_w"), ref["p2g"]):
        assert np.array_equal(ctx.download(name), want), name
    assert np.array_equal(ctx.download("active"), ref["band2"])


def test_full_first_substep_stage_by_stage(case):
    sc, o, ref, ctx = case
    dt = 0.004
    ctx.upload_scene(sc)
    ctx.bin_particles()
    ctx.p2g()
    ctx.mark_valid_from_active()
    ctx.extrapolate(2)
    for name, want in zip(("vel_u", "vel_v", "vel_w"), ref["extrap1"]):
        assert np.array_equal(ctx.download(name), want), name
    ctx.update_prev()
    ctx.add_forces(dt)
    n_rows = ctx.index_cells()
    assert n_rows == ref["n_rows"]
    assert np.array_equal(ctx.download("label"), ref["label"])
    assert np.array_equal(ctx.download("row_index"), ref["index"])
    ctx.face_fractions()
    for name, want in zip(("weight_u", "weight_v", "weight_w"), ref["w"]):
        assert np.array_equal(ctx.download(name), want), name
    for name, want in zip(("fluid_weight_u", "fluid_weight_v", "fluid_weight_w"), ref["fw"]):
        assert np.array_equal(ctx.download(name), want), name
    n_over = ctx.density()
    assert np.array_equal(ctx.download("density"), ref["density"]) and n_over == ref["n_over"]
    # the value above came out of the P2G walk; a level-set upload makes it stale and the stand-alone
    # density kernel must give the same field
    ctx.upload("phi_fluid", sc.phi_fluid)
    assert ctx.density() == ref["n_over"]
    assert np.array_equal(ctx.download("density"), ref["density"])
    # divergence system
    ctx.build_system(SOLVE_DIVERGENCE, dt)
    off, diag, rhs = ctx.download_system()
    assert np.array_equal(diag, ref["sys_div"][1])
    assert np.array_equal(rhs, ref["sys_div"][2])
    assert np.array_equal(off, ref["sys_div"][0])
    info = ctx.pcg_solve(RES_TOL)
    assert info["converged"] == 1 and info["rows"] == n_rows
    assert info["rel_residual_2"] < RES_TOL
    p = ctx.download("pressure")
    lab = ref["label"].astype(bool)
    assert not p[~lab].any()
    assert rel_linf(p[lab], ref["p_div"]) < 1e-9
    ctx.project_velocity(dt)
    for name, want in zip(("vel_u", "vel_v", "vel_w"), ref["projected"]):
        assert rel_linf(ctx.download(name), want) < VEL_RTOL, name
    # density system
    ctx.build_system(SOLVE_DENSITY, dt)
    off, diag, rhs = ctx.download_system()
    assert np.array_equal(diag, ref["sys_den"][1]) and np.array_equal(rhs, ref["sys_den"][2])
    assert np.array_equal(off, ref["sys_den"][0])
    info = ctx.pcg_solve(RES_TOL)
    assert info["converged"] == 1 and info["rel_residual_2"] < RES_TOL
    assert rel_linf(ctx.download("pressure")[lab], ref["p_den"]) < 1e-9
    ctx.project_displacement(dt)
    for name, want in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
        assert np.abs(ctx.download(name) - want).max() < POS_ATOL, name
    ctx.extrapolate(2)
    ctx.update_diff()
    ctx.g2p(0.95)
    ctx.apply_displacements()
    pos, vel = ctx.download_particles()
    assert np.abs(pos - ref["new_pos"]).max() < POS_ATOL
    assert rel_linf(vel, ref["new_vel"]) < VEL_RTOL
    for name, want in zip(("valid_u", "valid_v", "valid_w"), ref["valid"]):
        assert np.array_equal(ctx.download(name), want), name


def test_transfers_bit_exact_on_given_grids(case):
    """G2P and displacement sampling with oracle-provided grids: no solver in between, so bit for bit."""
    sc, o, ref, ctx = case
    ctx.upload_scene(sc)
    ctx.bin_particles()
    for name, arr in zip(("vel_u", "vel_v", "vel_w"), ref["final_vel"]):
        ctx.upload(name, arr)
    for name, arr in zip(("diff_u", "diff_v", "diff_w"), ref["diff"]):
        ctx.upload(name, arr)
    for name, arr in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
        ctx.upload(name, arr)
    ctx.g2p(0.95)
    ctx.apply_displacements()
    pos, vel = ctx.download_particles()
    assert np.array_equal(vel, ref["new_vel"])
    assert np.array_equal(pos, ref["new_pos"])


def test_projection_bit_exact_on_given_pressure(case):
    sc, o, ref, ctx = case
    dt = 0.004
    ctx.upload_scene(sc)
    ctx.bin_particles()
    ctx.index_cells()
    for name, arr in zip(("weight_u", "weight_v", "weight_w"), ref["w"]):
        ctx.upload(name, arr)
    lab = ref["label"].astype(bool)
    P = np.zeros(o.shape)
    P[lab] = ref["p_div"]
    ctx.upload("pressure", P)
    pre = ref["pre_project"]     # extrapolated velocity + gravity, as the oracle projected it
    want, valid = o.project_velocity(dt, ref["index"], ref["p_div"], sc.phi_fluid, ref["w"], pre)
    for name, arr in zip(("vel_u", "vel_v", "vel_w"), pre):
        ctx.upload(name, arr)
    ctx.project_velocity(dt)
    for name, w_ in zip(("vel_u", "vel_v", "vel_w"), want):
        assert np.array_equal(ctx.download(name), w_), name
    for name, v_ in zip(("valid_u", "valid_v", "valid_w"), valid):
        assert np.array_equal(ctx.download(name), v_), name
    P2 = np.zeros(o.shape)
    P2[lab] = ref["p_den"]
    ctx.upload("pressure", P2)
    ctx.project_displacement(dt)
    for name, w_ in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
        assert np.array_equal(ctx.download(name), w_), name


def test_operator_kernel_matches_csr_product(case):
    sc, o, ref, ctx = case
    dt = 0.004
    ctx.upload_scene(sc)
    ctx.bin_particles(); ctx.p2g(); ctx.index_cells(); ctx.face_fractions(); ctx.density()
    ctx.build_system(SOLVE_DENSITY, dt)
    off, diag, rhs = ctx.download_system()
    idx = ctx.download("row_index")
    rowptr, col, val, b = o.to_csr(idx, off, diag, rhs)
    rng = np.random.default_rng(3)
    xc = rng.normal(size=rowptr.size - 1)
    X = np.zeros(o.shape)
    X[idx >= 0] = xc[idx[idx >= 0]]
    Y = ctx.apply_operator(X)
    want = o.apply_operator(rowptr, col, val, xc)
    got = Y[idx >= 0]
    assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    assert not Y[idx < 0].any()


def test_against_stored_golden_stages(golden_stage):
    g = golden_stage
    desc = desc_from_golden(g)
    dt = float(g["dt"])
    with FlipContext(desc) as ctx:
        ctx.upload("phi_solid", g["phi_solid"]); ctx.upload("phi_fluid", g["phi_fluid"])
        ctx.upload_particles(g["pos"], g["vel"], g["mass"])
        ctx.bin_particles()
        cell_id, cell_start, perm = ctx.download_binning()
        assert np.array_equal(cell_id, g["cell_id"]) and np.array_equal(cell_start, g["cell_start"])
        assert np.array_equal(perm, g["perm"])
        ctx.p2g()
        got = np.stack([ctx.download(n) for n in ("vel_u", "vel_v", "vel_w")])
        assert np.array_equal(got, g["p2g"])
        ctx.mark_valid_from_active(); ctx.extrapolate(2); ctx.update_prev(); ctx.add_forces(dt)
        assert ctx.index_cells() == int(g["n_rows"])
        assert np.array_equal(ctx.download("label"), g["label"])
        ctx.face_fractions()
        assert np.array_equal(np.stack([ctx.download(n) for n in ("weight_u", "weight_v", "weight_w")]), g["w"])
        ctx.density()
        assert np.array_equal(ctx.download("density"), g["density"])
        ctx.build_system(SOLVE_DIVERGENCE, dt)
        off, diag, rhs = ctx.download_system()
        assert np.array_equal(diag, g["sys_div_diag"]) and np.array_equal(rhs, g["sys_div_rhs"])
        assert np.array_equal(off, g["sys_div_off"])
        assert ctx.pcg_solve(RES_TOL)["converged"] == 1
        ctx.project_velocity(dt)
        got = np.stack([ctx.download(n) for n in ("vel_u", "vel_v", "vel_w")])
        assert rel_linf(got, g["projected"]) < VEL_RTOL
        ctx.build_system(SOLVE_DENSITY, dt)
        assert ctx.pcg_solve(RES_TOL)["converged"] == 1
        ctx.project_displacement(dt)
        ctx.extrapolate(2); ctx.update_diff(); ctx.g2p(0.95); ctx.apply_displacements()
        pos, vel = ctx.download_particles()
        assert np.abs(pos - g["new_pos"]).max() < POS_ATOL and rel_linf(vel, g["new_vel"]) < VEL_RTOL


def test_second_substep_with_stale_weights_and_wall_rows():
    """Stale weights give rows to cells cut by the container (non-symmetric system, all-zero rows on the
    container's edges): labels bit-exact, both solves to the reference tolerance, same velocities."""
    sc = scenes.cube_drop(16, ppc=4, seed=21)
    o = Oracle(sc.desc())
    st = fresh_state(sc, o)
    out1 = o.substep(st, 0.01)
    phi2 = scenes.fluid_levelset(sc, out1["new_pos"])
    st2 = dict(st, w=out1["w"], valid=out1["valid"], pos=out1["new_pos"], vel=out1["new_vel"], phi_fluid=phi2)
    out2 = o.substep(st2, 0.01, max_iter=20000, density_projection=False)
    assert out2["err_div"] < RES_TOL
    from fluidsimulation_amd import FlipSubstep
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        step = FlipSubstep(ctx, density_projection=True)
        step.step(0.01)
        pos1, vel1 = ctx.download_particles()
        assert np.abs(pos1 - out1["new_pos"]).max() < POS_ATOL and rel_linf(vel1, out1["new_vel"]) < VEL_RTOL
        # take the oracle's particles and level set so that the second step starts from identical inputs
        ctx.upload_particles(out1["new_pos"], out1["new_vel"], sc.mass)
        ctx.upload("phi_fluid", phi2)
        for name, arr in zip(("weight_u", "weight_v", "weight_w"), out1["w"]):
            ctx.upload(name, arr)
        for name, arr in zip(("valid_u", "valid_v", "valid_w"), out1["valid"]):
            ctx.upload(name, arr)
        step2 = FlipSubstep(ctx, density_projection=False, max_iter=20000)
        info = step2.step(0.01)
        assert np.array_equal(ctx.download("label"), out2["label"])
        assert info["divergence"]["converged"] == 1 and info["divergence"]["rel_residual_2"] < RES_TOL
        lab = out2["label"].astype(bool)
        assert rel_linf(ctx.download("pressure")[lab], out2["p_div"]) < 1e-7
        _, vel2 = ctx.download_particles()
        assert rel_linf(vel2, out2["new_vel"]) < VEL_RTOL


def test_as_run_iteration_and_elimination_agree_on_the_gpu():
    """The same second-substep system solved both ways by the library (options of flip_set_system_options):
    CG over the whole non-symmetric matrix, as the reference runs it, and CG on the liquid block followed by
    substitution.  Same pressure, the oracle's in both cases."""
    sc = scenes.cube_drop(16, ppc=4, seed=21)
    o = Oracle(sc.desc())
    st = fresh_state(sc, o)
    out1 = o.substep(st, 0.01)
    phi2 = scenes.fluid_levelset(sc, out1["new_pos"])
    st2 = dict(st, w=out1["w"], valid=out1["valid"], pos=out1["new_pos"], vel=out1["new_vel"], phi_fluid=phi2)
    ref = o.substep(st2, 0.01, max_iter=20000, density_projection=False, eliminate=False)
    assert ref["err_div"] < RES_TOL and ref["dep"].any()
    lab = ref["label"].astype(bool)
    got = {}
    for eliminate in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.set_system_options(True, eliminate)
            ctx.upload("phi_solid", sc.phi_solid); ctx.upload("phi_fluid", phi2)
            ctx.upload_particles(out1["new_pos"], out1["new_vel"], sc.mass)
            for name, arr in zip(("weight_u", "weight_v", "weight_w"), out1["w"]):
                ctx.upload(name, arr)
            for name, arr in zip(("valid_u", "valid_v", "valid_w"), out1["valid"]):
                ctx.upload(name, arr)
            ctx.bin_particles(); ctx.p2g(); ctx.mark_valid_from_active(); ctx.extrapolate(2); ctx.update_prev()
            ctx.add_forces(0.01); ctx.index_cells(); ctx.face_fractions(); ctx.density()
            ctx.build_system(SOLVE_DIVERGENCE, 0.01)
            info = ctx.pcg_solve(RES_TOL, 20000)
            assert info["converged"] == 1
            assert info["dependent_rows"] == (int(ref["dep"].sum()) if eliminate else 0)
            got[eliminate] = (ctx.download("pressure")[lab], info["iterations"])
    assert rel_linf(got[False][0], ref["p_div"]) < 1e-7
    assert rel_linf(got[True][0], ref["p_div"]) < 1e-7
    assert got[True][1] < got[False][1]


def test_rigid_body_coupling():
    """Config 3 in small: a sphere strongly coupled through J^T M^-1 J (src/FluidSimulator.cpp:810-816)."""
    sc = scenes.cube_drop(24, ppc=4, seed=9)
    scenes.add_sphere(sc, radius_world=0.12, centre_world=(0.5, 0.3, 0.5), density=500.0, vel=(0.05, -0.3, 0.02, 0.4, 0.0, -0.2))
    # remove the particles inside the sphere, as the reference's advection keeps them outside solids
    dx = float(sc.dx)
    wpos = (sc.pos + 0.5) * dx
    keep = np.linalg.norm(wpos - np.array([0.5, 0.3, 0.5]), axis=1) > 0.12 + 0.5 * dx
    sc.pos, sc.vel, sc.mass = sc.pos[keep].copy(), sc.vel[keep].copy(), sc.mass[keep].copy()
    sc.phi_fluid = scenes.fluid_levelset(sc, sc.pos)
    rng = np.random.default_rng(2)
    sc.vel = rng.normal(0, 0.1, sc.pos.shape)
    o = Oracle(sc.desc())
    st = fresh_state(sc, o)
    st["obj"] = dict(phi=sc.phi_object, bg=sc.phi_object_bg, com=sc.object_com, minv=sc.object_minv, vel=sc.object_vel)
    dt = 0.005
    ref = o.substep(st, dt, density_projection=False)
    assert ref["err_div"] < RES_TOL
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles(); ctx.p2g(); ctx.mark_valid_from_active(); ctx.extrapolate(2); ctx.update_prev()
        ctx.add_forces(dt); ctx.index_cells(); ctx.face_fractions(); ctx.density()
        for name, want in zip(("obj_weight_u", "obj_weight_v", "obj_weight_w"), ref["ow"]):
            assert np.array_equal(ctx.download(name), want), name
        for name, want in zip(("weight_u", "weight_v", "weight_w"), ref["w"]):
            assert np.array_equal(ctx.download(name), want), name
        ctx.build_system(SOLVE_DIVERGENCE, dt)
        off, diag, rhs = ctx.download_system()
        lab = ref["label"].astype(bool)
        assert np.array_equal(diag, ref["sys_div"][1]) and np.array_equal(off, ref["sys_div"][0])
        assert np.allclose(rhs[lab], ref["sys_div"][3], rtol=0, atol=1e-18 + 1e-13 * np.abs(ref["sys_div"][3]).max())
        info = ctx.pcg_solve(RES_TOL)
        assert info["converged"] == 1
        assert rel_linf(ctx.download("pressure")[lab], ref["p_div"]) < 1e-8
        ctx.project_velocity(dt)
        for name, want in zip(("vel_u", "vel_v", "vel_w"), ref["projected"]):
            assert rel_linf(ctx.download(name), want) < VEL_RTOL, name
        assert rel_linf(ctx.solid_object_velocity(), ref["obj_vel"]) < 1e-8


def test_empty_and_outside_particles():
    sc = scenes.container(12)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload("phi_solid", sc.phi_solid)
        ctx.upload_particles(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.float32))
        assert ctx.bin_particles() == 0
        ctx.p2g()
        assert not ctx.download("vel_v").any()
        assert ctx.index_cells() == 0
        # particles outside the box are counted, binned last and ignored
        pos = np.array([[1e6, 0, 0], [3.2, 4.1, 5.3], [-1e6, 1, 1]], dtype=np.float64)
        ctx.upload_particles(pos, np.ones((3, 3)), np.ones(3, np.float32))
        assert ctx.bin_particles() == 2
        cell_id, cell_start, perm = ctx.download_binning()
        assert cell_id.tolist()[0] == -1 and cell_id.tolist()[2] == -1 and cell_start[-1] == 1 and perm[0] == 1
        ctx.p2g()
        got_pos, got_vel = ctx.download_particles()
        assert np.array_equal(got_pos, pos)


def test_crowded_cell_sort_is_stable():
    """Hundreds of particles in one cell exercise the heap-sort branch of the binning."""
    sc = scenes.container(12)
    o = Oracle(sc.desc())
    rng = np.random.default_rng(8)
    centre = np.array(sc.origin) + np.array(sc.dims) // 2
    pos = np.concatenate([centre + rng.uniform(-0.49, 0.49, (300, 3)), centre + 1 + rng.uniform(-0.49, 0.49, (60, 3)),
                          centre + rng.uniform(-3, 3, (500, 3))])
    rng.shuffle(pos)
    cell_id, cell_start, perm, _ = o.bin_particles(pos)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_particles(pos, np.zeros_like(pos), np.ones(pos.shape[0], np.float32))
        for _ in range(2):                      # the second pass sorts an already sorted set
            ctx.bin_particles()
            g_id, g_start, g_perm = ctx.download_binning()
            assert np.array_equal(g_id, cell_id) and np.array_equal(g_start, cell_start) and np.array_equal(g_perm, perm)


@pytest.mark.parametrize("n", [32, 48])
def test_multigrid_preconditioner_reaches_the_same_solution(n):
    """FLIP_PRECOND_MULTIGRID replaces the sequential MIC(0) of headers/pcgsolver; it has no counterpart in the
    reference, so the check is: same tolerance reached, same pressure as the as-run Jacobi CG (and as the
    oracle's CG), both solves, first and second substep, in far fewer iterations."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_JACOBI, PRECOND_MULTIGRID
    sc = scenes.dam_break(n, ppc=4, seed=6)
    rng = np.random.default_rng(12)
    sc.vel = rng.normal(0, 0.3, sc.pos.shape)
    o = Oracle(sc.desc(), threads=8)
    ref = o.substep(fresh_state(sc, o), 0.005)
    lab = ref["label"].astype(bool)
    results = {}
    for precond in (PRECOND_JACOBI, PRECOND_MULTIGRID):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            step = FlipSubstep(ctx, precond=precond)
            infos = [step.step(0.005)]
            p_den = ctx.download("pressure")[lab]
            pos, vel = ctx.download_particles()
            infos.append(step.step(0.005))       # second substep: stale weights, dependent rows
            pos2, vel2 = ctx.download_particles()
            results[precond] = (infos, p_den, pos, vel, pos2, vel2)
    ja, mg = results[PRECOND_JACOBI], results[PRECOND_MULTIGRID]
    for infos in (ja[0], mg[0]):
        for info in infos:
            for k in ("divergence", "density"):
                assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL
    assert rel_linf(mg[1], ref["p_den"]) < 1e-8 and rel_linf(ja[1], ref["p_den"]) < 1e-8
    assert np.abs(mg[2] - ref["new_pos"]).max() < POS_ATOL and rel_linf(mg[3], ref["new_vel"]) < VEL_RTOL
    assert np.abs(mg[4] - ja[4]).max() < POS_ATOL and rel_linf(mg[5], ja[5]) < VEL_RTOL
    for s in (0, 1):
        for k in ("divergence", "density"):
            assert mg[0][s][k]["iterations"] * 3 < ja[0][s][k]["iterations"], (s, k, mg[0][s][k], ja[0][s][k])


@pytest.mark.parametrize("n,ppc", [(32, 8)])
def test_larger_scene_properties(n, ppc):
    """Size-independent checks at a size the oracle still finishes in seconds: sortedness of the binning,
    P2G -> G2P round trip of a uniform field, converged residual, hydrostatic-like pressure sign."""
    sc = scenes.cube_drop(n, ppc=ppc, seed=4)
    o = Oracle(sc.desc(), threads=8)
    with FlipContext(sc.desc()) as ctx:
        sc.vel[:] = np.array([0.2, -0.5, 0.1])
        ctx.upload_scene(sc)
        ctx.bin_particles()
        cell_id, cell_start, perm = ctx.download_binning()
        ref_id, ref_start, ref_perm, _ = o.bin_particles(sc.pos)
        assert np.array_equal(cell_id, ref_id) and np.array_equal(cell_start, ref_start) and np.array_equal(perm, ref_perm)
        ctx.p2g()
        band = ctx.download("band")
        u, v, w = o.p2g(sc.pos, sc.vel, ref_start, ref_perm, (band >> 1) & 1)
        assert np.array_equal(ctx.download("vel_u"), u) and np.array_equal(ctx.download("vel_v"), v)
        assert np.array_equal(ctx.download("vel_w"), w)
        from fluidsimulation_amd import FlipSubstep
        ctx.upload_scene(sc)
        info = FlipSubstep(ctx).step(0.005)
        assert info["divergence"]["converged"] == 1 and info["density"]["converged"] == 1
        assert info["divergence"]["rel_residual_2"] < RES_TOL and info["density"]["rel_residual_2"] < RES_TOL


def test_particle_advection_uniform_field_and_push_out():
    """flip_advect_particles has no counterpart to compare with here (OpenVDB PointAdvect and the closest-point grid
    are absent: parity unpinned), so the checks are analytic: RK3 through a uniform field is a pure translation by
    v dt / dx index units, and a particle carried into the container wall ends on the wall's surface."""
    sc = scenes.cube_drop(16, ppc=2, seed=8)
    nz, ny, nx = sc.phi_solid.shape
    vel = np.array([0.3, -0.2, 0.1])
    dt = 0.01
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        for name, comp in zip(("vel_u", "vel_v", "vel_w"), vel):
            ctx.upload(name, np.full((nz, ny, nx), comp, np.float64))
        ctx.advect_particles(dt)
        pos, _ = ctx.download_particles()
    shift = vel * float(np.float32(dt)) / float(np.float32(sc.dx))      # dt crosses the ABI as a float
    expect = sc.pos + shift
    lo, hi = np.asarray(sc.box_lo), np.asarray(sc.box_hi)
    inside = np.all((expect > lo + 0.51) & (expect < hi - 0.51), axis=1)     # trilinear phi is exact away from edges
    assert inside.sum() > 100
    assert np.abs(pos[inside] - expect[inside]).max() < 1e-12
    crossed = expect[:, 1] < lo[1]                                           # carried below the floor (y is the only
    crossed &= np.all((expect[:, [0, 2]] > lo[[0, 2]] + 1.0) & (expect[:, [0, 2]] < hi[[0, 2]] - 1.0), axis=1)  # wall nearby)
    if crossed.any():
        assert np.abs(pos[crossed, 1] - lo[1]).max() < 1e-6
        assert np.abs(pos[crossed][:, [0, 2]] - expect[crossed][:, [0, 2]]).max() < 1e-9


def test_constrain_velocity_matches_the_oracle():
    """flip_constrain_velocity (both overloads of src/FluidSimulator.cpp:1180-1321) on random velocity fields with
    the face fractions of a scene that holds the floating sphere: same faces touched, values bit for bit."""
    sc = scenes.add_sphere(scenes.cube_drop(16, ppc=4, seed=21), vel=(0.1, -0.3, 0.05, 0.4, -0.2, 0.3))
    o = Oracle(sc.desc())
    rng = np.random.default_rng(33)
    uvw = [rng.normal(0.0, 0.5, o.shape) for _ in range(3)]
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles(); ctx.p2g(); ctx.index_cells(); ctx.face_fractions()
        w = [ctx.download(n) for n in ("weight_u", "weight_v", "weight_w")]
        ow = [ctx.download(n) for n in ("obj_weight_u", "obj_weight_v", "obj_weight_w")]
        assert sum(int((a == 1).sum()) for a in w) > 100 and sum(int((a == 1).sum()) for a in ow) > 10
        for name, a in zip(("vel_u", "vel_v", "vel_w"), uvw):
            ctx.upload(name, a)
        ctx.constrain_velocity(False)
        got = [ctx.download(n) for n in ("vel_u", "vel_v", "vel_w")]
        want = o.constrain_velocity(sc.phi_solid, sc.phi_solid_bg, w, uvw)
        for a, b, src in zip(got, want, uvw):
            assert np.array_equal(a, b)
            assert (a != src).sum() > 0
        ctx.constrain_velocity(True)
        got2 = [ctx.download(n) for n in ("vel_u", "vel_v", "vel_w")]
        want2 = o.constrain_velocity(sc.phi_object, sc.phi_object_bg, ow, want, sc.object_vel)
        for a, b in zip(got2, want2):
            assert np.array_equal(a, b)
        # a face deep inside the wall with a clean normal: the normal component is gone
        n = 16
        k, j, i = np.argwhere(w[1] == 1)[0]
        assert abs(got[1][k, j, i]) <= np.abs(uvw[1]).max()


def test_fluid_levelset_from_particles_is_the_exact_union_of_balls():
    """flip_fluid_levelset_from_particles against an independent nearest-neighbour search (k-d tree) on the same
    particles: identical up to the last bits of the distance (the two sum the squares in their own order)."""
    sc = scenes.dam_break(24, ppc=4, seed=13)
    radius_index = 1.01 * np.sqrt(3.0) / 2.0        # src/FluidSource.cpp:67 in voxels
    want = scenes.fluid_levelset(sc, sc.pos)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles()
        ctx.fluid_levelset_from_particles(radius_index * float(np.float32(sc.dx)), 3.0)
        got = ctx.download("phi_fluid")
        ctx.bin_particles()                          # the sort is still current: returns without re-sorting
        assert ctx.bin_particles() == 0
    assert np.abs(got - want).max() <= 2e-6 * float(sc.dx)
    assert np.array_equal(got < 0, want < 0)
    assert (got < 0).sum() > 1000 and (got == got.max()).sum() > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("precond", [0, 1])
def test_solving_both_systems_side_by_side_changes_nothing(precond):
    """flip_solve_pair assembles both systems first and iterates on them on two streams (the later batches as
    replayed graphs).  Each solve performs the operations of flip_build_system + flip_pcg_solve in the same order,
    so iteration counts, pressures and the particle state after the substep are identical bit for bit, substep
    after substep."""
    from fluidsimulation_amd import FlipContext, FlipSubstep, scenes
    sc = scenes.dam_break(40, ppc=6, seed=21)
    rng = np.random.default_rng(5)
    sc.vel = rng.normal(0.0, 0.3, sc.pos.shape)
    out = []
    for paired in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            st = FlipSubstep(ctx, precond=precond, max_iter=20000, concurrent_solves=paired)
            infos = [st.step(0.004) for _ in range(3)]
            pos, vel = ctx.download_particles()
            p_div = ctx.download("pressure") if paired else None      # the divergence system is the active one
            if paired:
                ctx.select_system(1)
            p_den = ctx.download("pressure")
            out.append((infos, pos, vel, p_div, p_den))
    (ia, pa, va, pda, pna), (ib, pb, vb, pdb, pnb) = out
    for a, b in zip(ia, ib):
        for k in ("divergence", "density"):
            assert a[k]["converged"] == 1 and b[k]["converged"] == 1
            assert a[k]["iterations"] == b[k]["iterations"] and a[k]["cg_rows"] == b[k]["cg_rows"]
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    assert np.array_equal(pna, pnb)          # density pressure of the last substep
    # sequentially the divergence pressure has been overwritten by the density solve; paired both are kept
    assert pda is None and np.abs(pdb).max() > 0


@pytest.mark.gpu
def test_side_by_side_solves_with_a_rigid_body_and_with_no_fluid():
    """The coupled divergence system takes the plain-launch path of flip_solve_pair (its kernels carry per-solve
    arguments); an empty scene has no rows in either system."""
    sc = scenes.cube_drop(24, ppc=4, seed=9)
    scenes.add_sphere(sc, radius_world=0.12, centre_world=(0.5, 0.3, 0.5), density=500.0, vel=(0.05, -0.3, 0.02, 0.4, 0.0, -0.2))
    dx = float(sc.dx)
    keep = np.linalg.norm((sc.pos + 0.5) * dx - np.array([0.5, 0.3, 0.5]), axis=1) > 0.12 + 0.5 * dx
    sc.pos, sc.vel, sc.mass = sc.pos[keep].copy(), sc.vel[keep].copy(), sc.mass[keep].copy()
    sc.phi_fluid = scenes.fluid_levelset(sc, sc.pos)
    sc.vel = np.random.default_rng(2).normal(0, 0.1, sc.pos.shape)
    from fluidsimulation_amd import FlipSubstep
    got = []
    for paired in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            st = FlipSubstep(ctx, precond=0, max_iter=20000, rigid_body=True, constrain=True, concurrent_solves=paired)
            infos = [st.step(0.004) for _ in range(2)]
            got.append((infos, ctx.download_particles(), ctx.solid_object_velocity()))
    for k in ("divergence", "density"):
        assert got[0][0][1][k]["iterations"] == got[1][0][1][k]["iterations"] and got[1][0][1][k]["converged"] == 1
    assert np.array_equal(got[0][1][0], got[1][1][0]) and np.array_equal(got[0][1][1], got[1][1][1])
    assert np.array_equal(got[0][2], got[1][2])
    empty = scenes.container(12)
    with FlipContext(empty.desc()) as ctx:
        ctx.upload("phi_solid", empty.phi_solid)
        ctx.upload_particles(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.float32))
        info = FlipSubstep(ctx, precond=1, concurrent_solves=True).step(0.004)
        for k in ("divergence", "density"):
            assert info[k]["rows"] == 0 and info[k]["iterations"] == 0


@pytest.mark.gpu
def test_p2g_on_a_wide_box_with_a_crowded_line_is_bit_exact():
    """A box more than a wave wide (64 cells) with a line of cells holding 40 particles each: velocities, activity
    and density equal the oracle's bit for bit."""
    sc = scenes.dam_break(112, ppc=3, seed=4)
    assert sc.dims[0] >= 64
    rng = np.random.default_rng(17)
    # a crowded patch: 40 extra particles per cell in a few cells of one x-line
    lo = np.array(sc.origin) + np.array(sc.dims) // 2
    extra = np.concatenate([lo + np.array([dxc, 0, 0]) + rng.uniform(-0.49, 0.49, (40, 3)) for dxc in range(-12, 12)])
    sc.pos = np.concatenate([sc.pos, extra])
    sc.vel = rng.normal(0.0, 0.4, sc.pos.shape)
    sc.mass = np.concatenate([sc.mass, np.full(extra.shape[0], sc.mass[0], np.float32)])
    o = Oracle(sc.desc())
    _, cell_start, perm, _ = o.bin_particles(sc.pos)
    pm2 = o.particle_mask(cell_start, 2)
    want = o.p2g(sc.pos, sc.vel, cell_start, perm, pm2)
    dens, _ = o.density(sc.pos, sc.mass, cell_start, perm, pm2, sc.phi_fluid)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles()
        ctx.p2g()
        for name, ref in zip(("vel_u", "vel_v", "vel_w"), want):
            assert np.array_equal(ctx.download(name), ref), name
        assert np.array_equal(ctx.download("active").astype(bool), pm2.astype(bool))
        assert np.array_equal(ctx.download("density"), dens)


def test_lds_staged_gather_gives_the_same_bits():
    """The LDS-staged form of the gather (flip_set_gather_mode 1) on the wide box with the crowded line: P2G, activity and
    density (fused and stand-alone) equal the oracle's bit for bit, with tiles both staged in LDS and -- around the
    crowded cells -- walked from global memory."""
    sc = scenes.dam_break(112, ppc=3, seed=4)
    rng = np.random.default_rng(17)
    lo = np.array(sc.origin) + np.array(sc.dims) // 2
    extra = np.concatenate([lo + np.array([dxc, 0, 0]) + rng.uniform(-0.49, 0.49, (90, 3)) for dxc in range(-12, 12)])
    sc.pos = np.concatenate([sc.pos, extra])
    sc.vel = rng.normal(0.0, 0.4, sc.pos.shape)
    sc.mass = np.concatenate([sc.mass, np.full(extra.shape[0], sc.mass[0], np.float32)])
    sc.mass = (sc.mass * rng.uniform(0.5, 1.5, sc.mass.shape)).astype(np.float32)
    o = Oracle(sc.desc())
    _, cell_start, perm, _ = o.bin_particles(sc.pos)
    pm2 = o.particle_mask(cell_start, 2)
    want = o.p2g(sc.pos, sc.vel, cell_start, perm, pm2)
    dens, _ = o.density(sc.pos, sc.mass, cell_start, perm, pm2, sc.phi_fluid)
    with FlipContext(sc.desc()) as ctx:
        ctx.set_gather_mode(1)
        ctx.upload_scene(sc)
        ctx.bin_particles()
        before = ctx.gather_stats()
        ctx.p2g()
        after = ctx.gather_stats()
        assert after["staged_tiles"] > before["staged_tiles"] and after["unstaged_tiles"] > before["unstaged_tiles"]
        for name, ref in zip(("vel_u", "vel_v", "vel_w"), want):
            assert np.array_equal(ctx.download(name), ref), name
        assert np.array_equal(ctx.download("active").astype(bool), pm2.astype(bool))
        assert np.array_equal(ctx.download("density"), dens)
        ctx.upload("phi_fluid", sc.phi_fluid)      # marks the fused density stale: the density-only launch runs
        ctx.upload("density", np.zeros_like(dens))
        ctx.density()
        assert np.array_equal(ctx.download("density"), dens)
        assert ctx.gather_stats()["staged_tiles"] > after["staged_tiles"]


def test_full_domain_dam_break_properties():
    """The container of the 512^3 configuration (walls at 1 and n-2 on every axis) at reduced size: binning sorted and
    complete, both solves converged to the reference tolerance, particles stay inside the container over several whole
    substeps."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(48, ppc=8, seed=3, exact_levelset=False, full_domain=True)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        assert ctx.bin_particles() == 0
        cell_id, cell_start, perm = ctx.download_binning()
        assert np.all(np.diff(cell_start) >= 0) and cell_start[-1] == sc.pos.shape[0]
        assert np.all(np.diff(cell_id[perm]) >= 0)                       # slots ascend by cell
        assert np.array_equal(np.sort(perm), np.arange(sc.pos.shape[0]))  # every particle once
        o = Oracle(sc.desc())
        want_id, want_start, want_perm, _ = o.bin_particles(sc.pos)
        assert np.array_equal(cell_id, want_id) and np.array_equal(cell_start, want_start) and np.array_equal(perm, want_perm)
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=5000, concurrent_solves=True, constrain=True, advect=True,
                           levelset_radius=radius)
        for _ in range(3):
            info = step.step(0.004)
            for k in ("divergence", "density"):
                assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL
        pos, vel = ctx.download_particles()
        assert np.isfinite(pos).all() and np.isfinite(vel).all()
        lo, hi = np.asarray(sc.box_lo), np.asarray(sc.box_hi)
        assert np.all(pos > lo - 0.5) and np.all(pos < hi + 0.5)
        assert ctx.bin_particles() == 0


def test_redistribution_of_overdense_cells_matches_the_oracle():
    """An over-dense block (32 particles per cell): the re-seeding of src/Functors.cpp:268-322, hash-driven on both sides,
    touches the same cells, keeps every particle in its cell and gives bit-equal positions; a second seed gives others."""
    sc = scenes.cube_drop(16, ppc=32, seed=21, exact_levelset=False)
    o = Oracle(sc.desc())
    cell_id, cell_start, perm, _ = o.bin_particles(sc.pos)
    band2 = o.particle_mask(cell_start, 2)
    dens, n_over = o.density(sc.pos, sc.mass * 4.0, cell_start, perm, band2, sc.phi_fluid)   # 4 x the mass: over-dense
    assert n_over > 50
    want, cells, moved = o.redistribute(sc.pos, cell_start, perm, dens, seed=1234)
    assert cells > 50 and moved >= 8 * cells
    with FlipContext(sc.desc()) as ctx:
        ctx.upload("phi_solid", sc.phi_solid)
        ctx.upload("phi_fluid", sc.phi_fluid)
        ctx.upload_particles(sc.pos, sc.vel, sc.mass * 4.0)
        ctx.bin_particles()
        ctx.p2g()
        assert np.array_equal(ctx.download("density"), dens)
        got_cells, got_moved = ctx.redistribute_overdense(1234)
        pos, _ = ctx.download_particles()
        assert (got_cells, got_moved) == (cells, moved)
        assert np.array_equal(pos, want)
        changed = np.any(pos != sc.pos, axis=1)
        assert changed.sum() == moved
        # every particle stays in its cell, and the untouched ones are untouched
        new_id, _, _, _ = o.bin_particles(pos)
        assert np.array_equal(new_id, cell_id)
        # each sub-cell of a touched cell's lattice receives its share: with 32 particles f = 3, 27 sub-cells
        c = int(cell_id[np.nonzero(changed)[0][0]])
        inside = pos[cell_id == c]
        if inside.shape[0] >= 27:
            k, j, i = np.unravel_index(c, (sc.dims[2], sc.dims[1], sc.dims[0]))
            corner = np.array([sc.origin[0] + i, sc.origin[1] + j, sc.origin[2] + k]) - 0.5
            sub = np.floor((inside - corner) * 3).astype(int)
            assert len({tuple(s) for s in sub}) == 27
        # another seed, other positions; the same seed again after a re-sort, the same cells
        ctx.upload_particles(sc.pos, sc.vel, sc.mass * 4.0)
        ctx.bin_particles()
        ctx.p2g()
        ctx.redistribute_overdense(99)
        pos2, _ = ctx.download_particles()
        assert not np.array_equal(pos2, pos) and np.array_equal(np.any(pos2 != sc.pos, axis=1), changed)


def test_redistribution_lattice_at_perfect_cube_counts():
    """src/Functors.cpp:288 sizes the lattice with floor(pow(n, 1./3.)), which gives 3 for 64 particles and 4 for 125 (the
    exponent is below a third); the oracle evaluates that expression with the C library, the device restates it in
    integers.  Cells with exactly 27, 64, 125 and 126 particles."""
    sc = scenes.cube_drop(16, ppc=8, seed=5, exact_levelset=False)
    o = Oracle(sc.desc())
    cell_id, cell_start, _, _ = o.bin_particles(sc.pos)
    counts = np.diff(cell_start[: sc.n_cells + 1])
    filled = np.nonzero(counts == 8)[0]
    rng = np.random.default_rng(8)
    extra = []
    targets = dict(zip(filled[[10, 80, 150, 220]], (27, 64, 125, 126)))
    for c, total in targets.items():
        k, j, i = np.unravel_index(c, (sc.dims[2], sc.dims[1], sc.dims[0]))
        centre = np.array([sc.origin[0] + i, sc.origin[1] + j, sc.origin[2] + k], dtype=np.float64)
        extra.append(centre + rng.uniform(-0.45, 0.45, (total - 8, 3)))
    pos = np.concatenate([sc.pos] + extra)
    vel = np.zeros_like(pos)
    mass = np.full(pos.shape[0], sc.mass[0] * 4.0, dtype=np.float32)
    cell_id, cell_start, perm, _ = o.bin_particles(pos)
    counts = np.diff(cell_start[: sc.n_cells + 1])
    assert all(counts[c] == t for c, t in targets.items())
    band2 = o.particle_mask(cell_start, 2)
    dens, _ = o.density(pos, mass, cell_start, perm, band2, sc.phi_fluid)
    assert all(dens.ravel()[c] > 1.5 * sc.desc()["rest_density"] for c in targets)
    want, cells, moved = o.redistribute(pos, cell_start, perm, dens, seed=77)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload("phi_solid", sc.phi_solid)
        ctx.upload("phi_fluid", sc.phi_fluid)
        ctx.upload_particles(pos, vel, mass)
        ctx.bin_particles()
        ctx.p2g()
        assert (cells, moved) == ctx.redistribute_overdense(77)
        got, _ = ctx.download_particles()
    assert np.array_equal(got, want)
    # the lattice the 64-particle cell received: 27 sub-cells of a third of a cell, not 64 of a quarter
    c64 = [c for c, t in targets.items() if t == 64][0]
    k, j, i = np.unravel_index(c64, (sc.dims[2], sc.dims[1], sc.dims[0]))
    corner = np.array([sc.origin[0] + i, sc.origin[1] + j, sc.origin[2] + k]) - 0.5
    inside = got[cell_id == c64]
    assert len({tuple(v) for v in np.floor((inside - corner) * 3).astype(int)}) == 27
    occupancy = np.unique(np.floor((inside - corner) * 3).astype(int), axis=0, return_counts=True)[1]
    assert occupancy.min() >= 2 and occupancy.max() <= 3


def test_particle_advection_matches_the_oracle_restatement():
    """flip_advect_particles against orc_advect_particles (RK3 + push-out, src/FluidDomain.cpp:63-75; OpenVDB's
    PointAdvect / cpt restated on both sides, parity unpinned) on a random velocity field, with the floating sphere as a
    second obstacle: same expressions in the same order, so positions agree bit for bit; some particles must have been
    pushed out of the wall and out of the sphere."""
    sc = scenes.add_sphere(scenes.cube_drop(16, ppc=6, seed=31))
    o = Oracle(sc.desc())
    rng = np.random.default_rng(5)
    uvw = [rng.normal(0.0, 1.5, o.shape) for _ in range(3)]
    dt = 0.02
    for with_object in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload("phi_solid", sc.phi_solid)
            ctx.upload("phi_fluid", sc.phi_fluid)
            ctx.upload_particles(sc.pos, sc.vel, sc.mass)
            if with_object:
                ctx.set_solid_object(sc.phi_object, sc.phi_object_bg, sc.object_com, sc.object_minv, sc.object_vel)
            for name, a in zip(("vel_u", "vel_v", "vel_w"), uvw):
                ctx.upload(name, a)
            ctx.advect_particles(dt)
            got, _ = ctx.download_particles()
        want = o.advect_particles(sc.pos, uvw, sc.phi_solid, dt, sc.phi_object if with_object else None, sc.phi_object_bg)
        assert np.array_equal(got, want)
        free = o.advect_particles(sc.pos, uvw, np.full(o.shape, 1.0, np.float32), dt)   # no obstacle anywhere
        pushed = np.any(free != want, axis=1)
        assert pushed.sum() > 10 and (~pushed).sum() > 100


def test_fluid_levelset_from_particles_matches_the_oracle_restatement():
    """flip_fluid_levelset_from_particles (shell search with pruning, three cells per thread) against the oracle's plain
    minimum over every particle within reach (src/FluidDomain.cpp:50-61 restated as the exact union of balls): the
    squared distance is the same expression on both sides, so the fields agree bit for bit."""
    for sc in (scenes.dam_break(24, ppc=4, seed=13), scenes.cube_drop(20, ppc=1, seed=2)):
        o = Oracle(sc.desc(), threads=8)
        _, cell_start, perm, _ = o.bin_particles(sc.pos)
        radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
        want = o.fluid_levelset(sc.pos, cell_start, perm, sc.phi_solid, radius, 3.0)
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            ctx.bin_particles()
            ctx.fluid_levelset_from_particles(radius, 3.0)
            got = ctx.download("phi_fluid")
        assert np.array_equal(got, want)
        assert (got < 0).sum() > 500 and (got == got.max()).sum() > 500


# ---- the benchmark configurations at full size: properties that do not need the oracle's solver ----

def _substep_properties(ctx, sc, n_steps, rigid_body=False):
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=5000, concurrent_solves=not rigid_body, constrain=True, advect=True,
                       levelset_radius=radius, rigid_body=rigid_body)
    for _ in range(n_steps):
        info = step.step(0.004)
        for k in ("divergence", "density"):
            assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL, (k, info[k])
    pos, vel = ctx.download_particles()
    assert np.isfinite(pos).all() and np.isfinite(vel).all()
    lo, hi = np.asarray(sc.box_lo), np.asarray(sc.box_hi)
    assert np.all(pos > lo - 0.5) and np.all(pos < hi + 0.5)          # nothing left the container
    assert ctx.bin_particles() == 0
    cell_id, cell_start, perm = ctx.download_binning()
    assert np.all(np.diff(cell_id[perm]) >= 0) and cell_start[-1] == pos.shape[0]
    return info


def test_config_cube_drop_128_p2g_against_the_oracle_and_substeps():
    """BASELINE config 2 (cube drop at 128^3, 8 particles per cell, 2.1 M particles): binning and the P2G + density gather
    bit for bit against the oracle at full size, then whole substeps with the size-independent checks."""
    sc = scenes.cube_drop(128, ppc=8, seed=12345, exact_levelset=False)
    rng = np.random.default_rng(3)
    sc.vel = rng.normal(0.0, 0.2, sc.pos.shape)
    o = Oracle(sc.desc(), threads=16)
    cell_id, cell_start, perm, _ = o.bin_particles(sc.pos)
    band2 = o.particle_mask(cell_start, 2)
    want = o.p2g(sc.pos, sc.vel, cell_start, perm, band2)
    want_density, _ = o.density(sc.pos, sc.mass, cell_start, perm, band2, sc.phi_fluid)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        assert ctx.bin_particles() == 0
        got_id, got_start, got_perm = ctx.download_binning()
        assert np.array_equal(got_id, cell_id) and np.array_equal(got_start, cell_start) and np.array_equal(got_perm, perm)
        ctx.p2g()
        for name, w in zip(("vel_u", "vel_v", "vel_w"), want):
            assert np.array_equal(ctx.download(name), w), name
        assert np.array_equal(ctx.download("density"), want_density)
        _substep_properties(ctx, sc, 2)


def test_config_floating_ball_128_substeps():
    """BASELINE config 3 (config 2 plus the half-density sphere, strongly coupled): the coupled solve converges to the
    reference tolerance at full size and the body's velocity stays finite and moves under gravity and pressure."""
    sc = scenes.add_sphere(scenes.cube_drop(128, ppc=8, seed=12345, exact_levelset=False), density=500.0)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        info = _substep_properties(ctx, sc, 2, rigid_body=True)
        v = ctx.solid_object_velocity()
        assert np.isfinite(v).all()
        assert info["divergence"]["rows"] > 200000


def test_config_dam_break_256_substeps():
    """BASELINE config 4 on one GPU (side-drop dam break at 256^3, 12.4 M particles, density projection on): sortedness,
    residuals at the reference tolerance, containment, over whole substeps at full size."""
    sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
    with FlipContext(sc.desc()) as ctx:
        ctx.reserve_particles(sc.pos.shape[0] + 4096)
        ctx.upload_scene(sc)
        info = _substep_properties(ctx, sc, 2)
        assert info["divergence"]["cg_rows"] > 1400000


@pytest.mark.parametrize("concurrent", [False, True])
def test_a_solve_does_not_depend_on_what_the_context_solved_before(concurrent):
    """Every solver array is dense over the box and only partly rewritten by a solve (the rows of the current system), so
    a context that has solved a system with other rows must clear what the stencil kernels read off the rows.  Two
    contexts run a substep on two different particle sets (different liquid regions, hence different rows) and then
    both run the same third set.  The labels of that substep legitimately depend on the history (the reference labels
    with the previous substep's face weights, src/FluidSimulator.cpp:292-300), but only through dependent rows -- cells
    with phi_fluid >= 0, which are substituted after the solve; the CG systems over the liquid cells are the same, so
    both solves must take the same iterations to the same residual and give the same pressure there, bit for bit."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(32, ppc=4, seed=6)
    rng = np.random.default_rng(3)
    vel = rng.normal(0, 0.3, sc.pos.shape)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    order = np.argsort(sc.pos[:, 1], kind="stable")
    first = {"low half": order[: len(order) // 2], "every third": np.arange(0, len(order), 3), "nothing": None}
    common = order[len(order) // 4:]
    results = {}
    for name, sel in first.items():
        with FlipContext(sc.desc()) as ctx:
            ctx.upload("phi_solid", sc.phi_solid)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, levelset_radius=radius, concurrent_solves=concurrent)
            rows_first = -1
            if sel is not None:
                ctx.upload_particles(sc.pos[sel], vel[sel], sc.mass[sel])
                rows_first = step.step(0.005)["divergence"]["cg_rows"]
            ctx.upload_particles(sc.pos[common], vel[common], sc.mass[common])
            info = step.step(0.005)
            for k in ("divergence", "density"):
                assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL, (name, k, info[k])
            assert info["divergence"]["cg_rows"] != rows_first
            liquid = ctx.download("phi_fluid") < 0
            keys = ("iterations", "cg_rows", "rel_residual_2", "rhs_norm_2")
            results[name] = (ctx.download("pressure")[liquid], [[info[k][q] for q in keys] for k in ("divergence", "density")])
    want = results["nothing"]
    assert want[0].size > 1000 and np.abs(want[0]).max() > 0
    for name in ("low half", "every third"):
        assert results[name][1] == want[1], (name, results[name][1], want[1])
        assert np.array_equal(results[name][0], want[0]), name


def test_forty_substeps_of_the_bench_scene_stay_finite_and_converged():
    """The whole substep (level set from particles, constrain, both solves side by side, advection) forty times in a row
    on the bench scene at 64^3: every solve converges in about the same number of iterations (the batch sizing of the
    solver loop relies on that), nothing becomes non-finite, and all but a handful of particles stay in the container
    (the ghost-fluid pressure gradient divides by theta clamped at 0.001, src/FluidSimulator.cpp:1067-1071, which throws single
    particles off the free surface in the reference's own scheme; the oracle reproduces that step by step)."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break_slab(64, ppc=8, seed=12345)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=2000, concurrent_solves=True, constrain=True, advect=True,
                           levelset_radius=radius)
        its = []
        for s in range(40):
            info = step.step(0.002)
            for k in ("divergence", "density"):
                assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL, (s, k, info[k])
            its.append((info["divergence"]["iterations"], info["density"]["iterations"]))
        pos, vel = ctx.download_particles()
    assert np.isfinite(pos).all() and np.isfinite(vel).all()
    later = np.array(its[2:])
    assert later.min() >= 10 and later.max() <= 40, its
    lo = np.array(sc.origin, dtype=np.float64) - 0.5
    hi = lo + np.array(sc.dims, dtype=np.float64)
    outside = ((pos < lo) | (pos > hi)).any(axis=1).sum()
    assert outside <= pos.shape[0] // 1000, outside


def test_a_context_on_a_non_blocking_stream_gives_the_same_substeps():
    """Everything the library enqueues must be ordered by the stream it was given and by its own events alone: a context
    created on a non-blocking stream (which the null stream does not synchronise with) has to reproduce the substeps of
    a context on the null stream bit for bit, from the very first one, when every array is freshly allocated."""
    import ctypes
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    hip = ctypes.CDLL("libamdhip64.so")
    sc = scenes.dam_break(32, ppc=4, seed=6)
    rng = np.random.default_rng(5)
    sc.vel = rng.normal(0, 0.3, sc.pos.shape)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    results = []
    for own_stream in (False, True):
        stream = ctypes.c_void_p(0)
        if own_stream:
            assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0   # 1 = hipStreamNonBlocking
        with FlipContext(sc.desc(), stream=stream.value) as ctx:
            ctx.upload_scene(sc)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
            infos = [step.step(0.004) for _ in range(3)]
            pos, vel = ctx.download_particles()
            results.append((pos, vel, ctx.download("pressure"), [(i["divergence"]["iterations"], i["density"]["iterations"]) for i in infos]))
        if own_stream:
            assert hip.hipStreamDestroy(stream) == 0
    a, b = results
    assert a[3] == b[3], (a[3], b[3])
    for q, name in enumerate(("positions", "velocities", "pressure")):
        assert np.array_equal(a[q], b[q]), (name, int((a[q] != b[q]).sum()), float(np.abs(a[q] - b[q]).max()))


@pytest.mark.parametrize("concurrent", [False, True])
def test_iteration_cap_is_honoured_whatever_the_previous_solve_took(concurrent):
    """The solver loop sizes its launch batches from the iteration count of the previous solve of the same kind; a cap
    below that count must still stop the solve at the cap (Eigen's NoConvergence), and the next uncapped solve must
    converge as before."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(32, ppc=4, seed=6)
    rng = np.random.default_rng(9)
    sc.vel = rng.normal(0, 0.3, sc.pos.shape)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        free = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, concurrent_solves=concurrent)
        first = free.step(0.004)
        assert first["density"]["converged"] == 1 and first["density"]["iterations"] > 8
        capped = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, concurrent_solves=concurrent, max_iter=3)
        info = capped.step(0.004)
        for k in ("divergence", "density"):
            assert info[k]["converged"] == 0 and info[k]["iterations"] == 3, (k, info[k])
        again = free.step(0.004)
        for k in ("divergence", "density"):
            assert again[k]["converged"] == 1 and again[k]["rel_residual_2"] < RES_TOL, (k, again[k])


def _whole_substeps_follow_the_oracle(sc, dt, n_steps):
    """Whole substeps -- level set rebuilt from the particles (src/FluidSimulator.cpp:324), both constrain_velocity calls,
    RK3 advection with push-out, density projection -- on the device and, independently, in the oracle: the state a
    substep inherits from the one before (valid flags, stale face weights, moved particles: :343-345, :486-488) is carried
    by each side on its own, neither is re-synchronised with the other.  After every substep: labels and row count
    bit-exact, face fractions bit-exact, fluid level set equal to rounding, positions <= 1e-5 cell, velocities <= 1e-4
    relative."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0    # src/FluidSource.cpp:67
    o = Oracle(sc.desc(), threads=16)
    st = fresh_state(sc, o)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, constrain=True, advect=True, levelset_radius=radius,
                           concurrent_solves=True)
        for k in range(n_steps):
            _, cell_start, perm, _ = o.bin_particles(st["pos"])
            st["phi_fluid"] = o.fluid_levelset(st["pos"], cell_start, perm, sc.phi_solid, radius)
            ref = o.substep(st, dt, max_iter=20000, constrain=True)
            assert ref["err_div"] < RES_TOL and ref["err_den"] < RES_TOL
            moved = o.advect_particles(st["pos"], ref["final_vel"], sc.phi_solid, dt)      # :426, before the displacements (:429)
            new_pos = o.apply_displacements(moved, ref["disp"])
            info = step.step(dt)
            assert info["divergence"]["converged"] == 1 and info["density"]["converged"] == 1
            assert info["divergence"]["rows"] == ref["n_rows"], k
            assert np.array_equal(ctx.download("label"), ref["label"]), k
            assert np.abs(ctx.download("phi_fluid") - st["phi_fluid"]).max() <= 1e-6 * float(sc.dx), k
            for name, want in zip(("weight_u", "weight_v", "weight_w"), ref["w"]):
                assert np.array_equal(ctx.download(name), want), (k, name)
            pos, vel = ctx.download_particles()
            assert np.abs(pos - new_pos).max() < POS_ATOL, k
            assert rel_linf(vel, ref["new_vel"]) < VEL_RTOL, k
            st = dict(st, pos=new_pos, vel=ref["new_vel"], w=ref["w"], valid=ref["valid"])
    return ref["n_rows"]


def test_four_substeps_of_the_default_scene_follow_the_oracle():
    """BASELINE config 1 (the cube drop at main.cpp's default 64^3) over FOUR whole substeps with the level set rebuilt
    from the particles, both constrain_velocity calls and the RK3 advection switched on."""
    sc = scenes.cube_drop(64, ppc=4, seed=11, exact_levelset=False)
    rng = np.random.default_rng(5)
    sc.vel = rng.normal(0.0, 0.1, sc.pos.shape)
    _whole_substeps_follow_the_oracle(sc, 0.004, 4)


def test_two_whole_substeps_of_the_bench_configuration_follow_the_oracle():
    """The configuration bench.py times, as it times it (side-drop dam break at 256^3, 12.4 M particles, dt = 0.001, the
    whole loop body of advance_flip_pic except the source), over two consecutive substeps at full size."""
    sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
    rng = np.random.default_rng(31)
    sc.vel = rng.normal(0.0, 0.15, sc.pos.shape)
    assert _whole_substeps_follow_the_oracle(sc, 0.001, 2) > 1500000


def test_config_floating_ball_128_stages_against_the_oracle():
    """BASELINE config 3 at full size (cube drop at 128^3 with the half-density sphere, 2.1 M particles): cell labels and row
    numbers, the container's, the fluid's and the body's face fractions, and both assembled systems (stencil, diagonal,
    right-hand side with the body's J^T V) against the 16-thread oracle, bit for bit except the J^T V sum; no CPU solve."""
    sc = scenes.add_sphere(scenes.cube_drop(128, ppc=8, seed=12345, exact_levelset=False), density=500.0,
                           vel=(0.02, -0.3, 0.01, 0.2, 0.0, -0.1))
    rng = np.random.default_rng(8)
    sc.vel = rng.normal(0.0, 0.15, sc.pos.shape)
    dt = 0.004
    o = Oracle(sc.desc(), threads=16)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles(); ctx.p2g(); ctx.mark_valid_from_active(); ctx.extrapolate(2); ctx.update_prev(); ctx.add_forces(dt)
        n_rows = ctx.index_cells(); ctx.face_fractions(); ctx.density()
        _, cell_start, _ = ctx.download_binning()
        uvw = [ctx.download(nm) for nm in ("vel_u", "vel_v", "vel_w")]
        dens = ctx.download("density")
        z32 = lambda: np.zeros(o.shape, np.float32)  # noqa: E731
        label, index, want_rows = o.index_cells(o.particle_mask(cell_start, 1), sc.phi_fluid, sc.phi_solid, [z32(), z32(), z32()])
        assert n_rows == want_rows and n_rows > 200000
        assert np.array_equal(ctx.download("label"), label) and np.array_equal(ctx.download("row_index"), index)
        ow = o.face_fractions(sc.phi_object, sc.phi_object_bg, o.weight_mask(sc.phi_object, sc.phi_object_bg))
        wm = o.weight_mask(sc.phi_fluid, sc.phi_fluid_bg)
        fw = o.face_fractions(sc.phi_fluid, sc.phi_fluid_bg, wm)
        w = o.face_fractions(sc.phi_solid, sc.phi_solid_bg, wm)
        for name, want in zip(("obj_weight_u", "obj_weight_v", "obj_weight_w"), ow):
            assert np.array_equal(ctx.download(name), want), name
        o.combine_weights(w, ow)
        for name, want in zip(("weight_u", "weight_v", "weight_w"), w):
            assert np.array_equal(ctx.download(name), want), name
        for name, want in zip(("fluid_weight_u", "fluid_weight_v", "fluid_weight_w"), fw):
            assert np.array_equal(ctx.download(name), want), name
        lab = label.astype(bool)
        # divergence system with the rank-6 coupling's right-hand side
        off, diag, rhs = o.build_system(0, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw)
        J = o.build_J(index, ow, sc.object_com)
        b = o.to_csr(index, off, diag, rhs)[3] - J.T @ np.asarray(sc.object_vel, np.float64)
        ctx.build_system(SOLVE_DIVERGENCE, dt)
        goff, gdiag, grhs = ctx.download_system()
        assert np.array_equal(gdiag, diag) and np.array_equal(goff, off)
        assert np.allclose(grhs[lab], b, rtol=0, atol=1e-18 + 1e-13 * np.abs(b).max())
        # density system (no coupling: src/FluidSimulator.cpp:846-849)
        off2, diag2, rhs2 = o.build_system(1, dt, label, sc.phi_fluid, sc.phi_solid, w, density=dens)
        ctx.build_system(SOLVE_DENSITY, dt)
        goff, gdiag, grhs = ctx.download_system()
        assert np.array_equal(gdiag, diag2) and np.array_equal(goff, off2) and np.array_equal(grhs, rhs2)


def test_velocity_maximum_and_cfl_against_numpy_and_the_oracle():
    """flip_velocity_max (the reduction of compute_cfl, src/FluidSimulator.cpp:448-458) is the plain component-wise maximum
    over the active voxels, bit for bit, the allocation-free twin agrees, and the oracle's compute_cfl gives the substep
    size the host class derives from it."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(32, ppc=4, seed=9)
    rng = np.random.default_rng(2)
    sc.vel = rng.normal(0.0, 0.3, sc.pos.shape)
    o = Oracle(sc.desc())
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000)
        for _ in range(2):
            step.step(0.004)
            uvw = [ctx.download(nm) for nm in ("vel_u", "vel_v", "vel_w")]
            active = ctx.download("active").astype(bool)
            want = np.array([a[active].max() for a in uvw])
            got = np.asarray(ctx.velocity_max())
            assert np.array_equal(got, want)
            assert np.array_equal(np.asarray(ctx.velocity_max_static()), want)
            dt, m = o.compute_cfl(uvw, active)
            assert np.array_equal(m, want)
            assert dt == np.float32(np.float64(np.float32(sc.dx)) / (want.max() + np.float64(np.float32(10e-37))))
            body = np.array([0.0, -2.0 * want.max() - 1.0, 0.0])
            dt_b, m_b = o.compute_cfl(uvw, active, body)
            assert m_b[1] == -body[1] and dt_b < dt


def test_buffer_swaps_exchange_front_and_back():
    """MacGrid::swapVelocityBuffers / swapValidMasks (src/MacGrid.cpp:65-70) on the device buffers."""
    sc = scenes.cube_drop(16, ppc=2, seed=3)
    rng = np.random.default_rng(1)
    with FlipContext(sc.desc()) as ctx:
        shape = ctx.download("vel_u").shape
        front = [rng.normal(size=shape) for _ in range(3)]
        back = [rng.normal(size=shape) for _ in range(3)]
        vf = [(rng.random(shape) < 0.5).astype(np.uint8) for _ in range(3)]
        vb = [(rng.random(shape) < 0.5).astype(np.uint8) for _ in range(3)]
        for a, n in enumerate("uvw"):
            ctx.upload(f"vel_{n}", front[a]); ctx.upload(f"vel_back_{n}", back[a])
            ctx.upload(f"valid_{n}", vf[a]); ctx.upload(f"valid_back_{n}", vb[a])
        ctx.swap_velocity_buffers()
        ctx.swap_valid_masks()
        for a, n in enumerate("uvw"):
            assert np.array_equal(ctx.download(f"vel_{n}"), back[a]) and np.array_equal(ctx.download(f"vel_back_{n}"), front[a])
            assert np.array_equal(ctx.download(f"valid_{n}"), vb[a]) and np.array_equal(ctx.download(f"valid_back_{n}"), vf[a])


def test_warm_start_reaches_the_same_pressure_in_fewer_iterations():
    """flip_set_warm_start: a solve that starts from the previous substep's converged solution (Eigen's solveWithGuess
    instead of the reference's solve(), src/FluidSimulator.cpp:823, :1007) stops by the same rule |r| < tol |b|, so it
    must give the same pressure to solver accuracy; the first substep has no previous solution and is identical."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID, SOLVE_DENSITY
    sc = scenes.dam_break_slab(64, ppc=8, seed=12345)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
    runs = {}
    for warm in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            ctx.set_warm_start(warm)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
            infos = [step.step(0.002) for _ in range(6)]
            p_div = ctx.download("pressure")
            ctx.select_system(SOLVE_DENSITY)
            p_den = ctx.download("pressure")
            pos, vel = ctx.download_particles()
            runs[warm] = (infos, p_div, p_den, pos, vel)
    cold, hot = runs[False], runs[True]
    for infos in (cold[0], hot[0]):
        for info in infos:
            for k in ("divergence", "density"):
                assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL, info[k]
    for k in ("divergence", "density"):
        assert hot[0][0][k]["iterations"] == cold[0][0][k]["iterations"]
        assert hot[0][0][k]["rhs_norm_2"] == cold[0][0][k]["rhs_norm_2"]
    its = lambda infos: sum(i[k]["iterations"] for i in infos[1:] for k in ("divergence", "density"))
    # at the reference's tolerance the guess saves little or nothing (the rows that change at the free surface leave a first
    # residual of the order of |b|); it must not cost iterations
    assert its(hot[0]) <= its(cold[0]), (its(hot[0]), its(cold[0]))
    # |b| of the later substeps is the norm of the right-hand side, not of the first residual
    for a, b in zip(cold[0][1:], hot[0][1:]):
        for k in ("divergence", "density"):
            assert abs(a[k]["rhs_norm_2"] - b[k]["rhs_norm_2"]) <= 1e-6 * a[k]["rhs_norm_2"], (a[k], b[k])
    assert rel_linf(hot[1], cold[1]) < 1e-8 and rel_linf(hot[2], cold[2]) < 1e-8
    assert np.abs(hot[3] - cold[3]).max() < POS_ATOL and rel_linf(hot[4], cold[4]) < VEL_RTOL


@pytest.mark.parametrize("precond", [0, 1])
def test_the_pcgsolver_stopping_rule_is_reported(precond):
    """flip_pcg_info::iterations_inf / rel_residual_inf watch the rule of headers/pcgsolver/pcg_solver.h:241-258
    (max|r| <= 1e-12 max|b|) beside Eigen's: a solve capped at the reported count satisfies the rule, one capped an
    iteration earlier does not, and the reported ratio is that of the true residual b - A x of the downloaded system."""
    from fluidsimulation_amd import SOLVE_DIVERGENCE
    sc = scenes.dam_break(32, ppc=4, seed=6)
    rng = np.random.default_rng(21)
    sc.vel = rng.normal(0, 0.3, sc.pos.shape)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.bin_particles()
        ctx.p2g()
        ctx.add_forces(0.005)
        ctx.index_cells()
        ctx.face_fractions()
        ctx.build_system(SOLVE_DIVERGENCE, 0.005)
        full = ctx.pcg_solve(1e-15, 0, precond)
        k = full["iterations_inf"]
        assert full["converged"] == 1 and 0 < k <= full["iterations"], full
        assert full["rel_residual_inf"] <= 1e-12
        at = ctx.pcg_solve(1e-15, k, precond)
        assert at["iterations"] == k and at["iterations_inf"] == k and at["rel_residual_inf"] <= 1e-12, at
        off, diag, rhs = ctx.download_system()
        x = ctx.download("pressure")
        xp = np.pad(x, 1)
        ax = diag * x
        ax += off[0] * xp[1:-1, 1:-1, :-2] + off[1] * xp[1:-1, 1:-1, 2:]
        ax += off[2] * xp[1:-1, :-2, 1:-1] + off[3] * xp[1:-1, 2:, 1:-1]
        ax += off[4] * xp[:-2, 1:-1, 1:-1] + off[5] * xp[2:, 1:-1, 1:-1]
        true_ratio = np.abs(rhs - ax).max() / np.abs(rhs).max()
        assert true_ratio <= 1.5e-12 and abs(true_ratio - at["rel_residual_inf"]) <= 0.05 * at["rel_residual_inf"] + 1e-15, (true_ratio, at)
        before = ctx.pcg_solve(1e-15, k - 1, precond)
        assert before["iterations"] == k - 1 and before["iterations_inf"] == -1 and before["rel_residual_inf"] > 1e-12, before


@pytest.mark.parametrize("advect,displace,body", [(True, True, False), (True, False, False), (False, True, False), (False, False, False),
                                                   (True, True, True)])
def test_the_one_pass_particle_update_is_the_three_calls(advect, displace, body):
    """flip_update_particles = flip_g2p, flip_advect_particles, flip_apply_displacements (src/FluidSimulator.cpp:420, :426,
    :429) in one kernel: positions and velocities after three substeps must equal those of the three calls bit for bit,
    with every combination of the two optional stages and with the rigid body's push-out."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.cube_drop(24, ppc=4, seed=9)
    if body:
        sc = scenes.add_sphere(sc, vel=(0.1, -0.3, 0.05, 0.4, -0.2, 0.3))
    rng = np.random.default_rng(17)
    sc.vel = rng.normal(0, 0.5, sc.pos.shape)
    results = []
    for fused in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, density_projection=displace, advect=advect, constrain=True, rigid_body=body,
                               fused_particle_update=fused)
            for _ in range(3):
                step.step(0.004)
            results.append(ctx.download_particles())
    (pa, va), (pb, vb) = results
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    assert np.abs(va - sc.vel).max() > 0 and (not (advect or displace) or np.abs(pa - sc.pos).max() > 0)


@pytest.mark.parametrize("n", [1, 3, 4, 255, 4095, 4096, 4097, 8191, 1_000_003, 4096 * 4096 + 5])
def test_device_scan_is_the_exclusive_prefix_sum(n):
    """The reduce-then-scan behind the binning, the labels, the row lists and the multigrid levels against numpy, at the
    sizes where its structure changes (one tile of 4096, one level of tile sums, two levels above 16.7 M items), at
    lengths that are no multiple of four, and with the data off a 16-byte boundary (scalar loads instead of int4)."""
    rng = np.random.default_rng(n)
    v = rng.integers(0, 7, size=n, dtype=np.int32)
    want = np.concatenate(([0], np.cumsum(v[:-1], dtype=np.int64))).astype(np.int32)
    sc = scenes.cube_drop(8, ppc=1, seed=0)
    with FlipContext(sc.desc()) as ctx:
        for offset in (0, 1, 3):
            assert np.array_equal(ctx.scan_i32(v, offset), want), offset


@pytest.mark.parametrize("body", [False, True])
def test_two_streams_inside_a_substep_give_the_bits_of_one(body):
    """flip_side_fork / flip_side_select / flip_side_join put the labels and face fractions beside the velocity chain
    (P2G, extrapolation, constrain_velocity, forces) and the displacement projection beside the velocity projection.
    flip_face_fractions writes the other buffer of a pair, so the constrain_velocity and label calls issued before it
    still read the previous substep's fractions (src/FluidSimulator.cpp:357 runs before :488).  Three whole substeps with
    the level set rebuilt from the particles, both constrain passes and the advection must end in the same particles,
    labels, face fractions and velocities as with flip_set_side_stream(ctx, 0), and misuse is refused."""
    from fluidsimulation_amd import FlipError, FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(32, ppc=4, seed=12)
    if body:
        sc = scenes.add_sphere(sc, vel=(0.1, -0.3, 0.05, 0.4, -0.2, 0.3))
    rng = np.random.default_rng(23)
    sc.vel = rng.normal(0, 0.4, sc.pos.shape)
    results = []
    for two in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            ctx.set_side_stream(two)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, constrain=True, advect=True, rigid_body=body, concurrent_solves=True,
                               levelset_radius=1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0)
            for _ in range(3):
                step.step(0.002)
            if two:
                with pytest.raises(FlipError):
                    ctx.side_select(True)          # no fork
                ctx.side_fork()
                ctx.side_select(True)
                with pytest.raises(FlipError):
                    ctx.side_fork()                # while the side stream is selected
                ctx.side_join()
            pos, vel = ctx.download_particles()
            results.append((pos, vel, ctx.download("label"), ctx.download("weight_u"), ctx.download("weight_w"),
                            ctx.download("vel_u"), ctx.download("vel_v"), ctx.download("disp_y"), ctx.download("phi_fluid")))
    for a, b in zip(*results):
        assert np.array_equal(a, b)


def test_pair_region_is_refused_where_it_cannot_work_and_leaves_the_context_as_it_was():
    """flip_pair_begin / flip_pair_select / flip_pair_end hand the two system sets of a context to a caller that interleaves
    two solves itself (the slab driver).  Calls that manage the sets on their own are refused in between, a select without
    a begin is an error, and an empty region changes nothing: the substep after it equals the one of an untouched context."""
    from fluidsimulation_amd import FlipError, FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break(32, ppc=4, seed=6)
    rng = np.random.default_rng(4)
    sc.vel = rng.normal(0, 0.3, sc.pos.shape)
    results = []
    for touched in (False, True):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, concurrent_solves=True)
            step.step(0.004)
            if touched:
                with pytest.raises(FlipError):
                    ctx._check(ctx.lib.flip_pair_select(ctx.h, 1))
                ctx._check(ctx.lib.flip_pair_begin(ctx.h))
                with pytest.raises(FlipError):
                    ctx._check(ctx.lib.flip_pair_begin(ctx.h))
                with pytest.raises(FlipError):
                    ctx.solve_pair(0.004, 1e-15, 0, PRECOND_MULTIGRID)
                with pytest.raises(FlipError):
                    ctx.pcg_solve(1e-15, 0, PRECOND_MULTIGRID)
                with pytest.raises(FlipError):
                    ctx._check(ctx.lib.flip_pair_select(ctx.h, 2))
                ctx._check(ctx.lib.flip_pair_select(ctx.h, 1))
                ctx._check(ctx.lib.flip_pair_select(ctx.h, 0))
                ctx._check(ctx.lib.flip_pair_select(ctx.h, 1))
                ctx._check(ctx.lib.flip_pair_end(ctx.h))
                with pytest.raises(FlipError):
                    ctx._check(ctx.lib.flip_pair_end(ctx.h))
            info = step.step(0.004)
            results.append((ctx.download_particles(), ctx.download("pressure"), [info[k]["iterations"] for k in ("divergence", "density")]))
    (pa, va), qa, ia = results[0]
    (pb, vb), qb, ib = results[1]
    assert ia == ib and np.array_equal(pa, pb) and np.array_equal(va, vb) and np.array_equal(qa, qb)


def test_config_dam_break_256_stages_against_the_oracle():
    """BASELINE config 4's scene on one GPU, the bench configuration itself (side-drop dam break at 256^3, 12.4 M particles,
    1.56 M rows), stage by stage against the 16-thread oracle at full size, bit for bit: binning, the P2G + density gather,
    cell labels and row numbers, face fractions, and both assembled systems (stencil, diagonal, right-hand side).  The
    solves are left to the property test: the oracle's Jacobi CG needs ~1200 iterations of 2 ms each per system here."""
    sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
    rng = np.random.default_rng(31)
    sc.vel = rng.normal(0.0, 0.15, sc.pos.shape)
    dt = 0.001
    o = Oracle(sc.desc(), threads=16)
    cell_id, cell_start, perm, _ = o.bin_particles(sc.pos)
    band2 = o.particle_mask(cell_start, 2)
    want_uvw = o.p2g(sc.pos, sc.vel, cell_start, perm, band2)
    want_density, _ = o.density(sc.pos, sc.mass, cell_start, perm, band2, sc.phi_fluid)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        assert ctx.bin_particles() == 0
        got_id, got_start, got_perm = ctx.download_binning()
        assert np.array_equal(got_id, cell_id) and np.array_equal(got_start, cell_start) and np.array_equal(got_perm, perm)
        del got_id, got_perm, cell_id
        ctx.p2g()
        for name, w in zip(("vel_u", "vel_v", "vel_w"), want_uvw):
            assert np.array_equal(ctx.download(name), w), name
        assert np.array_equal(ctx.download("density"), want_density)
        del want_uvw
        ctx.mark_valid_from_active(); ctx.extrapolate(2); ctx.update_prev(); ctx.add_forces(dt)
        n_rows = ctx.index_cells(); ctx.face_fractions(); ctx.density()
        uvw = [ctx.download(nm) for nm in ("vel_u", "vel_v", "vel_w")]
        z32 = lambda: np.zeros(o.shape, np.float32)  # noqa: E731
        label, index, want_rows = o.index_cells(o.particle_mask(cell_start, 1), sc.phi_fluid, sc.phi_solid, [z32(), z32(), z32()])
        assert n_rows == want_rows and n_rows > 1500000
        assert np.array_equal(ctx.download("label"), label) and np.array_equal(ctx.download("row_index"), index)
        wm = o.weight_mask(sc.phi_fluid, sc.phi_fluid_bg)
        fw = o.face_fractions(sc.phi_fluid, sc.phi_fluid_bg, wm)
        w = o.face_fractions(sc.phi_solid, sc.phi_solid_bg, wm)
        for name, want in zip(("weight_u", "weight_v", "weight_w"), w):
            assert np.array_equal(ctx.download(name), want), name
        for name, want in zip(("fluid_weight_u", "fluid_weight_v", "fluid_weight_w"), fw):
            assert np.array_equal(ctx.download(name), want), name
        off, diag, rhs = o.build_system(0, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw)
        ctx.build_system(SOLVE_DIVERGENCE, dt)
        goff, gdiag, grhs = ctx.download_system()
        assert np.array_equal(gdiag, diag) and np.array_equal(goff, off) and np.array_equal(grhs, rhs)
        del off, goff
        off2, diag2, rhs2 = o.build_system(1, dt, label, sc.phi_fluid, sc.phi_solid, w, density=want_density)
        ctx.build_system(SOLVE_DENSITY, dt)
        goff, gdiag, grhs = ctx.download_system()
        assert np.array_equal(gdiag, diag2) and np.array_equal(goff, off2) and np.array_equal(grhs, rhs2)


def test_config_dam_break_256_first_substep_against_the_oracle():
    """The whole hot-path substep of the bench configuration at full size against the 16-thread oracle, solves included
    (the oracle runs the reference's Jacobi CG to the same 1e-15: about 1200 iterations per system; the device its
    multigrid PCG, both solves side by side): pressure of the density solve on the labelled cells, projected velocity and
    displacement grids, and the particles' new positions and velocities, at the bars of the small-size tests."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
    rng = np.random.default_rng(31)
    sc.vel = rng.normal(0.0, 0.15, sc.pos.shape)
    dt = 0.001
    o = Oracle(sc.desc(), threads=16)
    ref = o.substep(fresh_state(sc, o), dt)
    assert ref["n_rows"] > 1500000
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        info = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True).step(dt)
        for k in ("divergence", "density"):
            assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL and info[k]["rows"] == ref["n_rows"], info[k]
        lab = ref["label"].astype(bool)
        assert rel_linf(ctx.download("pressure")[lab], ref["p_div"]) < 1e-8
        ctx.select_system(SOLVE_DENSITY)
        assert rel_linf(ctx.download("pressure")[lab], ref["p_den"]) < 1e-8
        ctx.select_system(SOLVE_DIVERGENCE)
        for name, want in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
            assert np.abs(ctx.download(name) - want).max() < POS_ATOL, name
        pos, vel = ctx.download_particles()
    assert np.abs(pos - ref["new_pos"]).max() < POS_ATOL
    assert rel_linf(vel, ref["new_vel"]) < VEL_RTOL


def test_config_cube_drop_128_first_substep_against_the_oracle():
    """BASELINE config 2 at full size (cube drop at 128^3, 2.1 M particles): the whole hot-path substep against the
    16-thread oracle, solves included (Jacobi CG to 1e-15 there, multigrid PCG with both solves side by side here), at the
    bars of the small-size tests."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.cube_drop(128, ppc=8, seed=12345, exact_levelset=False)
    rng = np.random.default_rng(3)
    sc.vel = rng.normal(0.0, 0.2, sc.pos.shape)
    dt = 0.002
    o = Oracle(sc.desc(), threads=16)
    ref = o.substep(fresh_state(sc, o), dt)
    assert ref["n_rows"] > 200000 and ref["err_div"] < RES_TOL and ref["err_den"] < RES_TOL
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        info = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True).step(dt)
        for k in ("divergence", "density"):
            assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL and info[k]["rows"] == ref["n_rows"], info[k]
        lab = ref["label"].astype(bool)
        assert rel_linf(ctx.download("pressure")[lab], ref["p_div"]) < 1e-8
        ctx.select_system(SOLVE_DENSITY)
        assert rel_linf(ctx.download("pressure")[lab], ref["p_den"]) < 1e-8
        ctx.select_system(SOLVE_DIVERGENCE)
        for name, want in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
            assert np.abs(ctx.download(name) - want).max() < POS_ATOL, name
        pos, vel = ctx.download_particles()
    assert np.abs(pos - ref["new_pos"]).max() < POS_ATOL
    assert rel_linf(vel, ref["new_vel"]) < VEL_RTOL


def test_config_floating_ball_128_first_substep_against_the_oracle():
    """BASELINE config 3 at full size (config 2 plus the half-density sphere, two-way coupled through J^T M^-1 J,
    src/FluidSimulator.cpp:810-816): the whole hot-path substep against the 16-thread oracle, the rank-6 coupled
    divergence solve and the density solve included -- pressures on the labelled cells, displacement grids, the particles
    and the body's new velocity.  The particles inside the sphere are taken out first, as the reference's advection keeps
    them outside solids (cells walled in by the body would otherwise form pockets whose pressure level is arbitrary)."""
    from fluidsimulation_amd import FlipSubstep, PRECOND_MULTIGRID
    sc = scenes.cube_drop(128, ppc=8, seed=12345, exact_levelset=False)
    centre, radius_world = (0.5, 0.45, 0.5), 0.1
    scenes.add_sphere(sc, radius_world=radius_world, centre_world=centre, density=500.0, vel=(0.05, -0.3, 0.02, 0.4, 0.0, -0.2))
    dx = float(sc.dx)
    keep = np.linalg.norm((sc.pos + 0.5) * dx - np.array(centre), axis=1) > radius_world + 0.5 * dx
    assert 1000 < (~keep).sum() < keep.sum()
    sc.pos, sc.vel, sc.mass = sc.pos[keep].copy(), sc.vel[keep].copy(), sc.mass[keep].copy()
    sc.phi_fluid = scenes.fluid_levelset(sc, sc.pos)
    rng = np.random.default_rng(2)
    sc.vel = rng.normal(0.0, 0.1, sc.pos.shape)
    dt = 0.002
    o = Oracle(sc.desc(), threads=16)
    st = fresh_state(sc, o)
    st["obj"] = dict(phi=sc.phi_object, bg=sc.phi_object_bg, com=sc.object_com, minv=sc.object_minv, vel=sc.object_vel)
    ref = o.substep(st, dt)
    assert ref["n_rows"] > 200000 and ref["err_div"] < RES_TOL and ref["err_den"] < RES_TOL
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        info = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, rigid_body=True).step(dt)
        for k in ("divergence", "density"):
            assert info[k]["converged"] == 1 and info[k]["rel_residual_2"] < RES_TOL and info[k]["rows"] == ref["n_rows"], info[k]
        lab = ref["label"].astype(bool)
        assert rel_linf(ctx.download("pressure")[lab], ref["p_den"]) < 1e-8      # the density system was solved last
        assert rel_linf(ctx.solid_object_velocity(), ref["obj_vel"]) < 1e-8
        for name, want in zip(("disp_x", "disp_y", "disp_z"), ref["disp"]):
            assert np.abs(ctx.download(name) - want).max() < POS_ATOL, name
        pos, vel = ctx.download_particles()
    assert np.abs(pos - ref["new_pos"]).max() < POS_ATOL
    assert rel_linf(vel, ref["new_vel"]) < VEL_RTOL
