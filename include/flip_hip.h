/*
 * flip_hip.h -- C ABI of the MI355X (gfx950) FLIP hot-path library, libflip_hip.so.
 *
 * The reference (PanagiotisP/FluidSimulation) has no plugin or FFI layer: its boundary is the
 * C++ class API FluidSimulator / FluidDomain / MacGrid / ParticleSet.  The host classes in
 * fluidsimulation_amd/host/ keep those signatures and implement their bodies with the entry
 * points below; every entry point cites the reference member function whose body it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - All state lives in an opaque context bound to one HIP device and one stream.  Calls are
 *     stream-ordered; only the *_download calls, flip_synchronize and the functions that
 *     return a host scalar wait for the device.  One context per host thread.
 *   - Every field is a dense array over one axis-aligned box of cells, x fastest:
 *         id(i,j,k) = i + nx*(j + ny*k),  index-space coordinate of (i,j,k) = (ox+i,oy+j,oz+k).
 *     The box stands for the active bounding box of the reference's solid level set
 *     (src/FluidSimulator.cpp:205,1040,1122).  A read outside the box returns the field's
 *     background value, as an OpenVDB accessor does for an unallocated voxel.
 *   - Particle positions are index-space doubles (headers/ParticleSet.h:36-40), passed as
 *     [n][3] arrays; velocities likewise; mass is float.
 *   - Return value 0 = success; anything else is an error whose text flip_last_error returns.
 *     The library never falls back to a CPU path: without a usable HIP device
 *     flip_ctx_create fails.
 */
#ifndef FLIP_HIP_H
#define FLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flip_ctx flip_ctx;

typedef struct flip_grid_desc {
    int32_t nx, ny, nz;   /* extent of the box in cells                                     */
    int32_t ox, oy, oz;   /* index-space coordinate of local cell (0,0,0)                   */
    float dx;             /* voxel size in world units (main.cpp:106)                       */
    float phi_fluid_bg;   /* fluid level-set background (+3 dx for createLevelSet)          */
    float phi_solid_bg;   /* solid level-set value outside the box                          */
    float rest_density;   /* FluidDomain::density (src/FluidDomain.cpp:12)                  */
} flip_grid_desc;

/* Dense fields that can be uploaded / downloaded.  Element type in brackets. */
enum flip_field {
    FLIP_F_PHI_FLUID = 0,   /* [f32]  FluidDomain::fluidLevelSet()                         */
    FLIP_F_PHI_SOLID = 1,   /* [f32]  FluidDomain::solidLevelSet()                         */
    FLIP_F_VEL_U = 2,       /* [f64]  MacGrid::velFront(), x component (stored at i-1/2)   */
    FLIP_F_VEL_V = 3,       /* [f64]                         y component                   */
    FLIP_F_VEL_W = 4,       /* [f64]                         z component                   */
    FLIP_F_PREV_U = 5,      /* [f64]  MacGrid::velPrev()                                   */
    FLIP_F_PREV_V = 6,
    FLIP_F_PREV_W = 7,
    FLIP_F_DIFF_U = 8,      /* [f64]  MacGrid::velDiff()                                   */
    FLIP_F_DIFF_V = 9,
    FLIP_F_DIFF_W = 10,
    FLIP_F_DISP_X = 11,     /* [f64]  MacGrid::displacementGrid()                          */
    FLIP_F_DISP_Y = 12,
    FLIP_F_DISP_Z = 13,
    FLIP_F_WEIGHT_U = 14,   /* [f32]  MacGrid::uWeights() (solid face fractions)           */
    FLIP_F_WEIGHT_V = 15,
    FLIP_F_WEIGHT_W = 16,
    FLIP_F_FLUID_WEIGHT_U = 17, /* [f32]  MacGrid::uFluidWeights()                         */
    FLIP_F_FLUID_WEIGHT_V = 18,
    FLIP_F_FLUID_WEIGHT_W = 19,
    FLIP_F_DENSITY = 20,    /* [f32]  MacGrid::densityGrid()                               */
    FLIP_F_PRESSURE = 21,   /* [f64]  solution of the last pressure solve, 0 off the rows  */
    FLIP_F_VALID_U = 22,    /* [u8]   MacGrid::validUFront()                               */
    FLIP_F_VALID_V = 23,
    FLIP_F_VALID_W = 24,
    FLIP_F_ACTIVE = 25,     /* [u8]   active voxels of velFront                            */
    FLIP_F_LABEL = 26,      /* [u8]   1 where the cell has a pressure row                  */
    FLIP_F_ROW_INDEX = 27,  /* [i32]  row number (ascending cell id) or -1                 */
    FLIP_F_BAND = 28,       /* [u8]   bit0: particle cells dilated 1, bit1: dilated 2      */
    FLIP_F_OBJ_WEIGHT_U = 29, /* [f32] SolidObject::uWeights() of the coupled rigid body   */
    FLIP_F_OBJ_WEIGHT_V = 30,
    FLIP_F_OBJ_WEIGHT_W = 31,
    FLIP_F_PHI_OBJECT = 32, /* [f32]  SolidObject::levelSet() of the coupled rigid body    */
    FLIP_F_VEL_BACK_U = 33, /* [f64]  MacGrid::velBack(), component-wise (headers/MacGrid.h:48) */
    FLIP_F_VEL_BACK_V = 34,
    FLIP_F_VEL_BACK_W = 35,
    FLIP_F_VALID_BACK_U = 36, /* [u8] MacGrid::validUBack() / VBack / WBack (headers/MacGrid.h:54-58) */
    FLIP_F_VALID_BACK_V = 37,
    FLIP_F_VALID_BACK_W = 38,
    FLIP_F_COUNT = 39
};

enum flip_solve_kind {
    FLIP_SOLVE_DIVERGENCE = 0, /* compute_pressure_divirgence_constraint                   */
    FLIP_SOLVE_DENSITY = 1     /* compute_pressure_density_constraint                      */
};

enum flip_precond {
    FLIP_PRECOND_JACOBI = 0,   /* Eigen DiagonalPreconditioner: what the reference runs    */
    FLIP_PRECOND_MULTIGRID = 1 /* geometric multigrid V-cycle (GPU-parallel replacement of  */
                               /* the sequential MIC(0) of headers/pcgsolver/pcg_solver.h)  */
};

typedef struct flip_pcg_info {
    int32_t converged;      /* 1 when |r|^2 < tol^2 |b|^2 was reached                      */
    int32_t iterations;     /* number of operator applications                             */
    int32_t rows;           /* system size                                                 */
    int32_t cg_rows;        /* rows in the CG iteration (rows - dependent_rows)            */
    int32_t active_chunks;  /* 256-cell chunks swept per kernel                            */
    double rel_residual_2;  /* sqrt(|r|^2 / |b|^2), r the recurrence residual (as Eigen)   */
    double rhs_norm_2;      /* |b|                                                         */
    double seconds;         /* device time of the solve (HIP events on the context stream) */
    double spmv_seconds;    /* device time spent in the operator kernel, when profiled     */
    int32_t spmv_launches;  /* launches covered by spmv_seconds                            */
    int32_t dependent_rows; /* rows solved by substitution after CG (see flip_set_system_options) */
    int32_t iterations_inf; /* iterations after which max|r| <= 1e-12 max|b| first held, the stopping rule of  */
                            /* headers/pcgsolver/pcg_solver.h:241-258 (-1: never, or not tracked: slab solves) */
    double rel_residual_inf; /* max|r| / max|b| when the solve stopped                                         */
} flip_pcg_info;

/* ---- context ---------------------------------------------------------------------------- */

/* Replaces the construction of FluidDomain/MacGrid storage (src/MacGrid.cpp:3-51,
 * src/FluidDomain.cpp:14-22).  `stream` is a hipStream_t (NULL = the device's null stream). */
int flip_ctx_create(const flip_grid_desc *desc, int device, void *stream, flip_ctx **out);
void flip_ctx_destroy(flip_ctx *ctx);
/* Text of the last error on this context (or of the last failed flip_ctx_create if ctx==NULL). */
const char *flip_last_error(const flip_ctx *ctx);
int flip_synchronize(flip_ctx *ctx);
int flip_get_desc(const flip_ctx *ctx, flip_grid_desc *out);
/* The hipStream_t every call on this context enqueues its work on, and the context's device: for callers that order
 * their own device work (communication, include/flip_dist.h) between the library's kernels. */
int flip_get_stream(const flip_ctx *ctx, void **stream, int *device);

/* ---- particles: ParticleSet (headers/ParticleSet.h:43-116) --------------------------------- */

/* ParticleSet::addParticle for a whole set: replaces the std::vector<Particle> store. */
int flip_particles_upload(flip_ctx *ctx, int64_t n, const double *pos, const double *vel, const float *mass);
/* ParticleSet::operator[] for a whole set, in upload order.  Any pointer may be NULL. */
int flip_particles_download(flip_ctx *ctx, double *pos, double *vel);
/* Particle::mass() for a whole set, in the order of flip_particles_download.  After a particle exchange of the slab
 * decomposition the set is not the uploaded one any more: count, then positions, velocities and masses describe it. */
int flip_particles_download_mass(flip_ctx *ctx, float *mass);
int64_t flip_particles_count(const flip_ctx *ctx);

/* Replaces ParticleAtlas::create + createPointMask + dilateActiveValues
 * (src/FluidSimulator.cpp:83-90, 281-283, 628-630, 686-688, 864-866): counting sort of the
 * particles by cell floor(pos+0.5), stable in upload order, plus the 1- and 2-cell dilated
 * occupancy masks (FLIP_F_BAND).  n_outside (may be NULL) receives the number of particles
 * outside the box; they take part in nothing until they return. */
int flip_bin_particles(flip_ctx *ctx, int64_t *n_outside);
/* Parity hook: cell_id[n] in upload order (-1 outside), cell_start[B+1], perm[n] (upload index
 * of the particle in each sorted slot).  Any pointer may be NULL. */
int flip_binning_download(flip_ctx *ctx, int32_t *cell_id, int32_t *cell_start, int32_t *perm);

/* ---- dense fields: MacGrid getters/setters (headers/MacGrid.h:47-66) ----------------------- */

int flip_field_upload(flip_ctx *ctx, int field, const void *host);
int flip_field_download(flip_ctx *ctx, int field, void *host);
int64_t flip_field_bytes(const flip_ctx *ctx, int field);

/* ---- transfers -------------------------------------------------------------------------------- */

/* FluidSimulator::transfer_from_particles_to_grid (src/FluidSimulator.cpp:71-118) with
 * GatherTransfer::operator() (src/Functors.cpp:10-45).  Needs flip_bin_particles first.
 * Result in velFront; FLIP_F_ACTIVE = particle cells dilated 2.  The same walk over the particles also
 * leaves FLIP_F_DENSITY up to date for the current level set (see flip_density). */
int flip_p2g(flip_ctx *ctx);
/* Diagnostic of the LDS-staged gather behind flip_p2g / flip_density (no counterpart in the reference): tiles of
 * 4 x 4 x 4 cells processed since the context was created, out[0] with their particle records staged in LDS, out[1]
 * walked from global memory because the tile's support held more records than the staging area (crowded cells). */
int flip_gather_stats(flip_ctx *ctx, int64_t out[2]);
/* Selects the form of the gather behind flip_p2g / flip_density (no counterpart in the reference; both forms give
 * the same bits).  0 (default): every lane walks its cells' particle records in global memory.  1: LDS-staged -- a
 * workgroup copies the records around a tile of 4 x 4 x 4 cells into LDS once and the cells walk them there. */
int flip_set_gather_mode(flip_ctx *ctx, int mode);
/* FluidSimulator::transfer_from_grid_to_particles (src/FluidSimulator.cpp:120-135). */
int flip_g2p(flip_ctx *ctx, float flip_pic_ratio);
/* FluidSimulator::apply_particle_displacements (src/FluidSimulator.cpp:137-151). */
int flip_apply_displacements(flip_ctx *ctx);

/* The three particle updates that close a substep in one pass over the particles: flip_g2p(flip_pic_ratio), then, if
 * advect != 0, flip_advect_particles(dt), then, if displace != 0, flip_apply_displacements -- the order of
 * src/FluidSimulator.cpp:420, :426, :429.  Same result as the three calls, bit for bit; one read and one write of every
 * position instead of three. */
int flip_update_particles(flip_ctx *ctx, float flip_pic_ratio, float dt, int advect, int displace);

/* FluidDomain::advectParticles (src/FluidDomain.cpp:63-75) with ParticleSet::advectAndEnsureOutsideObstacles
 * (src/ParticleSet.cpp:18-45): RK3 through velFront, then particles inside the container wall or the rigid
 * body are moved to its surface along the level-set gradient (the reference samples an OpenVDB closest-point
 * grid; for a signed-distance level set the two agree). */
int flip_advect_particles(flip_ctx *ctx, float dt);

/* ---- grid-side glue ------------------------------------------------------------------------- */

/* validU/V/WFront |= topology of velFront (src/FluidSimulator.cpp:343-345). */
int flip_mark_valid_from_active(flip_ctx *ctx);
/* FluidSimulator::extrapolate_data (src/FluidSimulator.cpp:181-265). */
int flip_extrapolate(flip_ctx *ctx, int iterations);
/* FluidDomain::constructFluidLevelSetFromMarkerParticles (src/FluidDomain.cpp:50-61) with LevelSet::unionOfBalls
 * (src/LevelSet.cpp:17-23): FLIP_F_PHI_FLUID = distance to the nearest particle minus radius_world, clipped to
 * +-half_width_voxels voxels, merged with the solid level set (headers/LevelSet.h:83-96).  Exact distances; the
 * reference's OpenVDB rasterisation is not available to compare with.  Needs flip_bin_particles first. */
int flip_fluid_levelset_from_particles(flip_ctx *ctx, float radius_world, float half_width_voxels);
/* FluidSimulator::constrain_velocity (src/FluidSimulator.cpp:1180-1247; rigid_body != 0: the overload for the
 * attached SolidObject, :1249-1321, with the velocity last given to flip_solid_object_set or updated by
 * flip_project_velocity).  Uses the face fractions currently stored, as the reference does. */
int flip_constrain_velocity(flip_ctx *ctx, int rigid_body);
/* FluidSimulator::add_forces, fluid part (src/FluidSimulator.cpp:460-473). */
int flip_add_forces(flip_ctx *ctx, float dt);
/* MacGrid::updatePreviousVelocityBuffer / updateDiffBuffers (src/MacGrid.cpp:55-63). */
int flip_update_prev(flip_ctx *ctx);
int flip_update_diff(flip_ctx *ctx);
/* MacGrid::swapVelocityBuffers / swapValidMasks (src/MacGrid.cpp:65-70): pointer swaps of the front and back buffers. */
int flip_swap_velocity_buffers(flip_ctx *ctx);
int flip_swap_valid_masks(flip_ctx *ctx);
/* max over the box of each velocity component (Grid::evalMinMax in compute_cfl,
 * src/FluidSimulator.cpp:448-458). */
int flip_velocity_max(flip_ctx *ctx, double out_max[3]);
/* flip_velocity_max for callers that cannot allocate.  No counterpart in the reference; it exists for tight serving
 * loops and harnesses that forbid hipMalloc: this call performs no device or host allocation of any kind.  It still
 * ends with a device-to-host copy of the three doubles and a stream synchronise, so it cannot be recorded into a HIP
 * graph.  Its per-workgroup partial maxima live in a static device array of 96 KB that every GPU the library runs on
 * holds until the process ends.  Same result as flip_velocity_max, bit for bit.  Limit: boxes of at most 67 108 864
 * cells (4096 workgroups x 16384 cells); larger ones are refused with an error.  All calls on one GPU share that GPU's
 * array: calls that overlap in time on one GPU (different contexts or streams) race. */
int flip_velocity_max_static(flip_ctx *ctx, double out_max[3]);

/* ---- pressure-solve preparation (src/FluidSimulator.cpp:482-492) --------------------------- */

/* FluidSimulator::index_fluid_cells (src/FluidSimulator.cpp:267-305).  Uses the weights
 * currently stored (the reference calls it before compute_face_fractions). */
int flip_index_cells(flip_ctx *ctx, int32_t *n_rows);
/* FluidSimulator::compute_face_fractions (src/FluidSimulator.cpp:527-623) and, when a rigid
 * body is attached, SolidObject::compute_face_fractions (src/SolidObject.cpp:103-152). */
int flip_face_fractions(flip_ctx *ctx);
/* FluidSimulator::compute_density (src/FluidSimulator.cpp:625-652) + DensityCalculator
 * (src/Functors.cpp:216-266).  n_overdense (may be NULL) = cells above 1.5 rest density, the
 * cells whose particles the reference re-seeds at random (src/Functors.cpp:277-322).  flip_p2g already
 * computes the field; this call only launches a kernel when particles, sort or level set changed since. */
int flip_density(flip_ctx *ctx, int64_t *n_overdense);

/* The side effect of DensityCalculator on the particles (src/Functors.cpp:268-322): the particles of every cell whose
 * density exceeds 1.5 rest density are spread over a shuffled lattice of sub-cells of that cell, each at a uniform random
 * offset.  The reference draws from a random_device-seeded generator shared by its TBB threads and does not reproduce
 * itself; every draw here is headers/util.h's randhash of (seed, cell, draw number), so equal inputs and seed give equal
 * positions.  Call after flip_p2g / flip_density of the same substep; n_cells / n_particles (may be NULL) receive how
 * many cells and particles were touched.  Off unless called: the reference's own results differ run to run here. */
int flip_redistribute_overdense(flip_ctx *ctx, uint32_t seed, int64_t *n_cells, int64_t *n_particles);

/* Attach (phi_object != NULL) or detach the rigid body that is strongly coupled in the
 * divergence solve: SolidObject (headers/SolidObject.h:9-62).  com = centre of mass in index
 * space, minv = diagonal of SolidObject::massMatrixInverse, vel = combinedVelocityVector. */
int flip_solid_object_set(flip_ctx *ctx, const float *phi_object, float phi_object_bg, const double com[3],
                          const double minv[6], const double vel[6]);
/* SolidObject::combinedVelocityVector after project_divirgence_constraint (:1095-1100). */
int flip_solid_object_velocity(flip_ctx *ctx, double vel[6]);

/* ---- solves and projections --------------------------------------------------------------- */

/* Assembly part of compute_pressure_{divirgence,density}_constraint
 * (src/FluidSimulator.cpp:654-816, 832-990) in matrix-free form. */
int flip_build_system(flip_ctx *ctx, int kind, float dt);
/* Two properties of the reference's system decide whether its CG can converge at all; both options default
 * to 1 and apply to the next flip_build_system.
 *   drop_null_rows: a labelled cell with no liquid or empty neighbour behind an open face (e.g. on an edge of
 *     the container) has an all-zero row.  With a non-zero right-hand side the system is inconsistent and CG as
 *     the reference runs it diverges until the iteration cap.  1 forces the right-hand side of such rows to
 *     zero, which leaves every other unknown's exact solution unchanged and is what headers/pcgsolver does
 *     with a null row (pcg_solver.h:121); 0 keeps the reference's behaviour.
 *   eliminate_dependent_rows: a labelled cell with phi_fluid >= 0 (labelled through the partial-face rule of
 *     src/FluidSimulator.cpp:297-300) refers to its liquid neighbours, but no row refers to it, so the matrix
 *     is block lower triangular and not symmetric.  1 runs CG on the symmetric liquid block and obtains the
 *     dependent rows by substitution: the same solution of the same A p = b, reached reliably.  0 runs CG on
 *     the whole non-symmetric matrix as the reference does.  With a rigid body whose coupling rows touch a
 *     dependent row the library solves the whole system regardless. */
int flip_set_system_options(flip_ctx *ctx, int drop_null_rows, int eliminate_dependent_rows);
/* Warm start of the solves (default 0: x0 = 0, as the reference's Eigen solve() starts, src/FluidSimulator.cpp:823,
 * :1007).  With 1 a solve starts from the converged solution of the previous system of the same kind on this context
 * instead (Eigen's solveWithGuess), masked to the rows of the new system.  The stopping rule |r| < tol |b| is the same,
 * so the result meets the same bound; only the iteration count changes -- and at the reference's tolerance of 1e-15 it
 * hardly does (50.6 against 50.75 iterations per substep on the 256^3 dam break: the rows that change at the free
 * surface leave a first residual of the order of |b|), which is why it is off.  Applies to the symmetric system
 * without a rigid body; every other solve starts from zero whatever this says. */
int flip_set_warm_start(flip_ctx *ctx, int enabled);
/* Parity hook: the assembled stencil on the box: off[6][B] (faces -x,+x,-y,+y,-z,+z), diag[B],
 * rhs[B]; zero off the rows.  Any pointer may be NULL. */
int flip_system_download(flip_ctx *ctx, double *off, double *diag, double *rhs);
/* Solver part (src/FluidSimulator.cpp:822-823, 1006-1007): conjugate gradients from x0 = 0
 * until |r|^2 < tol^2 |b|^2 or max_iter operator applications (max_iter <= 0: 2*rows, the
 * Eigen default).  Result in FLIP_F_PRESSURE. */
int flip_pcg_solve(flip_ctx *ctx, double tol, int32_t max_iter, int precond, flip_pcg_info *info);
/* Both solves of a substep at once (src/FluidSimulator.cpp:505 and :521).  Neither system depends on the other's
 * solution (the density system reads labels, level sets, face fractions and the density field only), so this call
 * assembles both, iterates on them side by side on two internal streams and returns when both have stopped; each solve
 * performs the same operations as flip_build_system + flip_pcg_solve.  A context then holds two assembled systems:
 * flip_select_system(ctx, kind) makes one of them the active one, which flip_project_velocity (FLIP_SOLVE_DIVERGENCE),
 * flip_project_displacement (FLIP_SOLVE_DENSITY), FLIP_F_PRESSURE and flip_system_download refer to.  On return the
 * divergence system is active. */
int flip_solve_pair(flip_ctx *ctx, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info_divergence,
                    flip_pcg_info *info_density);
int flip_select_system(flip_ctx *ctx, int kind);
/* Two systems in progress under the caller's control (no counterpart in the reference; the slab driver of
 * include/flip_dist.h interleaves the two solves of a substep with these).  Between flip_pair_begin and flip_pair_end
 * the context holds two system sets, each with a stream of its own; flip_pair_select(slot 0 | 1) makes one the active
 * set, to which every call then refers: flip_build_system, flip_pcg_dist_*, flip_get_stream (the set's stream), the
 * projections.  Work enqueued before begin is ordered before both streams; after end the context's own stream is ordered
 * after both and slot 0 is the active set.  flip_pcg_solve and flip_solve_pair manage the sets themselves and are
 * refused in between. */
int flip_pair_begin(flip_ctx *ctx);
int flip_pair_select(flip_ctx *ctx, int slot);
int flip_pair_end(flip_ctx *ctx);
/* A second stream inside a substep (no counterpart in the reference, whose stages run one after the other on the
 * CPU's thread pool).  Two chains of src/FluidSimulator.cpp:307-446 do not depend on each other once the fluid level
 * set exists: (a) flip_p2g, flip_mark_valid_from_active, flip_extrapolate, flip_constrain_velocity, flip_update_prev,
 * flip_add_forces (:337-:373; they read the level sets, the particles and the face fractions of the PREVIOUS substep and
 * write velocities, masks and the density) and (b) flip_index_cells, flip_face_fractions (:486-:488; they read the
 * level sets and the previous fractions and write labels, row indices and the new fractions); after the solves,
 * flip_project_displacement is independent of flip_project_velocity, flip_extrapolate, flip_constrain_velocity and
 * flip_update_diff.  flip_side_fork orders the side stream after everything enqueued on the context's stream so far;
 * between flip_side_select(ctx, 1) and flip_side_select(ctx, 0) every entry point enqueues on the side stream;
 * flip_side_join orders the context's stream after the side work and selects it.  flip_face_fractions writes the
 * other buffer of a double-buffered pair, so on whichever stream it runs, flip_constrain_velocity and flip_index_cells
 * calls made BEFORE it (in host order) read the previous substep's fractions, as the reference's order has it.  The
 * caller is responsible for putting only independent stages on the two streams.  flip_set_side_stream(ctx, 0) turns
 * the three calls into no-ops (everything stays on the one stream; the default is on).  Results are the same bits
 * either way. */
int flip_set_side_stream(flip_ctx *ctx, int enabled);
int flip_side_fork(flip_ctx *ctx);
int flip_side_select(flip_ctx *ctx, int side);
int flip_side_join(flip_ctx *ctx);
/* Exclusive prefix sum of n int32 values in place (parity hook for the scan behind the binning, the labels, the row
 * lists and the multigrid levels; no counterpart in the reference, which keeps sparse trees).  The values are placed
 * `offset` (0..3) items into a device buffer, so that the path for data off a 16-byte boundary can be checked too. */
int flip_scan_i32(flip_ctx *ctx, int32_t *host_values, int64_t n, int32_t offset);
/* y = A x for the assembled system on dense box vectors (parity hook for the SpMV kernel). */
int flip_apply_operator(flip_ctx *ctx, const double *x_host, double *y_host);
/* When enabled, flip_pcg_solve brackets a sample of its operator launches with HIP events and
 * reports their total in flip_pcg_info::spmv_seconds. */
int flip_set_profiling(flip_ctx *ctx, int enabled);
/* With profiling enabled, the first bracketed iteration of a multigrid solve also has each of its level-0 kernels
 * timed on its own (HIP events on the solve's stream, nothing else running on the device in flip_solve_pair).
 * us[0] operator + direction kernel, us[1] update + pre-smoothing, us[2] level-0 restriction, us[3] all coarse
 * levels together, us[4] level-0 correction + first post-smoothing step, us[5] second post-smoothing step (+ r.z).
 * Measurement aid; no counterpart in the reference (its timings are wall-clock printfs, src/FluidSimulator.cpp:437). */
int flip_solver_kernel_times(flip_ctx *ctx, double us[6]);

/* ---- z-slab decomposition of the solve over several GPUs (one process and one context per GPU) ----------
 * No counterpart in the reference, which is a single-process CPU program.  Each rank's box is its slab of
 * the global box plus ghost planes on either side; flip_set_slab names the planes [k0, k1) of the box whose
 * rows this context owns (the default is the whole box).  It applies to the next flip_build_system.
 *
 * The solve is split into phases because two things cross ranks and the library does no communication of
 * its own: the caller all-reduces (sum) buffer 0 (16 doubles) after begin(), after every direction() and
 * after every update(); and after begin() and every update() it sends planes k0 and k1-1 of buffer 1 (the
 * preconditioned residual, B doubles, plane k at offset k*nx*ny) to the lower / upper neighbour and receives
 * their planes into k0-1 and k1.  Iterate direction(), update() until flip_pcg_dist_status reports done != 0
 * (1 converged, 2 iteration cap; identical on every rank), exchange the ghost planes of buffer 2 (the
 * solution), and call finish(), which also fills FLIP_F_PRESSURE on the dependent rows.  The preconditioner
 * is applied per slab (block Jacobi over ranks with a V-cycle or point Jacobi inside each block). */
int flip_set_slab(flip_ctx *ctx, int32_t k0, int32_t k1);
/* The strongly coupled rigid body under slabs (src/FluidSimulator.cpp:810-816, 1095-1100): every rank attaches the same
 * body; the six sums J p of the coupling term travel in buffer 0 with p.Ap, so the solve needs no further call.  After
 * the solve and flip_project_velocity (which leaves the body alone on a slab context), flip_solid_coupling_partial
 * returns this context's share of J p (rows of its owned planes); the caller sums it over the ranks and passes the total
 * to flip_solid_velocity_apply on every rank: V += dt Minv (J p). */
int flip_solid_coupling_partial(flip_ctx *ctx, double jp[6]);
/* Dependent rows (cells with phi_fluid >= 0 that carry a row) with a rigid-body entry found by the last
 * flip_build_system.  Such rows rule out the elimination of the dependent rows (flip_set_system_options): a single
 * context then solves the whole system; a slab context keeps the eliminated form, which the slab solver needs, and the
 * ranks must agree on the sum of this count before they start the solve (flip_dist_solve does). */
int flip_system_dependent_coupled(flip_ctx *ctx, int32_t *count);
int flip_solid_velocity_apply(flip_ctx *ctx, float dt, const double jp_total[6]);
int flip_pcg_dist_buffer(flip_ctx *ctx, int which, void **device_ptr, int64_t *count);
int flip_pcg_dist_begin(flip_ctx *ctx, double tol, int32_t max_iter, int precond);
int flip_pcg_dist_direction(flip_ctx *ctx);
int flip_pcg_dist_update(flip_ctx *ctx);
int flip_pcg_dist_status(flip_ctx *ctx, int32_t *done, int32_t *iterations);
int flip_pcg_dist_finish(flip_ctx *ctx, flip_pcg_info *info);
/* Overlapping variant of the slab preconditioner (additive Schwarz; no counterpart in the reference).  With
 * flip_set_slab_overlap(ctx, m), 1 <= m <= ghost planes, the local preconditioner covers planes [k0-m, k1+m): the
 * residual on the overlap planes comes from the neighbour that owns them and the correction computed there is added to
 * that neighbour's.  It stays symmetric positive definite and needs far fewer CG iterations than disjoint blocks when
 * the slabs are many cells thick.  m = 0 restores the disjoint blocks.  Call it after flip_set_slab, with the same m on
 * every rank.  The phases change as follows: begin() and update() stop before the preconditioner; the caller then
 *   1. sends planes [k0, k0+m) and [k1-m, k1) of buffer 3 (the residual) to the lower / upper neighbour and receives
 *      theirs into [k0-m, k0) and [k1, k1+m);
 *   2. calls precondition();
 *   3. sends planes [k0-m, k0+1) and [k1-1, k1+m) of buffer 1 to the lower / upper neighbour and receives theirs into
 *      buffer 4 (from the lower neighbour) and buffer 5 (from the upper), (m+1)*nx*ny doubles each;
 *   4. calls combine() and all-reduces buffer 0.
 * The single-plane exchange of buffer 1 is not needed in this variant.  m must be smaller than the ghost thickness:
 * the end of the box acts as a wall in the local matrix.  Unless the environment variable FLIP_SLAB_WEIGHTED is 0 the
 * contributions are blended by weights that rise linearly across the 2m planes around a face (their square roots
 * before and after the local preconditioner), so that the overlap is not corrected twice. */
int flip_set_slab_overlap(flip_ctx *ctx, int32_t planes);
int flip_pcg_dist_precondition(flip_ctx *ctx);
int flip_pcg_dist_combine(flip_ctx *ctx);

/* Particle exchange of the slab decomposition.  A context is authoritative for the particles its last
 * flip_bin_particles sorted into cells of its owned planes.  flip_particles_pack copies those of them whose
 * cell plane (in the box's own plane numbering, which may fall outside 0..nz) now lies in
 * [plane_lo, plane_hi) into 64-byte records {x,y,z: f64; mass: f32; 4 bytes padding; vx,vy,vz: f64; 8 bytes padding} at
 * device_out, in slot order, and returns their number.  flip_particles_exchange_apply replaces the
 * particle set by the authoritative particles that are still inside the box followed by the records received
 * from the lower and the upper neighbour (device buffers); the set must be sorted again afterwards.
 * flip_particles_reserve sizes the particle arrays before the first upload so that the set can grow. */
int flip_particles_reserve(flip_ctx *ctx, int64_t capacity);
int flip_particles_pack(flip_ctx *ctx, int32_t plane_lo, int32_t plane_hi, void *device_out, int64_t capacity, int64_t *count);
int flip_particles_exchange_apply(flip_ctx *ctx, const void *recv_lo, int64_t n_lo, const void *recv_hi, int64_t n_hi);

/* FluidSimulator::project_divirgence_constraint (src/FluidSimulator.cpp:1017-1102). */
int flip_project_velocity(flip_ctx *ctx, float dt);
/* FluidSimulator::project_density_constraint (src/FluidSimulator.cpp:1104-1178). */
int flip_project_displacement(flip_ctx *ctx, float dt);

#ifdef __cplusplus
}
#endif
#endif /* FLIP_HIP_H */
