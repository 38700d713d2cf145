/* flip_dist.h -- C ABI of libflip_dist.so: the z-slab decomposition of the FLIP substep over several GPUs, one process
 * and one flip_ctx per GPU, communicating through RCCL (nccl API over xGMI) on the context's own stream.
 *
 * The reference is a single-process CPU program (its one orchestration point is
 * FluidSimulator::advance_flip_pic, src/FluidSimulator.cpp:307-446), so nothing here replaces a reference function
 * one to one: this library is that orchestration point for N contexts.  It calls libflip_hip.so only through
 * include/flip_hip.h.  What crosses ranks per substep:
 *   - per CG iteration two all-reduces of 16 doubles (p.Ap and the six rigid-body sums J p; r.r and r.z) and the plane
 *     exchanges of the slab preconditioner (flip_hip.h, "z-slab decomposition of the solve"); the two solves of a substep
 *     run side by side, each on a stream and a communicator of its own (flip_dist_solve_pair);
 *   - after the solve the ghost planes of the pressure;
 *   - after the particle update the particles near a slab face: one count message and one payload message per face
 *     (migration and the neighbours' ghost copies alike);
 *   - the maximum velocity of compute_cfl (src/FluidSimulator.cpp:448-458) and, with a rigid body, the six sums J p of
 *     its velocity update (:1095-1100).
 * Every collective and every send/receive is enqueued on the context's stream between the kernels that produce and
 * consume its buffers; the host waits only where the algorithm needs a value (the stop flag every 8 iterations, the
 * particle counts).
 *
 * With world == 1 the context is not put into slab mode and every call falls through to the single-context entry points
 * (flip_solve_pair for the two solves of a substep), bit for bit; FLIP_DIST_FORCE_SLAB makes a single rank take the slab
 * path anyway (its collectives run over a communicator of one rank), which is how the RCCL path is exercised on a box
 * with one GPU. */
#ifndef FLIP_DIST_H
#define FLIP_DIST_H

#include "flip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flip_dist flip_dist;

#define FLIP_DIST_UNIQUE_ID_BYTES 128
#define FLIP_DIST_FORCE_SLAB 1 /* flags of flip_dist_create*: take the slab path also with world == 1 */
#define FLIP_DIST_SEQUENTIAL_SOLVES 2 /* flip_dist_substep solves the two systems one after the other (default: side by side) */

enum flip_dist_reduce_op { FLIP_DIST_SUM = 0, FLIP_DIST_MAX = 1 };

/* The communication layer as the driver sees it.  flip_dist_create fills it with RCCL; flip_dist_create_with_transport
 * takes the caller's (the tests drive torch.distributed/gloo through it, two ranks sharing one GPU, which RCCL refuses).
 * Buffers are device memory.  Both operations are stream-ordered: they must take effect after the work already enqueued
 * on `stream` and before work enqueued on it later (a host-staged transport synchronises the stream). */
typedef struct flip_dist_transport {
    void *user;
    /* in-place all-reduce of count doubles */
    int (*all_reduce)(void *user, double *device_buf, int64_t count, int op, void *stream);
    /* one grouped exchange with both z neighbours: send_lo goes to rank - 1 and arrives in its recv_hi, send_hi goes to
     * rank + 1 and arrives in its recv_lo.  A side without a neighbour has NULL pointers and 0 bytes. */
    int (*exchange)(void *user, const void *send_lo, int64_t send_lo_bytes, void *recv_lo, int64_t recv_lo_bytes, const void *send_hi,
                    int64_t send_hi_bytes, void *recv_hi, int64_t recv_hi_bytes, void *stream);
} flip_dist_transport;

/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks by whatever channel started them. */
int flip_dist_unique_id(void *id_out);

/* One slab of the decomposition.  ctx: this rank's context over its EXTENDED box (owned planes plus `ghost` planes of
 * each neighbour, clipped to the global box), already created on this rank's device.  starts[world + 1]: first owned
 * plane of every rank in global plane numbers, starts[world] = planes of the global box.  overlap: planes of either
 * neighbour's slab the local preconditioner also covers (0: disjoint blocks; must be smaller than ghost).
 * particle_capacity: records one particle message can hold.  The RCCL communicator is created here
 * (ncclCommInitRank: collective over all ranks) with this process's current HIP device = the context's. */
int flip_dist_create(flip_ctx *ctx, int rank, int world, const void *unique_id, const int32_t *starts, int32_t ghost, int32_t overlap,
                     int64_t particle_capacity, int flags, flip_dist **out);
int flip_dist_create_with_transport(flip_ctx *ctx, int rank, int world, const flip_dist_transport *transport, const int32_t *starts,
                                    int32_t ghost, int32_t overlap, int64_t particle_capacity, int flags, flip_dist **out);
void flip_dist_destroy(flip_dist *d);
/* Text of the last error of this handle (or of the last failed create if d == NULL). */
const char *flip_dist_last_error(const flip_dist *d);

/* flip_build_system + the CG solve over all slabs (compute_pressure_{divirgence,density}_constraint,
 * src/FluidSimulator.cpp:654-830, 832-1014): every rank calls it with the same arguments and receives the same
 * iteration count and residual.  Ranks first agree that no rigid-body row is a dependent row anywhere (the slab solver
 * needs the eliminated system, see flip_set_system_options); if one is, every rank returns the same error before any
 * collective of the solve. */
int flip_dist_solve(flip_dist *d, int kind, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info);

/* Both solves of a substep (src/FluidSimulator.cpp:505 and :521) side by side: the two systems are assembled into two
 * system sets of the context (flip_pair_begin), each with a stream and a communicator of its own, and the two slab
 * solves advance in turn, so that the all-reduces and plane exchanges of one travel while the kernels of the other run.
 * Each solve performs the operations flip_dist_solve performs, hence the same results.  On return the divergence
 * system is the active one (flip_select_system switches, as after flip_solve_pair).  With one rank and no slab: flip_solve_pair. */
int flip_dist_solve_pair(flip_dist *d, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info_divergence,
                         flip_pcg_info *info_density);

/* Migration and ghost refresh after the particle update (no counterpart in the reference).  sent / received (may be
 * NULL): records toward / from the lower and the upper neighbour. */
int flip_dist_exchange_particles(flip_dist *d, int64_t sent[2], int64_t received[2]);

/* Records the last flip_dist_exchange_particles (also the one inside flip_dist_substep) sent toward / received from the
 * lower and the upper neighbour. */
int flip_dist_last_exchange(const flip_dist *d, int64_t sent[2], int64_t received[2]);

/* All-reduce of a few host doubles (count <= 16) in place: the velocity maximum of compute_cfl, the rigid body's J p. */
int flip_dist_all_reduce(flip_dist *d, double *values, int32_t count, int op);

typedef struct flip_dist_step_options {
    float flip_pic_ratio;        /* main.cpp:167 */
    double tol;                  /* src/FluidSimulator.cpp:660 */
    int32_t max_iter;
    int32_t precond;             /* enum flip_precond */
    int32_t density_projection;  /* DENSITY_CONSTRAINT_PPE, src/FluidSimulator.cpp:31 */
    int32_t constrain;           /* constrain_velocity, :357-361 and :404-408 */
    int32_t advect;              /* FluidDomain::advectParticles, :426 */
    int32_t rigid_body;          /* a SolidObject is attached on every rank (flip_solid_object_set) */
    float levelset_radius;       /* > 0: rebuild the fluid level set from the particles (:324) with this radius (world units) */
} flip_dist_step_options;

/* One substep over the decomposition: the loop body of FluidSimulator::advance_flip_pic (src/FluidSimulator.cpp:309-445)
 * without the fluid sources, the same call order as on one context, the two solves through flip_dist_solve and the
 * particle exchange at the end.  info[0]: divergence solve, info[1]: density solve (zeroed when off). */
int flip_dist_substep(flip_dist *d, float dt, const flip_dist_step_options *opt, flip_pcg_info info[2]);

#ifdef __cplusplus
}
#endif
#endif
