// flip_internal.h -- shared declarations of the libflip_hip.so translation units.
// gfx950 only: wave size 64 is hard-coded.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flip_hip.h"

#define FLIP_BLOCK 256          // threads per workgroup = cells per chunk
#define FLIP_WAVE 64
#define FLIP_MAX_PCG_BLOCKS 2048 // upper bound of the persistent PCG grid (partials per reduction)

// Per-cell flag byte of the assembled system.
#define FLIP_FLAG_ROW 0x40u      // the cell has a row
#define FLIP_FLAG_DEP 0x80u      // dependent row: no other row refers to it; solved by substitution after CG
// bits 0..5: the off-diagonal toward face f (-x,+x,-y,+y,-z,+z) is present

struct flip_pcg_scalars {
    double rz[2];        // r.z of the current / previous iteration (ping-pong by iteration parity)
    double rr;           // |r|^2 of the last update
    double rhs_norm2;    // |b|^2
    double threshold;    // tol^2 |b|^2
    double tol2;         // tol^2
    double jp[6];        // J p of the current iteration (rigid-body coupling)
    double beta;         // beta of the last direction kernel (the slab solver's ghost planes need it)
    int32_t it_a;        // iteration the next direction kernel runs
    int32_t it_b;        // iteration the next update kernel runs
    int32_t done;        // set once converged (or |b| = 0)
    int32_t max_iter;
    float creg;          // the coupling value of the rows flagged regular (see flip_ctx::reg)
    int32_t fused;       // the update kernel also pre-smooths (k_pcg_update_pre): the residual alternates between two
                         // buffers, the current one being number it_a & 1
    int32_t warm;        // the solve started from a guess: |b|^2 comes from partial slot 9, not from |r_0|^2
    // The stopping rule of headers/pcgsolver/pcg_solver.h:241-258, watched beside Eigen's: max|r| <= 1e-12 max|b|.
    int32_t track_inf;   // partial slots 10 (max|b|, written once) and 11 (max|r| of the last update) are maintained
    int32_t it_inf;      // iterations after which that rule first held (-1: not yet)
    double binf, rinf;   // max|b|, max|r| as of the last direction kernel
};

// Packed particle record read by the gather kernels (P2G, density): one 64-byte line per particle instead of
// one line in each of seven SoA arrays, because neighbouring lanes walk different particle ranges.
// 64-byte particle record: the wire format of the particle exchange (flip_particles_pack).
struct alignas(64) flip_prec {
    double x, y, z;
    float mass, pad0;
    double vx, vy, vz, pad1;
};
// The gather kernels read the particles in half-cell order from an array of such 64-byte records, addressed as two
// interleaved views of 32-byte halves (record d = element 2 d of either view): the position half carries the mass, the
// velocity half only the P2G walk reads.  Every record a walk visits lies inside the support of its cell, so P2G wants
// both halves and finds them in one cache line.  The level-set search reads a second array of position halves alone,
// in cell order (flip_ctx::pcell).
struct alignas(32) flip_ppos {
    double x, y, z;
    float mass, pad0;
};
struct alignas(32) flip_pvel {
    double vx, vy, vz, pad1;
};

// One coarse level of the multigrid preconditioner (flip_mg.hip): dense zero-padded arrays over the
// coarsened box plus the compact list of its rows.
struct flip_mg_level {
    int nx = 0, ny = 0, nz = 0;
    int64_t B = 0, pad = 0;
    double *cx = nullptr, *cy = nullptr, *cz = nullptr, *diag = nullptr, *inv = nullptr;
    double *x = nullptr, *b = nullptr, *t = nullptr;
    int32_t *rows = nullptr, *scan = nullptr, *n_rows = nullptr;
    // float copies of the five operator arrays, for a level large enough to be bound by its bytes (the list-driven
    // kernels read these; the next level is still coarsened from the doubles); nullptr on the small levels
    float *cx_f = nullptr, *cy_f = nullptr, *cz_f = nullptr, *diag_f = nullptr, *inv_f = nullptr;
};

// The second assembled system of a context (flip_solve_pair keeps the divergence and the density system side by
// side and iterates on both at once).  Field for field what flip_ctx holds for its active system; flip_swap_systems
// exchanges the two sets, so every kernel wrapper keeps reading the context's own fields.
struct flip_system_set {
    int sys_kind = -1;
    int x_kind = -1;
    float sys_dt = 0.f;
    double sys_s2 = 1.0;
    bool sys_coupled = false, sys_eliminated = false;
    float *term[3] = {};
    uint8_t *flags = nullptr;
    double *diag = nullptr, *invdiag = nullptr, *rhs = nullptr;
    double *x = nullptr, *r = nullptr, *r2 = nullptr, *z = nullptr, *pv[2] = {}, *ap = nullptr;
    float *cg_c[3] = {};
    uint8_t *reg = nullptr;
    float reg_value = 0.f;
    double *cg_diag = nullptr, *cg_inv = nullptr;
    int32_t *cg_rows = nullptr, *chunk_flag = nullptr, *chunk_list = nullptr, *d_counts = nullptr;
    int32_t n_active_chunks = 0;
    double *partial = nullptr;
    flip_pcg_scalars *scal = nullptr;
    std::vector<hipEvent_t> events;
    std::vector<flip_mg_level> mg;
    bool mg_narrowed = false;       // the assembly has written mg_diag_f / mg_inv_f for the system at hand
    float *mg_diag_f = nullptr, *mg_inv_f = nullptr;
    hipGraphExec_t batch_exec[2] = { nullptr, nullptr };
    uint64_t batch_key[2] = { 0, 0 };
    int32_t *cg_scan = nullptr, *scan_tmp = nullptr;   // scan scratch: the two systems are set up side by side
    int64_t scan_tmp_len = 0;
    // the slab solve's arrays and state (see flip_ctx): two slab solves can be in progress side by side (flip_pair_*)
    float *mg_cz = nullptr;
    double *dist_sums = nullptr;
    bool dist_active = false, dist_mg = false;
    int32_t dist_max_iter = 0;
    int dist_prof = 0;
    int32_t *mg_rows = nullptr;
    double *dist_recv[2] = {};
    int32_t dist_recv_planes = 0;
    double *dist_rw = nullptr;
    bool allocated = false;
};

struct flip_ctx {
    flip_grid_desc g;
    int64_t B = 0;       // cells in the box
    int32_t n_chunks = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- particles, kept in cell-sorted order; pid = upload index ----
    int64_t n = 0, cap = 0;
    double *p[2][6] = {};   // [buffer][x y z vx vy vz]
    float *mass[2] = {};
    int32_t *pid[2] = {};
    int cur = 0;            // buffer holding the current order
    int32_t *cell_id = nullptr;     // [cap] cell of each particle in current order (B = outside)
    int32_t *slot_src = nullptr;    // [cap] scatter scratch
    int32_t *slot_key = nullptr;    // [cap]
    int32_t *cell_start = nullptr;  // [B+2] exclusive scan of the per-cell counts (+1 bin for outside)
    int32_t *cell_fill = nullptr;   // [B+1]
    int32_t *scan_tmp = nullptr;    // block sums of the scan
    int64_t scan_tmp_len = 0;
    double *stage = nullptr;        // [cap*3] upload/download staging (AoS)
    flip_prec *prec = nullptr;      // [cap] x 64 bytes: position half then velocity half of each record, the gather kernels' copy,
                                    // sorted by half cell (see k_reorder_pack)
    flip_ppos *pcell = nullptr;     // [cap] positions + mass in the primary (cell) order, for the level-set search
    int32_t *sub_start = nullptr;   // [8 B + 1] run boundaries of the half-cell order: per grid row 4 sub-rows of 2 nx
    int32_t *ls_list = nullptr;     // [B] cells left to the second pass of the level-set search (allocated on first use)
    int32_t *sub_slot = nullptr;    // [cap] place of primary slot s in the half-cell order
    int64_t n_tiles = 0;            // 4 x 4 x 4-cell tiles of the box (the LDS-staged gather's work units)
    int32_t *tile_scan = nullptr;   // [n_tiles + 1] flags, then their exclusive scan; [n_tiles] = tiles that hold a band cell
    int32_t *slot_scan = nullptr;   // [B + 1] flags of the per-lane gather's slots, then their exclusive scan; [slots] = listed slots
    int32_t *slot_list = nullptr;   // [B] the slots that hold a cell of the band (k_gather_slot_flags)
    int32_t *gather_claim = nullptr; // [8] claim counters of k_gather, one per part of the list
    int slot_list_shape = 0;        // what the lists hold: cells per slot as ry * 16 + rz, -1 = the tile list, 0 = nothing
    int32_t *tile_list = nullptr;   // [n_tiles] those tiles, ascending
    int n_cu = 256;                 // compute units of the device
    int gather_mode = 0;            // 0: k_gather (per-lane walk of global memory), 1: k_gather_tile (LDS-staged)
    int32_t *gather_stats = nullptr; // [4] tiles of the LDS-staged gather since creation: 0 staged, 1 walked from global memory (too crowded)
    bool prec_valid = false;
    bool binned = false;
    bool slots_valid = false;       // cell_start still describes the slot order (no upload / exchange since the sort)
    bool density_valid = false;     // ctx->density matches the current particles, sort and phi_fluid
    int64_t n_outside = 0;

    // ---- dense fields ----
    float *phi_fluid = nullptr, *phi_solid = nullptr, *phi_obj = nullptr;
    double *vel[4][3] = {};         // 0 front, 1 back, 2 prev, 3 diff
    double *disp[3] = {};
    float *wgt[3] = {}, *fwgt[3] = {}, *owgt[3] = {};
    // flip_face_fractions writes these and swaps them with wgt / owgt, so that a stage enqueued before it on the other
    // stream (constrain_velocity, the labels) still reads the fractions of the previous substep
    float *wgt_back[3] = {}, *owgt_back[3] = {};
    float *density = nullptr;
    uint8_t *valid[2][3] = {};      // front / back
    uint8_t *active = nullptr, *active_back = nullptr;
    uint8_t *label = nullptr, *band = nullptr, *seed = nullptr, *tmp_u8 = nullptr;
    int32_t *row_index = nullptr;

    // ---- rigid body ----
    bool has_obj = false;
    float phi_obj_bg = 0.f;
    double obj_com[3] = {}, obj_minv[6] = {}, obj_vel[6] = {};

    // ---- assembled system + PCG ----
    int sys_kind = -1;
    int x_kind = -1;                // kind of the system whose converged solution x holds (-1: none); the next solve of
                                    // that kind on this set of arrays may start from it (warm_start)
    bool warm_start = false;        // flip_set_warm_start
    float sys_dt = 0.f;
    double sys_s2 = 1.0;            // dx^2 scaling of the divergence system
    bool sys_coupled = false;
    bool drop_null_rows = true;     // rhs of all-zero rows forced to 0
    bool eliminate_dep = true;      // rows of cells with phi_fluid >= 0 are eliminated from the CG system
    bool sys_eliminated = false;    // the assembled system carries FLIP_FLAG_DEP rows
    int32_t sys_dep_coupled = 0;    // dependent rows with a rigid-body entry found by the last assembly (they rule the elimination out)
    float *term[3] = {};            // (1 - W) * scale at the low x/y/z face of every cell
    uint8_t *flags = nullptr;
    double *diag = nullptr, *invdiag = nullptr, *rhs = nullptr;
    double *J[6] = {};              // rigid-body rows, dense over the box
    // Solver arrays carry `pad` zero elements on both sides (pointers are pre-offset), so the stencil
    // kernels read c +- 1, c +- nx, c +- nx*ny of any cell of the box without a bounds test.
    int64_t pad = 0;
    double *x = nullptr, *r = nullptr, *z = nullptr, *pv[2] = {}, *ap = nullptr;
    double *r2 = nullptr;           // second residual buffer of the fused update + pre-smoothing kernel
    // CG view of the system: symmetric coupling coefficient toward the low x/y/z neighbour (0 = none),
    // diagonal and Jacobi inverse; all zero on cells that are not rows of the CG system
    float *cg_c[3] = {};
    // rows all six of whose couplings have the one common value (interior liquid cells: no solid face, no free surface
    // next to them): the stencil kernels take the value from flip_pcg_scalars::creg and do not read the coupling
    // arrays for them.  All zero on a slab context.
    uint8_t *reg = nullptr;         // [B]
    float reg_value = 0.f;
    double *cg_diag = nullptr, *cg_inv = nullptr;
    int32_t *cg_scan = nullptr;     // [B+1] scan scratch
    int32_t *cg_rows = nullptr;     // [B] cell ids of the CG rows, ascending; count in d_counts[6]
    int32_t *chunk_flag = nullptr, *chunk_list = nullptr;   // [n_chunks+1]
    int32_t *d_counts = nullptr;    // [8] device counters: 0 rows, 1 active chunks, 2 overdense, 3 outside
    int32_t n_rows = 0, n_active_chunks = 0;
    int pcg_blocks = 1024;
    double *partial = nullptr;      // [16][FLIP_MAX_PCG_BLOCKS]
    flip_pcg_scalars *scal = nullptr;
    double *d_vmax = nullptr;       // [3*FLIP_MAX_PCG_BLOCKS]
    bool profiling = false;
    std::vector<hipEvent_t> events;
    // flip_set_profiling: one iteration of a solve has its level-0 kernels bracketed one by one (kprof_arm is set for
    // that iteration only; the V-cycle records its own marks): 0 start, 1 after the direction kernel, 2 after the
    // update (+ pre-smoothing), 3 after the level-0 restriction, 4 after the coarse levels, 5 after the level-0
    // correction + first post-smoothing step, 6 after the second
    hipEvent_t kprof[7] = {};
    bool kprof_arm = false, kprof_valid = false;
    std::vector<flip_mg_level> mg;  // coarse levels, finest first
    float *mg_diag_f = nullptr, *mg_inv_f = nullptr; // float copies of cg_diag / cg_inv for the V-cycle's level 0 (one allocation)
    bool mg_narrowed = false;       // the assembly has written mg_diag_f / mg_inv_f for the system at hand
    // one batch of solver iterations as an instantiated graph (the kernels' arguments do not change from solve to
    // solve; batch_key says which arguments it was captured with); [0] the long batch, [1] the short one used near the
    // expected end of a solve
    hipGraphExec_t batch_exec[2] = { nullptr, nullptr };
    uint64_t batch_key[2] = { 0, 0 };
    int32_t last_iterations[2] = { 0, 0 };   // iterations the last solve of each kind took (0: none yet)

    // ---- second system and the two streams of flip_solve_pair ----
    flip_system_set other;
    hipStream_t pair_stream[2] = { nullptr, nullptr };
    hipEvent_t pair_event[3] = { nullptr, nullptr, nullptr };

    // ---- z-slab decomposition of the solve (flip_set_slab): rows of planes [slab_k0, slab_k1) belong to this
    // context, the other planes of the box are ghosts whose values a neighbouring rank supplies ----
    bool slab_on = false;
    int32_t slab_k0 = 0, slab_k1 = 0;
    float *mg_cz = nullptr;         // z couplings without those across the slab faces: the local preconditioner's view
    double *dist_sums = nullptr;    // [8] local sums handed to the caller's all-reduce
    bool dist_active = false, dist_mg = false;
    int32_t dist_max_iter = 0;
    int dist_prof = 0;              // direction launches bracketed by events in this solve (profiling)
    // overlapping variant (flip_set_slab_overlap): the local preconditioner also covers `slab_overlap` planes of either
    // neighbour's slab; the residual there comes from the neighbour and the correction computed there goes back to it
    int32_t slab_overlap = 0;
    int32_t *mg_rows = nullptr;     // [B] rows of planes [slab_k0 - overlap, slab_k1 + overlap); count in d_counts[7]
    double *dist_recv[2] = {};      // [(overlap + 1) planes] corrections received from the lower / upper neighbour
    int32_t dist_recv_planes = 0;   // the overlap dist_recv was sized for
    bool slab_weighted = false;     // partition-of-unity weights across the overlap (see flip_set_slab_overlap)
    double *dist_rw = nullptr;      // weighted copy of the residual, the V-cycle's right-hand side when slab_weighted
    // flip_pair_begin .. flip_pair_end: the caller works on the two system sets in turn, each on a stream of its own
    bool pair_on = false;
    int pair_slot = 0;              // which set is the active one (0: the set that was active at flip_pair_begin)
    hipStream_t pair_main_stream = nullptr;   // the context's stream while ctx->stream points at a pair stream

    // ---- flip_side_fork / select / join: a second stream for the stages of a substep that do not depend on each other ----
    bool side_enabled = true;       // flip_set_side_stream
    bool side_forked = false, side_selected = false;
    hipStream_t side_stream = nullptr, side_main_stream = nullptr;
    hipEvent_t side_event[2] = { nullptr, nullptr };
};

int flip_default_gather_mode();            // FLIP_GATHER_TILE
int flip_alloc_system(flip_ctx *ctx);      // the active system's arrays (used for both sets)
void flip_free_system(flip_ctx *ctx);
int flip_zero_solver_vectors(flip_ctx *ctx, bool keep_x, hipStream_t st);   // one fill: the vectors are one allocation
void flip_swap_systems(flip_ctx *ctx);     // active <-> other
int flip_pair_streams(flip_ctx *ctx);      // creates the two solver streams and their events if need be
int flip_mg_setup(flip_ctx *ctx);   // coarse operators of the assembled system
int flip_mg_vcycle(flip_ctx *ctx, bool pre_smoothed = false);  // z = M^-1 r and the r.z partials; pre_smoothed: level 0's
                                                               // pre-smoothing has been done by k_pcg_update_pre
void flip_mg_free(flip_ctx *ctx);
void flip_mg_smoother(double *c1, double *a2, double *c2); // coefficients of the level smoother (see flip_mg.hip)

// dst[0] = src[0] on the device, as a one-thread kernel on `st`: stays inside the stream's kernel order whatever the
// runtime does with small device-to-device copies
int flip_copy_i32(flip_ctx *ctx, int32_t *dst, const int32_t *src, hipStream_t st);
// Zeroes freshly allocated device memory and waits for it.  hipMemset runs on the null stream and may return before
// the fill has executed; the solver's own streams are non-blocking and not ordered against it, so a kernel on one of
// them could otherwise write the new array before the fill does.  Allocation time only.
hipError_t flip_zero_new(void *p, size_t bytes);

int flip_fail(flip_ctx *ctx, const char *what, const char *file, int line);
int flip_fail_hip(flip_ctx *ctx, hipError_t e, const char *what, const char *file, int line);

#define FLIP_CHECK(call)                                                        \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) return flip_fail_hip(ctx, e_, #call, __FILE__, __LINE__); \
    } while (0)
#define FLIP_REQUIRE(cond, msg)                                                 \
    do {                                                                        \
        if (!(cond)) return flip_fail(ctx, msg, __FILE__, __LINE__);            \
    } while (0)
#define FLIP_LAUNCH_CHECK() FLIP_CHECK(hipGetLastError())

// Dynamic LDS of the per-row reordering kernel (k_reorder_pack): 8 nx half-cell counters + 4 wave totals, and the upload
// index (32 bit) and half-cell number (16 bit) of up to FLIP_ROW_QCAP particles of the row.
#define FLIP_ROW_QCAP 6144
static inline size_t flip_subrow_lds_bytes_for(int nx, int qcap) { return ((size_t)8 * nx + 4) * sizeof(int32_t) + (size_t)qcap * (sizeof(int32_t) + sizeof(uint16_t)); }
static inline size_t flip_subrow_lds_bytes(int nx) { return flip_subrow_lds_bytes_for(nx, FLIP_ROW_QCAP); }

static inline int flip_grid_blocks(int64_t n) { return (int)((n + FLIP_BLOCK - 1) / FLIP_BLOCK); }

// Exclusive scan of int32 data[0..n) in place (device); total written to *total_out (device) if not null.
int flip_exclusive_scan(flip_ctx *ctx, int32_t *data, int64_t n);

// ---------------------------------------------------------------------------------------------
// Device helpers.  Arithmetic types mirror the reference expression by expression.

struct GridDims {
    int nx, ny, nz, ox, oy, oz;
    long long B;
};
static inline GridDims flip_dims(const flip_ctx *c) {
    return GridDims{ c->g.nx, c->g.ny, c->g.nz, c->g.ox, c->g.oy, c->g.oz, (long long)c->B };
}

#ifdef __HIPCC__
__device__ __forceinline__ bool d_inside(const GridDims &g, int i, int j, int k) {
    return i >= 0 && j >= 0 && k >= 0 && i < g.nx && j < g.ny && k < g.nz;
}
__device__ __forceinline__ long long d_lid(const GridDims &g, int i, int j, int k) {
    return (long long)i + (long long)g.nx * ((long long)j + (long long)g.ny * k);
}
__device__ __forceinline__ void d_ijk(const GridDims &g, long long c, int &i, int &j, int &k) {
    i = (int)(c % g.nx);
    long long t = c / g.nx;
    j = (int)(t % g.ny);
    k = (int)(t / g.ny);
}
__device__ __forceinline__ float d_getf(const GridDims &g, const float *a, int i, int j, int k, float bg) {
    return d_inside(g, i, j, k) ? a[d_lid(g, i, j, k)] : bg;
}
__device__ __forceinline__ double d_getd(const GridDims &g, const double *a, int i, int j, int k) {
    return d_inside(g, i, j, k) ? a[d_lid(g, i, j, k)] : 0.0;
}

// headers/util.h:470-478 (T = double)
__device__ __forceinline__ double d_hat(double x) {
    if (-1. <= x && x <= 0.) return 1 + x;
    else if (0. <= x && x <= 1.) return 1 - x;
    else return 0;
}
// The same function for an argument known to lie in [-1, 1], and for one known to be >= -1: fewer instructions,
// identical bits (1 + x with x < 0 is the IEEE operation 1 - |x|).
__device__ __forceinline__ double d_hat_in(double x) { return 1 - fabs(x); }
__device__ __forceinline__ double d_hat_up(double x) { return x > 1. ? 0. : 1 - fabs(x); }
// headers/util.h:200-208
__device__ __forceinline__ float d_clampf(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }
// headers/util.h:493-501 (T = float)
__device__ __forceinline__ float d_frac2(float l, float r) {
    if (l < 0 && r < 0) return 1;
    if (l < 0 && r >= 0) return l / (l - r);
    if (l >= 0 && r < 0) return r / (r - l);
    return 0;
}
// max(0.001f, fraction_inside(a, b)) as used for the ghost-fluid theta
__device__ __forceinline__ float d_theta(float l, float r) {
    const float t = d_frac2(l, r);
    return 0.001f > t ? 0.001f : t;
}
// headers/util.h:505-580 (T = float)
__device__ float d_frac4(float phi_bl, float phi_br, float phi_tl, float phi_tr);

// Trilinear sample of a dense double field (OpenVDB BoxSampler): z, then y, then x; each step a + (b-a)*t.
__device__ __forceinline__ double d_lerp(double a, double b, double t) {
    const double tmp = (b - a) * t;
    return a + tmp;
}
__device__ __forceinline__ double d_box_sample(const GridDims &g, const double *__restrict__ f, double x, double y, double z) {
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    const int i = (int)fx - g.ox, j = (int)fy - g.oy, k = (int)fz - g.oz;
    const double tu = x - fx, tv = y - fy, tw = z - fz;
    double d000, d001, d010, d011, d100, d101, d110, d111;
    if (i >= 0 && j >= 0 && k >= 0 && i + 1 < g.nx && j + 1 < g.ny && k + 1 < g.nz) {
        const long long c = d_lid(g, i, j, k), sy = g.nx, sz = (long long)g.nx * g.ny;
        d000 = f[c]; d100 = f[c + 1];
        d010 = f[c + sy]; d110 = f[c + sy + 1];
        d001 = f[c + sz]; d101 = f[c + sz + 1];
        d011 = f[c + sz + sy]; d111 = f[c + sz + sy + 1];
    } else {
        d000 = d_getd(g, f, i, j, k); d100 = d_getd(g, f, i + 1, j, k);
        d010 = d_getd(g, f, i, j + 1, k); d110 = d_getd(g, f, i + 1, j + 1, k);
        d001 = d_getd(g, f, i, j, k + 1); d101 = d_getd(g, f, i + 1, j, k + 1);
        d011 = d_getd(g, f, i, j + 1, k + 1); d111 = d_getd(g, f, i + 1, j + 1, k + 1);
    }
    return d_lerp(d_lerp(d_lerp(d000, d001, tw), d_lerp(d010, d011, tw), tv),
                  d_lerp(d_lerp(d100, d101, tw), d_lerp(d110, d111, tw), tv), tu);
}

// Sum over the 256 threads of a workgroup in a fixed order (lane tree, then waves 0..3);
// every thread receives the result.  scratch: 4 doubles of LDS.
__device__ __forceinline__ double d_block_sum(double v, double *scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, FLIP_WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}
// d_block_sum of s and d_block_max of m behind one pair of barriers (scratch: 8 doubles)
__device__ __forceinline__ void d_block_sum_max(double &s, double &m, double *scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, FLIP_WAVE);
        m = fmax(m, __shfl_down(m, off, FLIP_WAVE));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch[wave] = s; scratch[4 + wave] = m; }
    __syncthreads();
    s = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
    m = fmax(fmax(scratch[4], scratch[5]), fmax(scratch[6], scratch[7]));
}
__device__ __forceinline__ double d_block_max(double v, double *scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, FLIP_WAVE));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return fmax(fmax(scratch[0], scratch[1]), fmax(scratch[2], scratch[3]));
}
#endif
