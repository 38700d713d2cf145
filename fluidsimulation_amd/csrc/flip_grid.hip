// flip_grid.hip -- grid-side kernels: face fractions, cell labels, extrapolation / forces / buffer
// updates, and the two projections.
#include "flip_internal.h"

// headers/util.h:505-580 (T = float); list rotation as cycle_array (util.h:485-490)
__device__ float d_frac4(float phi_bl, float phi_br, float phi_tl, float phi_tr) {
    const int inside_count = (phi_bl < 0 ? 1 : 0) + (phi_tl < 0 ? 1 : 0) + (phi_br < 0 ? 1 : 0) + (phi_tr < 0 ? 1 : 0);
    float l0 = phi_bl, l1 = phi_br, l2 = phi_tr, l3 = phi_tl;
#define CYCLE4() do { const float t_ = l0; l0 = l1; l1 = l2; l2 = l3; l3 = t_; } while (0)
    float rv;
    if (inside_count == 4) return 1;
    else if (inside_count == 3) {
        for (int q = 0; q < 4 && l0 < 0; ++q) CYCLE4();
        const float side0 = 1 - d_frac2(l0, l3);
        const float side1 = 1 - d_frac2(l0, l1);
        rv = 1 - 0.5f * side0 * side1;
    } else if (inside_count == 2) {
        for (int q = 0; q < 4 && (l0 >= 0 || !(l1 < 0 || l2 < 0)); ++q) CYCLE4();
        if (l1 < 0) {
            const float side_left = d_frac2(l0, l3);
            const float side_right = d_frac2(l1, l2);
            rv = 0.5f * (side_left + side_right);
        } else {
            const float middle = 0.25f * (l0 + l1 + l2 + l3);
            if (middle < 0) {
                float area = 0;
                const float side1 = 1 - d_frac2(l0, l3);
                const float side3 = 1 - d_frac2(l2, l3);
                area += 0.5f * side1 * side3;
                const float side2 = 1 - d_frac2(l2, l1);
                const float side0 = 1 - d_frac2(l0, l1);
                area += 0.5f * side0 * side2;
                rv = 1 - area;
            } else {
                float area = 0;
                const float side0 = d_frac2(l0, l1);
                const float side1 = d_frac2(l0, l3);
                area += 0.5f * side0 * side1;
                const float side2 = d_frac2(l2, l1);
                const float side3 = d_frac2(l2, l3);
                area += 0.5f * side2 * side3;
                rv = area;
            }
        }
    } else if (inside_count == 1) {
        for (int q = 0; q < 4 && l0 >= 0; ++q) CYCLE4();
        const float side0 = d_frac2(l0, l3);
        const float side1 = d_frac2(l0, l1);
        rv = 0.5f * side0 * side1;
    } else
        rv = 0;
#undef CYCLE4
    if ((double)rv > 0.9) return 1;
    if ((double)rv < 0.1) return 0;
    return d_clampf(rv, 0.f, 1.f);
}

// ---------------------------------------------------------------------------------------------
// Face fractions (src/FluidSimulator.cpp:527-623, src/SolidObject.cpp:103-152)

// Seed of the mask that pre-allocates the weight grids: phi < 0, or phi >= 0 with a face neighbour < 0
// (isovalue-crossing voxels on the outer side).  The mask itself is the seed dilated over the
// 26-neighbourhood (src/FluidSimulator.cpp:562-566); the fraction kernel tests that on the fly.
__global__ void k_weight_seed(GridDims g, const float *__restrict__ phi, float bg, uint8_t *__restrict__ seed) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    bool s = phi[c] < 0;
    if (!s)
        s = d_getf(g, phi, i - 1, j, k, bg) < 0 || d_getf(g, phi, i + 1, j, k, bg) < 0 || d_getf(g, phi, i, j - 1, k, bg) < 0
            || d_getf(g, phi, i, j + 1, k, bg) < 0 || d_getf(g, phi, i, j, k - 1, bg) < 0 || d_getf(g, phi, i, j, k + 1, bg) < 0;
    seed[c] = s ? 1 : 0;
}

// OpenVDB BoxSampler on a float grid: (b - a) * t is a double product narrowed back to float.
__device__ __forceinline__ float d_lerpf(float a, float b) {
    const double tmp = (double)(b - a) * 0.5;
    return a + (float)tmp;
}

// One thread per cell: the level-set values of the 3x3x3 neighbourhood give the seven corner values
// (trilinear resample half a voxel below the centres, src/FluidSimulator.cpp:554-560, 585-591).
__global__ void k_face_fractions(GridDims g, const float *__restrict__ phi, float bg, const uint8_t *__restrict__ seed,
                                 float *__restrict__ wu, float *__restrict__ wv, float *__restrict__ ww) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    bool in_mask = false;
    for (int kk = k - 1; kk <= k + 1 && !in_mask; ++kk)
        for (int jj = j - 1; jj <= j + 1 && !in_mask; ++jj)
            for (int ii = i - 1; ii <= i + 1; ++ii)
                if (d_inside(g, ii, jj, kk) && seed[d_lid(g, ii, jj, kk)]) { in_mask = true; break; }
    if (!in_mask) {
        wu[c] = 0.f; wv[c] = 0.f; ww[c] = 0.f;
        return;
    }
    float p[3][3][3]; // [x][y][z] offsets -1..1
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int d = 0; d < 3; ++d) p[a][b][d] = d_getf(g, phi, i - 1 + a, j - 1 + b, k - 1 + d, bg);
    // corner (a,b,d) in {0,1}^3: low corner of cell (i+a, j+b, k+d) = mean over cells (i+a-1..i+a, ...)
#define CORNER(a, b, d)                                                                                          \
    d_lerpf(d_lerpf(d_lerpf(p[a][b][d], p[a][b][d + 1]), d_lerpf(p[a][b + 1][d], p[a][b + 1][d + 1])),            \
            d_lerpf(d_lerpf(p[a + 1][b][d], p[a + 1][b][d + 1]), d_lerpf(p[a + 1][b + 1][d], p[a + 1][b + 1][d + 1])))
    const float p000 = CORNER(0, 0, 0), p001 = CORNER(0, 0, 1), p010 = CORNER(0, 1, 0), p011 = CORNER(0, 1, 1);
    const float p100 = CORNER(1, 0, 0), p101 = CORNER(1, 0, 1), p110 = CORNER(1, 1, 0);
#undef CORNER
    wu[c] = d_frac4(p000, p010, p001, p011);
    wv[c] = d_frac4(p000, p001, p100, p101);
    ww[c] = d_frac4(p000, p010, p100, p110);
}

// src/FluidSimulator.cpp:603-619: add the rigid body's fractions and clamp to [0,1]
__global__ void k_combine_weights(long long B, float *__restrict__ w0, float *__restrict__ w1, float *__restrict__ w2,
                                  const float *__restrict__ o0, const float *__restrict__ o1, const float *__restrict__ o2,
                                  int has_obj) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    w0[c] = d_clampf(has_obj ? w0[c] + o0[c] : w0[c], 0.f, 1.f);
    w1[c] = d_clampf(has_obj ? w1[c] + o1[c] : w1[c], 0.f, 1.f);
    w2[c] = d_clampf(has_obj ? w2[c] + o2[c] : w2[c], 0.f, 1.f);
}

extern "C" int flip_face_fractions(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const GridDims g = flip_dims(ctx);
    const int nb = flip_grid_blocks(ctx->B);
    hipStream_t st = ctx->stream;
    // The solid and the rigid body's fractions go to the back buffers, which then become the front ones: a stage enqueued
    // earlier on the context's other stream (flip_side_select) keeps reading the previous substep's values undisturbed.
    if (ctx->has_obj) { // SolidObject::compute_face_fractions: mask from the body's own level set
        k_weight_seed<<<nb, FLIP_BLOCK, 0, st>>>(g, ctx->phi_obj, ctx->phi_obj_bg, ctx->seed);
        FLIP_LAUNCH_CHECK();
        k_face_fractions<<<nb, FLIP_BLOCK, 0, st>>>(g, ctx->phi_obj, ctx->phi_obj_bg, ctx->seed, ctx->owgt_back[0], ctx->owgt_back[1],
                                                   ctx->owgt_back[2]);
        FLIP_LAUNCH_CHECK();
        for (int a = 0; a < 3; ++a) std::swap(ctx->owgt[a], ctx->owgt_back[a]);
    }
    k_weight_seed<<<nb, FLIP_BLOCK, 0, st>>>(g, ctx->phi_fluid, ctx->g.phi_fluid_bg, ctx->seed);
    FLIP_LAUNCH_CHECK();
    k_face_fractions<<<nb, FLIP_BLOCK, 0, st>>>(g, ctx->phi_fluid, ctx->g.phi_fluid_bg, ctx->seed, ctx->fwgt[0], ctx->fwgt[1],
                                               ctx->fwgt[2]);
    FLIP_LAUNCH_CHECK();
    k_face_fractions<<<nb, FLIP_BLOCK, 0, st>>>(g, ctx->phi_solid, ctx->g.phi_solid_bg, ctx->seed, ctx->wgt_back[0], ctx->wgt_back[1],
                                               ctx->wgt_back[2]);
    FLIP_LAUNCH_CHECK();
    for (int a = 0; a < 3; ++a) std::swap(ctx->wgt[a], ctx->wgt_back[a]);
    k_combine_weights<<<nb, FLIP_BLOCK, 0, st>>>(ctx->B, ctx->wgt[0], ctx->wgt[1], ctx->wgt[2], ctx->owgt[0], ctx->owgt[1],
                                                ctx->owgt[2], ctx->has_obj ? 1 : 0);
    FLIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int flip_solid_object_set(flip_ctx *ctx, const float *phi_object, float bg, const double com[3],
                                     const double minv[6], const double vel[6]) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (!phi_object) {
        ctx->has_obj = false;
        for (int a = 0; a < 3; ++a) FLIP_CHECK(hipMemsetAsync(ctx->owgt[a], 0, (size_t)ctx->B * 4, ctx->stream));
        return 0;
    }
    FLIP_REQUIRE(com && minv && vel, "flip_solid_object_set: null argument");
    FLIP_CHECK(hipMemcpyAsync(ctx->phi_obj, phi_object, (size_t)ctx->B * 4, hipMemcpyHostToDevice, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->phi_obj_bg = bg;
    for (int a = 0; a < 3; ++a) ctx->obj_com[a] = com[a];
    for (int a = 0; a < 6; ++a) { ctx->obj_minv[a] = minv[a]; ctx->obj_vel[a] = vel[a]; }
    ctx->has_obj = true;
    return 0;
}
extern "C" int flip_solid_object_velocity(flip_ctx *ctx, double vel[6]) {
    FLIP_REQUIRE(ctx && vel, "null argument");
    for (int a = 0; a < 6; ++a) vel[a] = ctx->obj_vel[a];
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Cell labels (src/FluidSimulator.cpp:267-305)

__global__ void k_label(long long B, const uint8_t *__restrict__ band, const float *__restrict__ phi_f,
                        const float *__restrict__ phi_s, const float *__restrict__ wu, const float *__restrict__ wv,
                        const float *__restrict__ ww, uint8_t *__restrict__ label, int32_t *__restrict__ scan_in) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > B) return;
    if (c == B) { scan_in[c] = 0; return; }
    uint8_t l = 0;
    if (band[c] & 1) {
        const float uf = 1 - wu[c], vf = 1 - wv[c], wf = 1 - ww[c];
        if (phi_f[c] < 0 || (((0 < uf && uf < 1) || (0 < vf && vf < 1) || (0 < wf && wf < 1)) && phi_s[c] <= 0)) l = 1;
    }
    label[c] = l;
    scan_in[c] = l;
}
__global__ void k_finish_index(long long B, const uint8_t *__restrict__ label, int32_t *__restrict__ index) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (!label[c]) index[c] = -1;
}

extern "C" int flip_index_cells(flip_ctx *ctx, int32_t *n_rows) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->binned, "flip_index_cells: call flip_bin_particles first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t B = ctx->B;
    k_label<<<flip_grid_blocks(B + 1), FLIP_BLOCK, 0, st>>>(B, ctx->band, ctx->phi_fluid, ctx->phi_solid, ctx->wgt[0], ctx->wgt[1],
                                                          ctx->wgt[2], ctx->label, ctx->row_index);
    FLIP_LAUNCH_CHECK();
    if (int rc = flip_exclusive_scan(ctx, ctx->row_index, B + 1)) return rc;
    if (int rc = flip_copy_i32(ctx, ctx->d_counts, ctx->row_index + B, st)) return rc;
    k_finish_index<<<flip_grid_blocks(B), FLIP_BLOCK, 0, st>>>(B, ctx->label, ctx->row_index);
    FLIP_LAUNCH_CHECK();
    // the count stays on the device unless the caller asks for it: reading it back makes the host wait for the stream and
    // leaves the device idle until the next launch arrives (flip_pcg_solve fetches it if it needs Eigen's default limit)
    ctx->n_rows = -1;
    if (n_rows) {
        int32_t n = 0;
        FLIP_CHECK(hipMemcpyAsync(&n, ctx->d_counts, 4, hipMemcpyDeviceToHost, st));
        FLIP_CHECK(hipStreamSynchronize(st));
        ctx->n_rows = n;
        *n_rows = n;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Glue between P2G and the solves

__global__ void k_mark_valid(long long B, const uint8_t *__restrict__ active, uint8_t *v0, uint8_t *v1, uint8_t *v2) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (active[c]) { v0[c] = 1; v1[c] = 1; v2[c] = 1; }
}
extern "C" int flip_mark_valid_from_active(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    k_mark_valid<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->active, ctx->valid[0][0], ctx->valid[0][1],
                                                                          ctx->valid[0][2]);
    FLIP_LAUNCH_CHECK();
    return 0;
}

// One Jacobi sweep of src/FluidSimulator.cpp:181-265 for the three components: a face that is not
// valid takes the average of its valid face neighbours (sum kept in float, :212).  The reference reads
// the front buffers and writes the back buffers; here the values stay in place -- a sweep reads values only at
// faces that were valid before it and writes only at faces that were not, so nothing a thread reads is written in
// the same launch -- and only the flags go from one set of arrays to the other.  Same values, and a sweep moves
// 8 bytes per cell instead of 56.
struct ExtrapPtrs {
    double *f[3];
    const uint8_t *vin[3];
    uint8_t *vout[3];
};
__global__ void k_extrapolate(GridDims g, ExtrapPtrs P, const uint8_t *__restrict__ act_in, uint8_t *__restrict__ act_out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    const long long sy = g.nx, sz = (long long)g.nx * g.ny;
    uint8_t act = act_in[c];
    const uint8_t ok3[3] = { P.vin[0][c], P.vin[1][c], P.vin[2][c] };
    if (ok3[0] && ok3[1] && ok3[2]) { // the common case: nothing to do but carry the flags over
        P.vout[0][c] = ok3[0]; P.vout[1][c] = ok3[1]; P.vout[2][c] = ok3[2];
        act_out[c] = act;
        return;
    }
    int i, j, k;
    d_ijk(g, c, i, j, k);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double *f = P.f[a];
        const uint8_t *v = P.vin[a];
        uint8_t ok = ok3[a];
        if (!ok) {
            float nv = 0;
            int cnt = 0;
            if (i > 0 && v[c - 1]) { nv = (float)((double)nv + f[c - 1]); ++cnt; }
            if (j > 0 && v[c - sy]) { nv = (float)((double)nv + f[c - sy]); ++cnt; }
            if (k > 0 && v[c - sz]) { nv = (float)((double)nv + f[c - sz]); ++cnt; }
            if (i < g.nx - 1 && v[c + 1]) { nv = (float)((double)nv + f[c + 1]); ++cnt; }
            if (j < g.ny - 1 && v[c + sy]) { nv = (float)((double)nv + f[c + sy]); ++cnt; }
            if (k < g.nz - 1 && v[c + sz]) { nv = (float)((double)nv + f[c + sz]); ++cnt; }
            if (cnt > 0) {
                f[c] = (double)(nv / cnt);
                ok = 1;
                act = 1;
            }
        }
        P.vout[a][c] = ok;
    }
    act_out[c] = act;
}
extern "C" int flip_extrapolate(flip_ctx *ctx, int iterations) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    for (int it = 0; it < iterations; ++it) {
        ExtrapPtrs P;
        for (int a = 0; a < 3; ++a) {
            P.f[a] = ctx->vel[0][a]; P.vin[a] = ctx->valid[0][a];
            P.vout[a] = ctx->valid[1][a];
        }
        k_extrapolate<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(flip_dims(ctx), P, ctx->active, ctx->active_back);
        FLIP_LAUNCH_CHECK();
        for (int a = 0; a < 3; ++a) std::swap(ctx->valid[0][a], ctx->valid[1][a]);
        std::swap(ctx->active, ctx->active_back);
    }
    return 0;
}

__global__ void k_add_gravity(long long B, const uint8_t *__restrict__ active, double *__restrict__ v, float gdt) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (active[c]) v[c] = v[c] - (double)gdt;
}
extern "C" int flip_add_forces(flip_ctx *ctx, float dt) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const float grav = 9.81f; // headers/FluidSimulator.h:19
    k_add_gravity<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->active, ctx->vel[0][1], grav * dt);
    FLIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int flip_update_prev(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    for (int a = 0; a < 3; ++a)
        FLIP_CHECK(hipMemcpyAsync(ctx->vel[2][a], ctx->vel[0][a], (size_t)ctx->B * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}
__global__ void k_diff(long long B, const double *__restrict__ f0, const double *__restrict__ f1, const double *__restrict__ f2,
                       const double *__restrict__ p0, const double *__restrict__ p1, const double *__restrict__ p2,
                       double *__restrict__ d0, double *__restrict__ d1, double *__restrict__ d2) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    d0[c] = f0[c] - p0[c];
    d1[c] = f1[c] - p1[c];
    d2[c] = f2[c] - p2[c];
}
extern "C" int flip_update_diff(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    k_diff<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2],
                                                                    ctx->vel[2][0], ctx->vel[2][1], ctx->vel[2][2], ctx->vel[3][0],
                                                                    ctx->vel[3][1], ctx->vel[3][2]);
    FLIP_LAUNCH_CHECK();
    return 0;
}

__global__ void k_vel_max(long long B, const uint8_t *__restrict__ active, const double *__restrict__ u,
                          const double *__restrict__ v, const double *__restrict__ w, double *__restrict__ out, int nblk) {
    __shared__ double scratch[4];
    double m0 = -1e300, m1 = -1e300, m2 = -1e300;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < B; c += (long long)gridDim.x * blockDim.x)
        if (active[c]) { m0 = fmax(m0, u[c]); m1 = fmax(m1, v[c]); m2 = fmax(m2, w[c]); }
    m0 = d_block_max(m0, scratch);
    m1 = d_block_max(m1, scratch);
    m2 = d_block_max(m2, scratch);
    if (threadIdx.x == 0) { out[blockIdx.x] = m0; out[nblk + blockIdx.x] = m1; out[2 * nblk + blockIdx.x] = m2; }
}
extern "C" int flip_velocity_max(flip_ctx *ctx, double out_max[3]) {
    FLIP_REQUIRE(ctx && out_max, "null argument");
    FLIP_CHECK(hipSetDevice(ctx->device));
    const int nblk = 1024;
    k_vel_max<<<nblk, FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->active, ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2], ctx->d_vmax, nblk);
    FLIP_LAUNCH_CHECK();
    std::vector<double> h(3 * nblk);
    FLIP_CHECK(hipMemcpyAsync(h.data(), ctx->d_vmax, sizeof(double) * 3 * nblk, hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < 3; ++a) {
        double m = -1e300;
        for (int b = 0; b < nblk; ++b) m = h[a * nblk + b] > m ? h[a * nblk + b] : m;
        out_max[a] = m == -1e300 ? 0.0 : m;
    }
    return 0;
}

// The same reduction for callers that must not allocate (tight serving loops, harnesses that forbid hipMalloc): the
// per-workgroup partial maxima live in a file-scope device array instead of memory the library allocated, and the host
// side uses no container.  It is not a call for a stream under graph capture: the result goes back with a copy into the
// caller's (pageable) host memory and a stream synchronise, neither of which a capture allows.  The array's name is only a host-side handle: its device address is
// looked up once per device with hipGetSymbolAddress and kept; every GPU holds its own copy at its own address.
#define FLIP_STATIC_VMAX_BLOCKS 4096          // workgroups at most
#define FLIP_STATIC_VMAX_CELLS_PER_BLOCK 16384 // 256 threads x 64 cells
#define FLIP_STATIC_VMAX_MAX_DEVICES 64
// [3][FLIP_STATIC_VMAX_BLOCKS] partial maxima, then the 3 results: 96 KB + 256 B per GPU, held until the process ends
static __device__ __align__(256) double flip_static_vmax_scratch[3 * FLIP_STATIC_VMAX_BLOCKS + 32];
static double *g_static_vmax_ptr[FLIP_STATIC_VMAX_MAX_DEVICES] = {};

__global__ void __launch_bounds__(FLIP_BLOCK) k_vel_max_static(long long B, const uint8_t *__restrict__ active, const double *__restrict__ u,
                                                              const double *__restrict__ v, const double *__restrict__ w,
                                                              double *__restrict__ part) {
    __shared__ double scratch[4];
    double m0 = -1e300, m1 = -1e300, m2 = -1e300;
    const long long lo = (long long)blockIdx.x * FLIP_STATIC_VMAX_CELLS_PER_BLOCK;
    const long long hi = lo + FLIP_STATIC_VMAX_CELLS_PER_BLOCK < B ? lo + FLIP_STATIC_VMAX_CELLS_PER_BLOCK : B;
    for (long long c = lo + threadIdx.x; c < hi; c += FLIP_BLOCK)
        if (active[c]) { m0 = fmax(m0, u[c]); m1 = fmax(m1, v[c]); m2 = fmax(m2, w[c]); }
    m0 = d_block_max(m0, scratch);
    m1 = d_block_max(m1, scratch);
    m2 = d_block_max(m2, scratch);
    // every workgroup of the launch writes its three slots, so nothing a previous call left behind is ever read
    if (threadIdx.x == 0) {
        part[blockIdx.x] = m0;
        part[FLIP_STATIC_VMAX_BLOCKS + blockIdx.x] = m1;
        part[2 * FLIP_STATIC_VMAX_BLOCKS + blockIdx.x] = m2;
    }
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_vel_max_static_final(int nblk, double *__restrict__ part) {
    __shared__ double scratch[4];
    for (int a = 0; a < 3; ++a) {
        double m = -1e300;
        for (int q = threadIdx.x; q < nblk; q += FLIP_BLOCK) m = fmax(m, part[a * FLIP_STATIC_VMAX_BLOCKS + q]);
        m = d_block_max(m, scratch);
        if (threadIdx.x == 0) part[3 * FLIP_STATIC_VMAX_BLOCKS + a] = m == -1e300 ? 0.0 : m;
    }
}

extern "C" int flip_velocity_max_static(flip_ctx *ctx, double out_max[3]) {
    FLIP_REQUIRE(ctx && out_max, "flip_velocity_max_static: null argument");
    FLIP_REQUIRE(ctx->B <= (int64_t)FLIP_STATIC_VMAX_BLOCKS * FLIP_STATIC_VMAX_CELLS_PER_BLOCK,
                 "flip_velocity_max_static: the box has more than 67108864 cells, the limit of the static workspace "
                 "(use flip_velocity_max)");
    FLIP_REQUIRE(ctx->device >= 0 && ctx->device < FLIP_STATIC_VMAX_MAX_DEVICES, "flip_velocity_max_static: device index beyond 63");
    FLIP_CHECK(hipSetDevice(ctx->device));
    double *part = g_static_vmax_ptr[ctx->device];
    if (!part) {
        // first call on this device, with this device current: the address of this GPU's copy of the array
        FLIP_CHECK(hipGetSymbolAddress((void **)&part, HIP_SYMBOL(flip_static_vmax_scratch)));
        FLIP_REQUIRE(part != nullptr, "flip_velocity_max_static: no device address for the static workspace");
        g_static_vmax_ptr[ctx->device] = part;
    }
    const int nblk = (int)((ctx->B + FLIP_STATIC_VMAX_CELLS_PER_BLOCK - 1) / FLIP_STATIC_VMAX_CELLS_PER_BLOCK);
    k_vel_max_static<<<nblk, FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->active, ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2], part);
    FLIP_LAUNCH_CHECK();
    k_vel_max_static_final<<<1, FLIP_BLOCK, 0, ctx->stream>>>(nblk, part);
    FLIP_LAUNCH_CHECK();
    FLIP_CHECK(hipMemcpyAsync(out_max, part + 3 * FLIP_STATIC_VMAX_BLOCKS, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// constrain_velocity: src/FluidSimulator.cpp:1180-1247 (container) and :1249-1321 (rigid body).  On every face
// whose solid fraction is exactly 1 the sampled velocity loses its component along the level set's normal (for
// the body it takes the body's normal velocity instead).  Restates openvdb::tools::gradient (second-order central
// differences scaled to world units, float), the BoxSampler on that Vec3f grid, Vec3::normalize (multiplies by
// the reciprocal length, leaves a vector shorter than 1e-7 alone) and Vec3::projection (eps 1e-10 on the squared
// length).  Reads the front buffer, writes the back buffer, then the buffers swap as in the reference.
// The reference's call also switches on the voxels of the weight grids' topology in velFront; no output of the path
// depends on that and it is not reproduced.

struct ConstrainArgs {
    GridDims g;
    const float *phi;
    float phi_bg, dx;
    const float *wu, *wv, *ww;
    const double *u, *v, *w;
    double *ou, *ov, *ow;
    int has_obj;
    double vt[3], va[3];
};
__device__ __forceinline__ float d_lerpf_half(float a, float b) {
    const double tmp = (double)(b - a) * 0.5;
    return a + (float)tmp;
}
__device__ void d_level_set_gradient(const ConstrainArgs &A, int i, int j, int k, float out[3]) {
    const double inv_dx = 1.0 / (double)A.dx;
    const float dxp = (d_getf(A.g, A.phi, i + 1, j, k, A.phi_bg) - d_getf(A.g, A.phi, i - 1, j, k, A.phi_bg)) * 0.5f;
    const float dyp = (d_getf(A.g, A.phi, i, j + 1, k, A.phi_bg) - d_getf(A.g, A.phi, i, j - 1, k, A.phi_bg)) * 0.5f;
    const float dzp = (d_getf(A.g, A.phi, i, j, k + 1, A.phi_bg) - d_getf(A.g, A.phi, i, j, k - 1, A.phi_bg)) * 0.5f;
    out[0] = (float)((double)dxp * inv_dx);
    out[1] = (float)((double)dyp * inv_dx);
    out[2] = (float)((double)dzp * inv_dx);
}
__device__ double d_constrained_component(const ConstrainArgs &A, int i, int j, int k, int axis) {
    int lo[3] = { i, j, k };
    lo[axis] -= 1;
    float ga[3], gb[3], nrm[3];
    d_level_set_gradient(A, lo[0], lo[1], lo[2], ga);
    d_level_set_gradient(A, i, j, k, gb);
    for (int a = 0; a < 3; ++a) nrm[a] = d_lerpf_half(ga[a], gb[a]);
    if (!(nrm[0] == 0.f && nrm[1] == 0.f && nrm[2] == 0.f)) {
        const float len = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (!(fabsf(len) <= 1e-7f)) {
            const float r = 1.0f / len;
            for (int a = 0; a < 3; ++a) nrm[a] = nrm[a] * r;
        }
    }
    double p[3] = { (double)(A.g.ox + i), (double)(A.g.oy + j), (double)(A.g.oz + k) };
    p[axis] -= 0.5;
    const double vel[3] = { d_box_sample(A.g, A.u, p[0] + 0.5, p[1], p[2]), d_box_sample(A.g, A.v, p[0], p[1] + 0.5, p[2]),
                            d_box_sample(A.g, A.w, p[0], p[1], p[2] + 0.5) };
    const double n[3] = { (double)nrm[0], (double)nrm[1], (double)nrm[2] };
    const double l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    double proj = 0.0;
    if (!(fabs(l) <= 1e-10)) proj = n[axis] * ((vel[0] * n[0] + vel[1] * n[1] + vel[2] * n[2]) * (1.0 / l));
    double out = vel[axis] - proj;
    if (A.has_obj) {
        // solidVel = vel_trans + voxel * (vel_ang x (coord - e/2)), src/FluidSimulator.cpp:1287
        const double vs = (double)A.dx;
        const double c[3] = { vs * A.va[1] * p[2] - vs * A.va[2] * p[1], vs * A.va[2] * p[0] - vs * A.va[0] * p[2],
                              vs * A.va[0] * p[1] - vs * A.va[1] * p[0] };
        const double sv[3] = { A.vt[0] + c[0], A.vt[1] + c[1], A.vt[2] + c[2] };
        out = out + (n[0] * sv[0] + n[1] * sv[1] + n[2] * sv[2]) * n[axis];
    }
    return out;
}
__global__ void k_constrain(ConstrainArgs A) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.g.B) return;
    int i, j, k;
    d_ijk(A.g, c, i, j, k);
    A.ou[c] = A.wu[c] == 1.f ? d_constrained_component(A, i, j, k, 0) : A.u[c];
    A.ov[c] = A.wv[c] == 1.f ? d_constrained_component(A, i, j, k, 1) : A.v[c];
    A.ow[c] = A.ww[c] == 1.f ? d_constrained_component(A, i, j, k, 2) : A.w[c];
}

extern "C" int flip_constrain_velocity(flip_ctx *ctx, int rigid_body) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!rigid_body || ctx->has_obj, "flip_constrain_velocity: no rigid body attached");
    FLIP_CHECK(hipSetDevice(ctx->device));
    ConstrainArgs A;
    A.g = flip_dims(ctx);
    A.dx = ctx->g.dx;
    A.has_obj = rigid_body ? 1 : 0;
    if (rigid_body) {
        A.phi = ctx->phi_obj; A.phi_bg = ctx->phi_obj_bg;
        A.wu = ctx->owgt[0]; A.wv = ctx->owgt[1]; A.ww = ctx->owgt[2];
    } else {
        A.phi = ctx->phi_solid; A.phi_bg = ctx->g.phi_solid_bg;
        A.wu = ctx->wgt[0]; A.wv = ctx->wgt[1]; A.ww = ctx->wgt[2];
    }
    A.u = ctx->vel[0][0]; A.v = ctx->vel[0][1]; A.w = ctx->vel[0][2];
    A.ou = ctx->vel[1][0]; A.ov = ctx->vel[1][1]; A.ow = ctx->vel[1][2];
    for (int a = 0; a < 3; ++a) { A.vt[a] = ctx->obj_vel[a]; A.va[a] = ctx->obj_vel[3 + a]; }
    k_constrain<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(A);
    FLIP_LAUNCH_CHECK();
    for (int a = 0; a < 3; ++a) std::swap(ctx->vel[0][a], ctx->vel[1][a]);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Projections (src/FluidSimulator.cpp:1017-1102 and :1104-1178): every cell of the box; pressure
// narrowed to float and the gradient evaluated in float, as the reference does.

template <int DISPLACEMENT>
__global__ void k_project(GridDims g, float dt, float rho, float vs, float phi_bg, const uint8_t *__restrict__ label,
                          const double *__restrict__ p, const float *__restrict__ phi, const float *__restrict__ wu,
                          const float *__restrict__ wv, const float *__restrict__ ww, const double *__restrict__ u,
                          const double *__restrict__ v, const double *__restrict__ w, double *__restrict__ o0,
                          double *__restrict__ o1, double *__restrict__ o2, uint8_t *__restrict__ val0,
                          uint8_t *__restrict__ val1, uint8_t *__restrict__ val2, uint8_t *__restrict__ active) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    int i, j, k;
    d_ijk(g, c, i, j, k);
    const long long sy = g.nx, sz = (long long)g.nx * g.ny;
    const float pc = label[c] ? (float)p[c] : 0.f;
    const float p_i = (i > 0 && label[c - 1]) ? (float)p[c - 1] : 0.f;
    const float p_j = (j > 0 && label[c - sy]) ? (float)p[c - sy] : 0.f;
    const float p_k = (k > 0 && label[c - sz]) ? (float)p[c - sz] : 0.f;
    const float phi_c = phi[c];
    const float phi_i = i > 0 ? phi[c - 1] : phi_bg;
    const float phi_j = j > 0 ? phi[c - sy] : phi_bg;
    const float phi_k = k > 0 ? phi[c - sz] : phi_bg;
    double r0, r1, r2;
    if (DISPLACEMENT) { r0 = 0; r1 = 0; r2 = 0; }
    else { r0 = u[c]; r1 = v[c]; r2 = w[c]; }
    uint8_t f0 = 0, f1 = 0, f2 = 0;
    if ((1 - wu[c]) > 0 && (phi_c < 0 || phi_i < 0)) {
        float theta = 1;
        if (phi_c >= 0 || phi_i >= 0) theta = d_theta(phi_i, phi_c);
        if (DISPLACEMENT) r0 = (double)(-dt * dt / rho * (pc - p_i) / vs / theta);
        else r0 -= (double)(dt / rho * (pc - p_i) / vs / theta);
        f0 = 1;
    }
    if ((1 - wv[c]) > 0 && (phi_c < 0 || phi_j < 0)) {
        float theta = 1;
        if (phi_c >= 0 || phi_j >= 0) theta = d_theta(phi_j, phi_c);
        if (DISPLACEMENT) r1 = (double)(-dt * dt / rho * (pc - p_j) / vs / theta);
        else r1 -= (double)(dt / rho * (pc - p_j) / vs / theta);
        f1 = 1;
    }
    if ((1 - ww[c]) > 0 && (phi_c < 0 || phi_k < 0)) {
        float theta = 1;
        if (phi_c >= 0 || phi_k >= 0) theta = d_theta(phi_k, phi_c);
        if (DISPLACEMENT) r2 = (double)(-dt * dt / rho * (pc - p_k) / vs / theta);
        else r2 -= (double)(dt / rho * (pc - p_k) / vs / theta);
        f2 = 1;
    }
    if (DISPLACEMENT) {
        o0[c] = r0 / (double)vs; o1[c] = r1 / (double)vs; o2[c] = r2 / (double)vs;
    } else {
        o0[c] = r0; o1[c] = r1; o2[c] = r2;
        val0[c] = f0; val1[c] = f1; val2[c] = f2;
        active[c] = 1; // every cell of the box is written (src/FluidSimulator.cpp:1092)
    }
}

int flip_pcg_coupling_jp(flip_ctx *ctx, double jp[6]); // flip_pcg.hip: J * pressure for the rigid body

extern "C" int flip_project_velocity(flip_ctx *ctx, float dt) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    k_project<0><<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(
        flip_dims(ctx), dt, ctx->g.rest_density, ctx->g.dx, ctx->g.phi_fluid_bg, ctx->label, ctx->x, ctx->phi_fluid, ctx->wgt[0],
        ctx->wgt[1], ctx->wgt[2], ctx->vel[0][0], ctx->vel[0][1], ctx->vel[0][2], ctx->vel[1][0], ctx->vel[1][1], ctx->vel[1][2],
        ctx->valid[0][0], ctx->valid[0][1], ctx->valid[0][2], ctx->active);
    FLIP_LAUNCH_CHECK();
    for (int a = 0; a < 3; ++a) std::swap(ctx->vel[0][a], ctx->vel[1][a]);
    // on a slab context the sum J p spans every rank's rows: the caller completes it (flip_solid_coupling_partial)
    if (ctx->has_obj && ctx->sys_coupled && !ctx->slab_on) { // V += dt * Minv * (J p)   (src/FluidSimulator.cpp:1095-1100)
        double jp[6];
        if (int rc = flip_pcg_coupling_jp(ctx, jp)) return rc;
        for (int a = 0; a < 6; ++a) ctx->obj_vel[a] = ctx->obj_vel[a] + (double)dt * (ctx->obj_minv[a] * jp[a]);
    }
    return 0;
}

extern "C" int flip_project_displacement(flip_ctx *ctx, float dt) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_CHECK(hipSetDevice(ctx->device));
    k_project<1><<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(
        flip_dims(ctx), dt, ctx->g.rest_density, ctx->g.dx, ctx->g.phi_fluid_bg, ctx->label, ctx->x, ctx->phi_fluid, ctx->wgt[0],
        ctx->wgt[1], ctx->wgt[2], nullptr, nullptr, nullptr, ctx->disp[0], ctx->disp[1], ctx->disp[2], nullptr, nullptr, nullptr,
        nullptr);
    FLIP_LAUNCH_CHECK();
    return 0;
}
