// flip_pcg.hip -- matrix-free assembly of the two pressure systems and the conjugate-gradient solver.
//
// The system of src/FluidSimulator.cpp:654-816 / 832-990 is never stored as a sparse matrix.  Per cell
// the library keeps: three float face terms (1 - W) * scale at the low x/y/z faces (the high faces of
// a cell are the low faces of its neighbours), one flag byte (row present, which of the six
// off-diagonals exist), the diagonal and its Jacobi inverse in double, and the right-hand side.
// Vectors are dense over the box and zero off the rows.  Work is organised in chunks of 256
// consecutive cells; only chunks that contain a row are swept (compact list built at assembly).
//
// One CG iteration is two kernels:
//   k_pcg_direction: beta from the reduced partials; p = z + beta p (evaluated on the fly at the 7
//                    stencil points), Ap = A p, partial p.Ap
//   k_pcg_update   : alpha; x += alpha p; r -= alpha Ap; z = Dinv r; partials r.r and r.z
// Dot products are reduced without atomics: every workgroup writes one partial per quantity, and
// every workgroup of the next kernel sums all partials in the same fixed order, so all workgroups
// hold the same scalar and the result is reproducible run to run.
#include "flip_internal.h"

#include <cstdlib>

#include <cfloat>

// ---------------------------------------------------------------------------------------------
// Assembly

struct BuildArgs {
    GridDims g;
    int kind;
    float dt, rho, vs, scale, fbg, sbg;
    double s2;
    const uint8_t *label;
    const float *phi_f, *phi_s, *wu, *wv, *ww, *density;
    const double *u, *v, *w;
    float *tx, *ty, *tz;
    uint8_t *flags;
    double *diag, *invdiag, *rhs;
    float *cgx, *cgy, *cgz;   // CG view (see flip_ctx)
    double *cg_diag, *cg_inv;
    int32_t *chunk_flag;
    // rigid-body coupling (kind 0 only)
    int coupled;
    int drop_null;
    int eliminate;
    int32_t *counters; // [4] dependent rows, [5] dependent rows with a non-zero rigid-body entry
    const float *ou, *ov, *ow;
    double com[3], minv[6], vel[6];
    double *J[6];
    // by-products for the solver's set-up, each a launch of its own otherwise: the mark of the CG rows among the planes
    // [own_lo, own_hi) this context owns (input of the row list's scan), the cleared "regular row" bytes, and the float
    // copies of the CG diagonal and its inverse (null on a slab context, whose diagonal is masked after the assembly)
    long long own_lo, own_hi;
    int32_t *cg_mark;
    uint8_t *reg;
    float *diag_f, *inv_f;
};

__global__ void k_build_system(BuildArgs A) {
    const GridDims &g = A.g;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int is_row = 0, dep = 0, dep_coupled = 0;
    if (c < g.B) {
        int i, j, k;
        d_ijk(g, c, i, j, k);
        const float w_lo[3] = { A.wu[c], A.wv[c], A.ww[c] };
        A.tx[c] = (1 - w_lo[0]) * A.scale;
        A.ty[c] = (1 - w_lo[1]) * A.scale;
        A.tz[c] = (1 - w_lo[2]) * A.scale;
        uint8_t fl = 0;
        double dg = 0.0, inv = 1.0, rh = 0.0;
        double Jr[6] = { 0, 0, 0, 0, 0, 0 };
        if (A.label[c]) {
            is_row = 1;
            fl = FLIP_FLAG_ROW;
            const float phi_c = A.phi_f[c];
            double diagonal = 0, r = 0;
            float wsum = 0;
            const int di[6] = { -1, 1, 0, 0, 0, 0 }, dj[6] = { 0, 0, -1, 1, 0, 0 }, dk[6] = { 0, 0, 0, 0, -1, 1 };
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int ni = i + di[f], nj = j + dj[f], nk = k + dk[f];
                const bool in = d_inside(g, ni, nj, nk);
                const long long nb = in ? d_lid(g, ni, nj, nk) : 0;
                const float phi_n = in ? A.phi_f[nb] : A.fbg;
                const float sphi_n = in ? A.phi_s[nb] : A.sbg;
                const float *wa = f < 2 ? A.wu : (f < 4 ? A.wv : A.ww);
                const float wval = (f & 1) ? (in ? wa[nb] : 0.f) : w_lo[f >> 1];
                const float weight = 1 - wval;
                const double term = (double)(weight * A.scale);
                if (phi_n < 0) {
                    // the off-diagonal multiplies a vector entry only where the neighbour has a row
                    if (in && A.label[nb]) fl |= (uint8_t)(1u << f);
                    diagonal += term;
                } else if (sphi_n > 0) {
                    const float th = (f & 1) ? d_theta(phi_c, phi_n) : d_theta(phi_n, phi_c);
                    diagonal += term * ((double)1.f / (double)th);
                }
                if (A.kind == 0) {
                    const double *va = f < 2 ? A.u : (f < 4 ? A.v : A.w);
                    if (f & 1) r -= (double)weight * (in ? va[nb] : 0.0);
                    else r += (double)weight * va[c];
                } else {
                    wsum = wsum + weight;
                }
            }
            if (A.kind == 0) {
                dg = diagonal * A.s2;
                rh = r * A.s2;
            } else {
                dg = diagonal;
                float rr = -1 / A.dt * (1 - d_clampf(A.density[c] / A.rho, 0.5f, 1.5f));
                if (wsum == 0) rr = 0;
                rh = (double)rr;
            }
            double dfull = dg;
            if (A.coupled) { // SolidObject::constructJMatrix (src/SolidObject.cpp:159-185)
                const float vs2 = A.vs * A.vs;
                const float du = d_getf(g, A.ou, i + 1, j, k, 0.f) - A.ou[c];
                const float dv = d_getf(g, A.ov, i, j + 1, k, 0.f) - A.ov[c];
                const float dw = d_getf(g, A.ow, i, j, k + 1, 0.f) - A.ow[c];
                const double rx = (double)(g.ox + i) - A.com[0], ry = (double)(g.oy + j) - A.com[1],
                             rz = (double)(g.oz + k) - A.com[2];
                Jr[0] = (double)(vs2 * du);
                Jr[1] = (double)(vs2 * dv);
                Jr[2] = (double)(vs2 * dw);
                Jr[3] = (double)vs2 * ((double)(-A.vs) * rz * (double)dv + (double)A.vs * ry * (double)dw);
                Jr[4] = (double)vs2 * ((double)(-A.vs) * rx * (double)dw + (double)A.vs * rz * (double)du);
                Jr[5] = (double)vs2 * ((double)(-A.vs) * ry * (double)du + (double)A.vs * rx * (double)dv);
                double s = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    s += Jr[a] * A.vel[a];                                  // rhs -= J^T V        (:815)
                    dfull += (double)A.dt * A.minv[a] * Jr[a] * Jr[a];      // diag of dt J^T Minv J (:813)
                }
                rh = rh - s;
            }
            inv = dfull != 0 ? 1.0 / dfull : 1.0; // Eigen DiagonalPreconditioner
            if (A.drop_null && dfull == 0) rh = 0.0; // all-zero row: nothing to solve for
            if (A.eliminate && !(phi_c < 0)) {
                // A labelled cell that is not liquid enters no other row (an off-diagonal needs phi < 0 at the
                // neighbour), so the matrix is block lower triangular: [liquid rows | 0 ; C | diag].
                fl |= FLIP_FLAG_DEP;
                dep = 1;
                if (A.coupled && (Jr[0] != 0 || Jr[1] != 0 || Jr[2] != 0 || Jr[3] != 0 || Jr[4] != 0 || Jr[5] != 0)) dep_coupled = 1;
            }
        }
        // CG view: rows that take part in the CG iteration; the coupling of two such rows is symmetric
        const bool cg = (fl & (FLIP_FLAG_ROW | FLIP_FLAG_DEP)) == FLIP_FLAG_ROW;
        A.cgx[c] = (cg && (fl & 1)) ? (1 - w_lo[0]) * A.scale : 0.f;
        A.cgy[c] = (cg && (fl & 4)) ? (1 - w_lo[1]) * A.scale : 0.f;
        A.cgz[c] = (cg && (fl & 16)) ? (1 - w_lo[2]) * A.scale : 0.f;
        A.cg_diag[c] = cg ? dg : 0.0;
        A.cg_inv[c] = cg ? inv : 0.0;
        A.cg_mark[c] = (cg && c >= A.own_lo && c < A.own_hi) ? 1 : 0;
        if (c == 0) A.cg_mark[g.B] = 0;   // terminator of the scan
        A.reg[c] = 0;
        if (A.diag_f) { A.diag_f[c] = (float)(cg ? dg : 0.0); A.inv_f[c] = (float)(cg ? inv : 0.0); }
        A.flags[c] = fl;
        A.diag[c] = dg;
        A.invdiag[c] = inv;
        A.rhs[c] = rh;
        if (A.coupled)
#pragma unroll
            for (int a = 0; a < 6; ++a) A.J[a][c] = Jr[a];
    }
    // one atomic per wave for the two counters (dependent rows cluster along the free surface: per-row atomics on one
    // address serialise)
    if (A.eliminate) {
        const unsigned long long m = __ballot(dep), mc = __ballot(dep_coupled);
        if ((threadIdx.x & (FLIP_WAVE - 1)) == 0) {
            if (m) atomicAdd(A.counters + 4, (int)__popcll(m));
            if (mc) atomicAdd(A.counters + 5, (int)__popcll(mc));
        }
    }
    const int any = __syncthreads_or(is_row);
    if (threadIdx.x == 0) A.chunk_flag[blockIdx.x] = any ? 1 : 0;
}

__global__ void k_compact_chunks(int n_chunks, const int32_t *__restrict__ flag_scan, int32_t *__restrict__ list,
                                 int32_t *__restrict__ count) {
    // flag_scan holds the exclusive scan of the chunk flags over n_chunks+1 entries; chunk q is active
    // when scan[q+1] > scan[q]; the total goes to *count
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_chunks) return;
    if (q == 0) *count = flag_scan[n_chunks];
    if (flag_scan[q + 1] > flag_scan[q]) list[flag_scan[q]] = q;
}

// [own_lo, own_hi): cell ids of the planes this context owns (the whole box unless flip_set_slab narrowed it)
__global__ void k_mark_cg_rows(long long B, long long own_lo, long long own_hi, const uint8_t *__restrict__ flags,
                               int32_t *__restrict__ mark) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > B) return;
    mark[c] = (c >= own_lo && c < own_hi && (flags[c] & (FLIP_FLAG_ROW | FLIP_FLAG_DEP)) == FLIP_FLAG_ROW) ? 1 : 0;
}
__global__ void k_compact_cg_rows(long long B, long long own_lo, long long own_hi, const uint8_t *__restrict__ flags,
                                  const int32_t *__restrict__ scan, int32_t *__restrict__ rows, int32_t *__restrict__ count) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (c == 0) *count = scan[B];
    if (c >= own_lo && c < own_hi && (flags[c] & (FLIP_FLAG_ROW | FLIP_FLAG_DEP)) == FLIP_FLAG_ROW) rows[scan[c]] = (int32_t)c;
}
// Slab mode: [reg_lo, reg_hi) are the cells the local preconditioner covers (the owned planes, widened by the
// overlap).  Cells outside it are not rows of anything local (diagonal and Jacobi inverse zero), and the
// preconditioner sees no coupling across the two faces of the region (stored with the cell on the high side).
__global__ void k_slab_mask(long long B, long long sz, long long reg_lo, long long reg_hi, const float *__restrict__ cgz,
                            float *__restrict__ mg_cz, double *__restrict__ cg_diag, double *__restrict__ cg_inv) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const bool inside = c >= reg_lo && c < reg_hi;
    if (!inside) { cg_diag[c] = 0.0; cg_inv[c] = 0.0; }
    const bool on_face = (c >= reg_lo && c < reg_lo + sz) || (c >= reg_hi && c < reg_hi + sz);
    mg_cz[c] = (inside && !on_face) ? cgz[c] : 0.f;
}
__global__ void k_mark_region_rows(long long B, long long lo, long long hi, const double *__restrict__ cg_inv, int32_t *__restrict__ mark) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > B) return;
    mark[c] = (c >= lo && c < hi && cg_inv[c] != 0.0) ? 1 : 0;
}
__global__ void k_compact_region_rows(long long B, const int32_t *__restrict__ scan, int32_t *__restrict__ rows) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (scan[c + 1] > scan[c]) rows[scan[c]] = (int32_t)c;
}

__global__ void k_mark_regular(const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows, long long sy, long long sz, float S,
                               const float *__restrict__ cx, const float *__restrict__ cy, const float *__restrict__ cz,
                               uint8_t *__restrict__ reg) {
    const long long n = *n_rows;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        const long long c = rows[q];
        reg[c] = (cx[c] == S && cx[c + 1] == S && cy[c] == S && cy[c + sy] == S && cz[c] == S && cz[c + sz] == S) ? 1 : 0;
    }
}

// The slab solve's arrays of the ACTIVE system set, for the slab and overlap the context is set to (both sets need them
// when two slab solves run side by side, flip_pair_begin).
static int slab_alloc_state(flip_ctx *ctx) {
    const size_t plane = (size_t)ctx->g.nx * ctx->g.ny;
    if (ctx->slab_on && !ctx->mg_cz) {
        // padded like the other coupling arrays: the stencil kernels read one plane beyond either end
        float *base = nullptr;
        FLIP_CHECK(hipMalloc((void **)&base, (size_t)(ctx->B + 2 * ctx->pad) * sizeof(float)));
        FLIP_CHECK(flip_zero_new(base, (size_t)(ctx->B + 2 * ctx->pad) * sizeof(float)));
        ctx->mg_cz = base + ctx->pad;
    }
    if (!ctx->dist_sums) {
        FLIP_CHECK(hipMalloc((void **)&ctx->dist_sums, 16 * sizeof(double)));
        FLIP_CHECK(flip_zero_new(ctx->dist_sums, 16 * sizeof(double)));
    }
    const int32_t planes = ctx->slab_overlap;
    if (planes != ctx->dist_recv_planes) {
        for (int s = 0; s < 2; ++s)
            if (ctx->dist_recv[s]) { FLIP_CHECK(hipFree(ctx->dist_recv[s])); ctx->dist_recv[s] = nullptr; }
        ctx->dist_recv_planes = 0;
    }
    if (planes > 0 && ctx->slab_on) {
        if (!ctx->mg_rows) FLIP_CHECK(hipMalloc((void **)&ctx->mg_rows, (size_t)ctx->B * 4));
        for (int s = 0; s < 2; ++s)
            if (!ctx->dist_recv[s]) {
                FLIP_CHECK(hipMalloc((void **)&ctx->dist_recv[s], (size_t)(planes + 1) * plane * 8));
                FLIP_CHECK(flip_zero_new(ctx->dist_recv[s], (size_t)(planes + 1) * plane * 8));
            }
        ctx->dist_recv_planes = planes;
    }
    if (ctx->slab_weighted && !ctx->dist_rw) {
        double *base = nullptr;
        FLIP_CHECK(hipMalloc((void **)&base, (size_t)(ctx->B + 2 * ctx->pad) * sizeof(double)));
        FLIP_CHECK(flip_zero_new(base, (size_t)(ctx->B + 2 * ctx->pad) * sizeof(double)));
        ctx->dist_rw = base + ctx->pad;
    }
    return 0;
}

extern "C" int flip_set_slab(flip_ctx *ctx, int32_t k0, int32_t k1) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(k0 >= 0 && k1 > k0 && k1 <= ctx->g.nz, "flip_set_slab: need 0 <= k0 < k1 <= nz");
    FLIP_REQUIRE(!ctx->pair_on, "flip_set_slab: not between flip_pair_begin and flip_pair_end");
    FLIP_CHECK(hipSetDevice(ctx->device));
    ctx->slab_on = !(k0 == 0 && k1 == ctx->g.nz);
    ctx->slab_k0 = k0;
    ctx->slab_k1 = k1;
    ctx->slab_overlap = 0;
    ctx->slab_weighted = false;
    if (int rc = slab_alloc_state(ctx)) return rc;
    ctx->sys_kind = -1; // a system assembled under another ownership is stale
    ctx->other.sys_kind = -1;
    return 0;
}

extern "C" int flip_set_slab_overlap(flip_ctx *ctx, int32_t planes) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->dist_sums, "flip_set_slab_overlap: call flip_set_slab first");
    FLIP_REQUIRE(planes >= 0, "flip_set_slab_overlap: the overlap cannot be negative");
    FLIP_REQUIRE(!ctx->dist_active && !ctx->pair_on, "flip_set_slab_overlap: a solve is in progress");
    // a face with a neighbour behind it has ghost planes; the overlap must fit into them
    FLIP_REQUIRE(ctx->slab_k0 == 0 || planes <= ctx->slab_k0, "flip_set_slab_overlap: more planes than the lower ghost zone holds");
    FLIP_REQUIRE(ctx->slab_k1 == ctx->g.nz || planes <= ctx->g.nz - ctx->slab_k1,
                 "flip_set_slab_overlap: more planes than the upper ghost zone holds");
    FLIP_CHECK(hipSetDevice(ctx->device));
    ctx->slab_overlap = ctx->slab_on ? planes : 0;
    ctx->slab_weighted = false;
    if (ctx->slab_overlap > 0) {
        const char *e = getenv("FLIP_SLAB_WEIGHTED");
        ctx->slab_weighted = !(e && atoi(e) == 0);
    }
    if (int rc = slab_alloc_state(ctx)) return rc;
    ctx->sys_kind = -1;
    ctx->other.sys_kind = -1;
    return 0;
}

extern "C" int flip_set_system_options(flip_ctx *ctx, int drop_null_rows, int eliminate_dependent_rows) {
    FLIP_REQUIRE(ctx, "null context");
    ctx->drop_null_rows = drop_null_rows != 0;
    ctx->eliminate_dep = eliminate_dependent_rows != 0;
    return 0;
}

extern "C" int flip_set_warm_start(flip_ctx *ctx, int enabled) {
    FLIP_REQUIRE(ctx, "null context");
    ctx->warm_start = enabled != 0;
    return 0;
}

extern "C" int flip_build_system(flip_ctx *ctx, int kind, float dt) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(kind == FLIP_SOLVE_DIVERGENCE || kind == FLIP_SOLVE_DENSITY, "flip_build_system: unknown kind");
    FLIP_REQUIRE(dt > 0.f, "flip_build_system: dt must be positive");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BuildArgs A;
    memset(&A, 0, sizeof A);
    A.g = flip_dims(ctx);
    A.kind = kind;
    A.dt = dt;
    A.rho = ctx->g.rest_density;
    A.vs = ctx->g.dx;
    // float expressions of src/FluidSimulator.cpp:692 and :870
    A.scale = kind == 0 ? dt / (A.rho * A.vs) : dt / (A.rho * (A.vs * A.vs));
    A.s2 = kind == 0 ? (double)(A.vs * A.vs) : 1.0;
    A.fbg = ctx->g.phi_fluid_bg;
    A.sbg = ctx->g.phi_solid_bg;
    A.label = ctx->label;
    A.phi_f = ctx->phi_fluid; A.phi_s = ctx->phi_solid;
    A.wu = ctx->wgt[0]; A.wv = ctx->wgt[1]; A.ww = ctx->wgt[2];
    A.density = ctx->density;
    A.u = ctx->vel[0][0]; A.v = ctx->vel[0][1]; A.w = ctx->vel[0][2];
    A.tx = ctx->term[0]; A.ty = ctx->term[1]; A.tz = ctx->term[2];
    A.flags = ctx->flags; A.diag = ctx->diag; A.invdiag = ctx->invdiag; A.rhs = ctx->rhs;
    A.chunk_flag = ctx->chunk_flag;
    A.cgx = ctx->cg_c[0]; A.cgy = ctx->cg_c[1]; A.cgz = ctx->cg_c[2];
    A.cg_diag = ctx->cg_diag; A.cg_inv = ctx->cg_inv;
    A.coupled = (kind == 0 && ctx->has_obj) ? 1 : 0;
    A.drop_null = ctx->drop_null_rows ? 1 : 0;
    A.eliminate = ctx->eliminate_dep ? 1 : 0;
    A.counters = ctx->d_counts;
    ctx->sys_dep_coupled = 0;
    A.ou = ctx->owgt[0]; A.ov = ctx->owgt[1]; A.ow = ctx->owgt[2];
    for (int a = 0; a < 3; ++a) A.com[a] = ctx->obj_com[a];
    for (int a = 0; a < 6; ++a) { A.minv[a] = ctx->obj_minv[a]; A.vel[a] = ctx->obj_vel[a]; A.J[a] = ctx->J[a]; }
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const long long own_lo = ctx->slab_on ? ctx->slab_k0 * plane : 0, own_hi = ctx->slab_on ? ctx->slab_k1 * plane : ctx->B;
    A.own_lo = own_lo; A.own_hi = own_hi;
    A.cg_mark = ctx->cg_scan;
    A.reg = ctx->reg;
    // float copies of the diagonal for the V-cycle's level 0, once they exist (flip_mg_setup allocates them and fills them
    // itself the first time, and always on a slab context)
    A.diag_f = ctx->slab_on ? nullptr : ctx->mg_diag_f;
    A.inv_f = ctx->slab_on ? nullptr : ctx->mg_inv_f;
    ctx->mg_narrowed = A.diag_f != nullptr;
    FLIP_CHECK(hipMemsetAsync(ctx->chunk_flag + ctx->n_chunks, 0, 4, st));
    FLIP_CHECK(hipMemsetAsync(ctx->d_counts + 4, 0, 8, st));
    k_build_system<<<ctx->n_chunks, FLIP_BLOCK, 0, st>>>(A);
    FLIP_LAUNCH_CHECK();
    if (A.eliminate && A.coupled) {
        // The rank-6 term couples every row with a non-zero rigid-body entry to every other such row; if a
        // dependent row has one, the block-triangular structure is gone and the system is solved whole.
        int32_t cnt[2] = { 0, 0 };
        FLIP_CHECK(hipMemcpyAsync(cnt, ctx->d_counts + 4, 8, hipMemcpyDeviceToHost, st));
        FLIP_CHECK(hipStreamSynchronize(st));
        ctx->sys_dep_coupled = cnt[1];
        // A slab context keeps the eliminated system: whether to drop the elimination is a property of all slabs together,
        // and the slab solver has no other form; the caller asks flip_system_dependent_coupled on every rank and agrees.
        if (cnt[1] > 0 && !ctx->slab_on) {
            A.eliminate = 0;
            FLIP_CHECK(hipMemsetAsync(ctx->d_counts + 4, 0, 8, st));
            k_build_system<<<ctx->n_chunks, FLIP_BLOCK, 0, st>>>(A);
            FLIP_LAUNCH_CHECK();
        }
    }
    if (int rc = flip_exclusive_scan(ctx, ctx->chunk_flag, ctx->n_chunks + 1)) return rc;
    k_compact_chunks<<<flip_grid_blocks(ctx->n_chunks), FLIP_BLOCK, 0, st>>>(ctx->n_chunks, ctx->chunk_flag, ctx->chunk_list, ctx->d_counts + 1);
    FLIP_LAUNCH_CHECK();
    // compact list of the cells that are rows of the CG system, ascending cell id
    bool marked = true;   // k_build_system has left the marks in cg_scan
    if (ctx->slab_on) {
        const int ov = ctx->slab_overlap;
        const long long reg_lo = (ctx->slab_k0 > ov ? ctx->slab_k0 - ov : 0) * plane;
        const long long reg_hi = (ctx->slab_k1 + ov < ctx->g.nz ? ctx->slab_k1 + ov : ctx->g.nz) * plane;
        k_slab_mask<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(ctx->B, plane, reg_lo, reg_hi, ctx->cg_c[2], ctx->mg_cz,
                                                                    ctx->cg_diag, ctx->cg_inv);
        FLIP_LAUNCH_CHECK();
        if (ov > 0) {
            // rows of the preconditioner's region (ascending cell ids), the CG rows among them
            k_mark_region_rows<<<flip_grid_blocks(ctx->B + 1), FLIP_BLOCK, 0, st>>>(ctx->B, reg_lo, reg_hi, ctx->cg_inv, ctx->cg_scan);
            FLIP_LAUNCH_CHECK();
            if (int rc = flip_exclusive_scan(ctx, ctx->cg_scan, ctx->B + 1)) return rc;
            k_compact_region_rows<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(ctx->B, ctx->cg_scan, ctx->mg_rows);
            FLIP_LAUNCH_CHECK();
            if (int rc = flip_copy_i32(ctx, ctx->d_counts + 7, ctx->cg_scan + ctx->B, st)) return rc;
            marked = false;   // the region's row list went through cg_scan
        }
    }
    if (!marked) {
        k_mark_cg_rows<<<flip_grid_blocks(ctx->B + 1), FLIP_BLOCK, 0, st>>>(ctx->B, own_lo, own_hi, ctx->flags, ctx->cg_scan);
        FLIP_LAUNCH_CHECK();
    }
    if (int rc = flip_exclusive_scan(ctx, ctx->cg_scan, ctx->B + 1)) return rc;
    k_compact_cg_rows<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(ctx->B, own_lo, own_hi, ctx->flags, ctx->cg_scan, ctx->cg_rows, ctx->d_counts + 6);
    FLIP_LAUNCH_CHECK();
    // rows whose six couplings are the plain face term (1 - 0) * scale (k_build_system has cleared the bytes)
    ctx->reg_value = A.scale;
    if (A.eliminate && !ctx->slab_on) {
        k_mark_regular<<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(ctx->cg_rows, ctx->d_counts + 6, ctx->g.nx, plane, A.scale, ctx->cg_c[0],
                                                              ctx->cg_c[1], ctx->cg_c[2], ctx->reg);
        FLIP_LAUNCH_CHECK();
    }
    ctx->sys_kind = kind;
    ctx->sys_dt = dt;
    ctx->sys_s2 = A.s2;
    ctx->sys_coupled = A.coupled != 0;
    ctx->sys_eliminated = A.eliminate != 0;
    return 0;
}

extern "C" int flip_system_dependent_coupled(flip_ctx *ctx, int32_t *count) {
    FLIP_REQUIRE(ctx && count, "flip_system_dependent_coupled: null argument");
    FLIP_REQUIRE(ctx->sys_kind >= 0, "flip_system_dependent_coupled: no system assembled");
    *count = ctx->sys_dep_coupled;
    return 0;
}

// expands the compact representation into the 7-point stencil for the parity hook
__global__ void k_expand_stencil(GridDims g, double s2, const uint8_t *__restrict__ flags, const float *__restrict__ tx,
                                 const float *__restrict__ ty, const float *__restrict__ tz, double *__restrict__ off) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.B) return;
    const uint8_t fl = flags[c];
    const long long sy = g.nx, sz = (long long)g.nx * g.ny;
    off[0 * g.B + c] = (fl & 1) ? -((double)tx[c] * s2) : 0.0;
    off[1 * g.B + c] = (fl & 2) ? -((double)tx[c + 1] * s2) : 0.0;
    off[2 * g.B + c] = (fl & 4) ? -((double)ty[c] * s2) : 0.0;
    off[3 * g.B + c] = (fl & 8) ? -((double)ty[c + sy] * s2) : 0.0;
    off[4 * g.B + c] = (fl & 16) ? -((double)tz[c] * s2) : 0.0;
    off[5 * g.B + c] = (fl & 32) ? -((double)tz[c + sz] * s2) : 0.0;
}

extern "C" int flip_system_download(flip_ctx *ctx, double *off, double *diag, double *rhs) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->sys_kind >= 0, "flip_system_download: no system assembled");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t B = ctx->B;
    if (off) {
        double *tmp = nullptr;
        FLIP_CHECK(hipMalloc((void **)&tmp, (size_t)B * 6 * 8));
        k_expand_stencil<<<flip_grid_blocks(B), FLIP_BLOCK, 0, st>>>(flip_dims(ctx), ctx->sys_s2, ctx->flags, ctx->term[0],
                                                                    ctx->term[1], ctx->term[2], tmp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(off, tmp, (size_t)B * 6 * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(tmp);
        FLIP_CHECK(e);
    }
    if (diag) FLIP_CHECK(hipMemcpyAsync(diag, ctx->diag, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    if (rhs) FLIP_CHECK(hipMemcpyAsync(rhs, ctx->rhs, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Solver

struct PcgArgs {
    long long B, sy, sz;
    int nblk;
    double s2;
    const int32_t *chunk_list;
    const int32_t *n_active; // device scalar: active chunks
    const int32_t *rows;     // cell ids of the CG rows
    const int32_t *n_cg;     // device scalar: CG rows
    const uint8_t *flags;
    const float *tx, *ty, *tz;
    const double *diag, *invdiag, *rhs;
    const float *cgx, *cgy, *cgz;
    const uint8_t *reg;      // rows whose couplings all equal scal->creg
    const double *cg_diag, *cg_inv;
    double *x, *r, *z, *p0, *p1, *ap;
    double *partial; // [16][FLIP_MAX_PCG_BLOCKS]
    flip_pcg_scalars *scal;
    int coupled;
    double cdt;       // dt of the coupling term
    double minv[6];
    const double *J[6];
};

// The workgroups that share `blockIdx % 8` share an XCD (round-robin dispatch); giving them one
// contiguous run of the chunk list keeps the stencil's y/z neighbours in that XCD's L2.
__device__ __forceinline__ void d_chunk_range(const PcgArgs &A, int &lo, int &hi) {
    const int per = A.nblk >> 3;
    const int rank = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long long na = *A.n_active;
    lo = (int)((na * rank) / A.nblk);
    hi = (int)((na * (rank + 1)) / A.nblk);
}

// Sum of the nblk per-workgroup partials of one quantity, identical in every workgroup.
__device__ __forceinline__ double d_reduce_partials(const double *__restrict__ part, int nblk, double *scratch) {
    double v = 0.0;
    for (int q = threadIdx.x; q < nblk; q += FLIP_BLOCK) v += part[q];
    return d_block_sum(v, scratch);
}

// |r|^2, r.z (slots 0 and 1, summed exactly as d_reduce_partials does) and, if asked, max|r| (slot 11) behind one pair of
// barriers.  scratch: 12 doubles.
__device__ __forceinline__ void d_reduce_rr_rz_rinf(const double *__restrict__ partial, int nblk, bool with_max, double *scratch, double &rr,
                                                    double &rz, double &rinf) {
    double a = 0.0, b = 0.0, m = 0.0;
    for (int q = threadIdx.x; q < nblk; q += FLIP_BLOCK) {
        a += partial[0 * FLIP_MAX_PCG_BLOCKS + q];
        b += partial[1 * FLIP_MAX_PCG_BLOCKS + q];
        if (with_max) m = fmax(m, partial[11 * FLIP_MAX_PCG_BLOCKS + q]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, FLIP_WAVE);
        b += __shfl_down(b, off, FLIP_WAVE);
        m = fmax(m, __shfl_down(m, off, FLIP_WAVE));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { scratch[wave] = a; scratch[4 + wave] = b; scratch[8 + wave] = m; }
    __syncthreads();
    rr = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
    rz = ((scratch[4] + scratch[5]) + scratch[6]) + scratch[7];
    rinf = fmax(fmax(scratch[8], scratch[9]), fmax(scratch[10], scratch[11]));
}
__device__ __forceinline__ double d_block_max_of(const double *__restrict__ part, int nblk, double *scratch) {
    double v = 0.0;
    for (int q = threadIdx.x; q < nblk; q += FLIP_BLOCK) v = fmax(v, part[q]);
    return d_block_max(v, scratch);
}

// Row c of A applied to the vector whose entry at cell q is get(q).
template <class Get>
__device__ __forceinline__ double d_apply_row(const PcgArgs &A, long long c, uint8_t fl, double pc, Get get) {
    double s = A.diag[c] * pc;
    if (fl & 1) s -= ((double)A.tx[c] * A.s2) * get(c - 1);
    if (fl & 2) s -= ((double)A.tx[c + 1] * A.s2) * get(c + 1);
    if (fl & 4) s -= ((double)A.ty[c] * A.s2) * get(c - A.sy);
    if (fl & 8) s -= ((double)A.ty[c + A.sy] * A.s2) * get(c + A.sy);
    if (fl & 16) s -= ((double)A.tz[c] * A.s2) * get(c - A.sz);
    if (fl & 32) s -= ((double)A.tz[c + A.sz] * A.s2) * get(c + A.sz);
    return s;
}

// The CG kernels walk the compact list of CG rows (ascending cell ids): every lane has a row whatever the
// shape of the liquid.  Same XCD-aware split as the chunk list: one contiguous run of rows per XCD.
__device__ __forceinline__ void d_row_range(const PcgArgs &A, long long &lo, long long &hi) {
    const int per = A.nblk >> 3;
    const int rank = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long long n = *A.n_cg;
    lo = (n * rank) / A.nblk;
    hi = (n * (rank + 1)) / A.nblk;
}

// x, z, p0, p1 are zeroed by the host before this kernel (cells off the CG rows must read as zero).
// WARM: x holds a guess (zero off the CG rows, k_warm_mask), so r = b - A x with the symmetric CG couplings, and |b|^2
// goes to partial slot 9 because |r_0|^2 no longer is the norm the stopping rule refers to.
template <bool WARM>
__global__ void __launch_bounds__(FLIP_BLOCK) k_pcg_init(PcgArgs A) {
    __shared__ double scratch[8];
    long long lo, hi;
    d_row_range(A, lo, hi);
    double rr = 0.0, rz = 0.0, bb = 0.0, binf = 0.0, rinf = 0.0;
    const long long sy = A.sy, sz = A.sz;
    const double *__restrict__ x = A.x;
    for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
        const long long c = A.rows[q];
        double r = A.rhs[c];
        binf = fmax(binf, fabs(r));
        if (WARM) {
            bb += r * r;
            double acc = (double)A.cgx[c] * x[c - 1];
            acc += (double)A.cgx[c + 1] * x[c + 1];
            acc += (double)A.cgy[c] * x[c - sy];
            acc += (double)A.cgy[c + sy] * x[c + sy];
            acc += (double)A.cgz[c] * x[c - sz];
            acc += (double)A.cgz[c + sz] * x[c + sz];
            r -= A.cg_diag[c] * x[c] - A.s2 * acc;
        }
        const double z = A.cg_inv[c] * r;
        rr += r * r;
        rz += r * z;
        rinf = fmax(rinf, fabs(r));
        A.r[c] = r;
        A.z[c] = z;
    }
    d_block_sum_max(rr, binf, scratch);
    d_block_sum_max(rz, rinf, scratch);
    if (WARM) bb = d_block_sum(bb, scratch);
    if (threadIdx.x == 0) {
        A.partial[0 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rr;
        A.partial[1 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rz;
        if (WARM) A.partial[9 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = bb;
        A.partial[10 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = binf;
        A.partial[11 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rinf;
    }
}
// the guess of a warm start must be zero wherever the new system has no CG row (the stencil reads x at neighbours)
__global__ void k_warm_mask(long long B, const double *__restrict__ cg_inv, double *__restrict__ x) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const double v = x[c];
    // also drops anything non-finite a diverged earlier solve may have left
    if (cg_inv[c] == 0.0 || !(fabs(v) <= DBL_MAX)) x[c] = 0.0;
}

// SYM = true: the CG rows couple symmetrically (dependent rows eliminated), so the coefficient toward the
// high neighbour is the neighbour's low coefficient and every load is unconditional (the arrays are padded
// and zero off the CG rows): no flags, no branches, all 27 loads of a cell in flight together.
// SYM = false: CG over all rows of the non-symmetric matrix as the reference runs it (flag-driven).
// What k_pcg_direction reads of a row before it knows the iteration's scalars: z at the 7 stencil points, the
// couplings (unless the row is a regular one), the diagonal.  (Which of the two direction buffers holds the old
// direction depends on the iteration number, so p is read later.)
struct DirRow {
    long long c;
    double z[7];
    float a[6];
    double d;
    bool reg;
};
__device__ __forceinline__ DirRow d_dir_load(const PcgArgs &A, long long c) {
    DirRow r;
    const double *__restrict__ z = A.z;
    const long long sy = A.sy, sz = A.sz;
    r.c = c;
    r.z[0] = z[c]; r.z[1] = z[c - 1]; r.z[2] = z[c + 1]; r.z[3] = z[c - sy]; r.z[4] = z[c + sy]; r.z[5] = z[c - sz]; r.z[6] = z[c + sz];
    r.reg = A.reg[c] != 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) r.a[q] = 0.f;
    if (!r.reg) {
        r.a[0] = A.cgx[c]; r.a[1] = A.cgx[c + 1]; r.a[2] = A.cgy[c]; r.a[3] = A.cgy[c + sy]; r.a[4] = A.cgz[c]; r.a[5] = A.cgz[c + sz];
    }
    r.d = A.cg_diag[c];
    return r;
}

template <bool SYM>
__global__ void __launch_bounds__(FLIP_BLOCK) k_pcg_direction(PcgArgs A) {
    __shared__ double scratch[12];
    flip_pcg_scalars *S = A.scal;
    // The loads of a thread's first row do not depend on the scalars of the iteration: they are issued before the stop
    // flag and the partial sums are read, so that the two chains of dependent loads (flag -> partials -> sum, and
    // row count -> row list -> vectors) run side by side instead of one after the other.
    long long lo, hi;
    d_row_range(A, lo, hi);
    DirRow first = DirRow();
    const bool have_first = SYM && lo + threadIdx.x < hi;
    if (have_first) first = d_dir_load(A, A.rows[lo + threadIdx.x]);
    if (S->done) return;
    const int it = S->it_a;
    double rr, rz, rinf;
    d_reduce_rr_rz_rinf(A.partial, A.nblk, S->track_inf != 0, scratch, rr, rz, rinf);
    if (S->track_inf) {
        const double binf = it == 0 ? d_block_max_of(A.partial + 10 * FLIP_MAX_PCG_BLOCKS, A.nblk, scratch) : S->binf;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (it == 0) S->binf = binf;
            S->rinf = rinf;
            if (S->it_inf < 0 && rinf <= 1e-12 * binf) S->it_inf = it;
        }
    }
    double beta = 0.0;
    if (it == 0) {
        // Eigen: rhsNorm2 == 0 -> x = 0; threshold = max(tol^2 |b|^2, DBL_MIN)
        const double bb = S->warm ? d_reduce_partials(A.partial + 9 * FLIP_MAX_PCG_BLOCKS, A.nblk, scratch) : rr;
        double thr = S->tol2 * bb;
        if (thr < DBL_MIN) thr = DBL_MIN;
        const bool stop = bb == 0.0 || rr < thr;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            S->rhs_norm2 = bb;
            S->threshold = thr;
            S->rr = rr;
            if (stop) S->done = 1;
        }
        if (stop) {
            if (S->warm && bb == 0.0) {   // the answer to a zero right-hand side is zero, not the guess
                long long lo, hi;
                d_row_range(A, lo, hi);
                for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) A.x[A.rows[q]] = 0.0;
            }
            return;
        }
    } else {
        const bool conv = rr < S->threshold;
        const bool stop = conv || it >= S->max_iter;
        if (stop) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                S->rr = rr;
                S->done = conv ? 1 : 2;
            }
            return;
        }
        beta = rz / S->rz[(it - 1) & 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S->rz[it & 1] = rz;
        S->rr = rr;
        S->it_b = it;
        S->beta = beta;
    }
    const double *__restrict__ pold = (it & 1) ? A.p0 : A.p1;
    double *__restrict__ pnew = (it & 1) ? A.p1 : A.p0;
    const double *__restrict__ z = A.z;
    double pap = 0.0;
    double jp[6] = { 0, 0, 0, 0, 0, 0 };
    if (SYM) {
        const float creg = S->creg;
        // one row ahead: the next row's loads are in flight while this one is combined
        DirRow cur = first;
        for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
            DirRow nxt = cur;
            if (q + FLIP_BLOCK < hi) nxt = d_dir_load(A, A.rows[q + FLIP_BLOCK]);
            const long long c = cur.c;
            // which of the two direction buffers holds the old direction depends on the iteration number: p is read here
            const double p0 = pold[c], pxm = pold[c - 1], pxp = pold[c + 1], pym = pold[c - A.sy], pyp = pold[c + A.sy],
                         pzm = pold[c - A.sz], pzp = pold[c + A.sz];
            float axm = creg, axp = creg, aym = creg, ayp = creg, azm = creg, azp = creg;
            if (!cur.reg) { axm = cur.a[0]; axp = cur.a[1]; aym = cur.a[2]; ayp = cur.a[3]; azm = cur.a[4]; azp = cur.a[5]; }
            const double pc = cur.z[0] + beta * p0;
            double acc = (double)axm * (cur.z[1] + beta * pxm);
            acc += (double)axp * (cur.z[2] + beta * pxp);
            acc += (double)aym * (cur.z[3] + beta * pym);
            acc += (double)ayp * (cur.z[4] + beta * pyp);
            acc += (double)azm * (cur.z[5] + beta * pzm);
            acc += (double)azp * (cur.z[6] + beta * pzp);
            const double s = cur.d * pc - A.s2 * acc;
            pnew[c] = pc;
            A.ap[c] = s;
            pap += pc * s;
            if (A.coupled)
#pragma unroll
                for (int a = 0; a < 6; ++a) jp[a] += A.J[a][c] * pc;
            cur = nxt;
        }
    } else {
        for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
            const long long c = A.rows[q];
            const uint8_t fl = A.flags[c];
            const double pc = z[c] + beta * pold[c];
            const double s = d_apply_row(A, c, fl, pc, [&](long long n) { return z[n] + beta * pold[n]; });
            pnew[c] = pc;
            A.ap[c] = s;
            pap += pc * s;
            if (A.coupled)
#pragma unroll
                for (int a = 0; a < 6; ++a) jp[a] += A.J[a][c] * pc;
        }
    }
    pap = d_block_sum(pap, scratch);
    if (threadIdx.x == 0) A.partial[2 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = pap;
    if (A.coupled) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double v = d_block_sum(jp[a], scratch);
            if (threadIdx.x == 0) A.partial[(3 + a) * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = v;
        }
    }
}

__global__ void __launch_bounds__(FLIP_BLOCK) k_pcg_update(PcgArgs A) {
    __shared__ double scratch[8];
    flip_pcg_scalars *S = A.scal;
    if (S->done) return;
    const int it = S->it_b;
    double pap = d_reduce_partials(A.partial + 2 * FLIP_MAX_PCG_BLOCKS, A.nblk, scratch);
    double gj[6] = { 0, 0, 0, 0, 0, 0 };
    if (A.coupled) {
        // p.(A p) with A = A0 + dt J^T Minv J; gj = dt Minv (J p)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double jpa = d_reduce_partials(A.partial + (3 + a) * FLIP_MAX_PCG_BLOCKS, A.nblk, scratch);
            gj[a] = A.cdt * A.minv[a] * jpa;
            pap += jpa * gj[a];
        }
    }
    const double alpha = S->rz[it & 1] / pap;
    const double *__restrict__ p = (it & 1) ? A.p1 : A.p0;
    long long lo, hi;
    d_row_range(A, lo, hi);
    double rr = 0.0, rz = 0.0, rinf = 0.0;
    const double *__restrict__ inv = A.cg_inv;
#pragma unroll 2
    for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
        const long long c = A.rows[q];
        double apv = A.ap[c];
        const double pc = p[c], xc = A.x[c], rc = A.r[c], ic = inv[c];
        if (A.coupled) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) s += A.J[a][c] * gj[a];
            apv += s;
        }
        A.x[c] = xc + alpha * pc;
        const double r = rc - alpha * apv;
        const double zz = ic * r;
        A.r[c] = r;
        A.z[c] = zz;
        rr += r * r;
        rz += r * zz;
        rinf = fmax(rinf, fabs(r));
    }
    d_block_sum_max(rr, rinf, scratch);
    rz = d_block_sum(rz, scratch);
    if (threadIdx.x == 0) {
        A.partial[0 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rr;
        A.partial[1 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rz;
        A.partial[11 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rinf;
        if (blockIdx.x == 0) S->it_a = it + 1;
    }
}

// k_pcg_update and the pre-smoothing of the V-cycle's level 0 (k_mg_pre in flip_mg.hip) in one pass: the smoother
// needs the new residual at the 7 stencil points, which is r - alpha Ap there, so the residual is read from one
// buffer and written to the other (a neighbour's old value must survive this kernel).  Same operations in the same
// order as the two kernels, hence the same bits; one launch and one sweep over r less per iteration.
struct FusedArgs {
    const float *inv_f, *diag_f;  // float copies of the CG diagonal and its inverse (the V-cycle's level-0 view)
    double c1, a2, c2;            // smoother coefficients
    double *r2;
};
// What k_pcg_update_pre reads of a row before it knows the iteration's scalars (see k_pcg_direction): Ap and the float
// inverse diagonal at the 7 stencil points, the couplings unless the row is a regular one, the float diagonal, x.
// The residual and the direction alternate between two buffers by iteration parity and are read afterwards.
struct PreRow {
    long long c;
    double ap[7];
    float inv[7];
    float a[6];
    float diag;
    double x;
    bool reg;
};
__device__ __forceinline__ PreRow d_pre_load(const PcgArgs &A, const FusedArgs &F, long long c) {
    PreRow r;
    const double *__restrict__ ap = A.ap;
    const float *__restrict__ inv = F.inv_f;
    const long long sy = A.sy, sz = A.sz;
    r.c = c;
    r.ap[0] = ap[c]; r.ap[1] = ap[c - 1]; r.ap[2] = ap[c + 1]; r.ap[3] = ap[c - sy]; r.ap[4] = ap[c + sy]; r.ap[5] = ap[c - sz]; r.ap[6] = ap[c + sz];
    r.inv[0] = inv[c]; r.inv[1] = inv[c - 1]; r.inv[2] = inv[c + 1]; r.inv[3] = inv[c - sy]; r.inv[4] = inv[c + sy]; r.inv[5] = inv[c - sz]; r.inv[6] = inv[c + sz];
    r.reg = A.reg[c] != 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) r.a[q] = 0.f;
    if (!r.reg) {
        r.a[0] = A.cgx[c]; r.a[1] = A.cgx[c + 1]; r.a[2] = A.cgy[c]; r.a[3] = A.cgy[c + sy]; r.a[4] = A.cgz[c]; r.a[5] = A.cgz[c + sz];
    }
    r.diag = F.diag_f[c];
    r.x = A.x[c];
    return r;
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_pcg_update_pre(PcgArgs A, FusedArgs F) {
    __shared__ double scratch[8];
    flip_pcg_scalars *S = A.scal;
    long long lo, hi;
    d_row_range(A, lo, hi);
    PreRow cur = PreRow();
    if (lo + threadIdx.x < hi) cur = d_pre_load(A, F, A.rows[lo + threadIdx.x]);
    if (S->done) return;
    const int it = S->it_b;
    const double pap = d_reduce_partials(A.partial + 2 * FLIP_MAX_PCG_BLOCKS, A.nblk, scratch);
    const double alpha = S->rz[it & 1] / pap;
    const double *__restrict__ p = (it & 1) ? A.p1 : A.p0;
    const double *__restrict__ ro = (it & 1) ? F.r2 : A.r;
    double *__restrict__ rn = (it & 1) ? A.r : F.r2;
    const long long sy = A.sy, sz = A.sz;
    double rr = 0.0, rinf = 0.0;
    const float creg = S->creg;
    for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
        PreRow nxt = cur;
        if (q + FLIP_BLOCK < hi) nxt = d_pre_load(A, F, A.rows[q + FLIP_BLOCK]);
        const long long c = cur.c;
        const double rc = ro[c] - alpha * cur.ap[0];
        const double rxm = ro[c - 1] - alpha * cur.ap[1], rxp = ro[c + 1] - alpha * cur.ap[2];
        const double rym = ro[c - sy] - alpha * cur.ap[3], ryp = ro[c + sy] - alpha * cur.ap[4];
        const double rzm = ro[c - sz] - alpha * cur.ap[5], rzp = ro[c + sz] - alpha * cur.ap[6];
        const double pc = p[c];
        const double ic = (double)cur.inv[0];
        const double t0 = F.c1 * ic * rc;
        float axm = creg, axp = creg, aym = creg, ayp = creg, azm = creg, azp = creg;
        if (!cur.reg) { axm = cur.a[0]; axp = cur.a[1]; aym = cur.a[2]; ayp = cur.a[3]; azm = cur.a[4]; azp = cur.a[5]; }
        double acc = (double)axm * (F.c1 * (double)cur.inv[1] * rxm);
        acc += (double)axp * (F.c1 * (double)cur.inv[2] * rxp);
        acc += (double)aym * (F.c1 * (double)cur.inv[3] * rym);
        acc += (double)ayp * (F.c1 * (double)cur.inv[4] * ryp);
        acc += (double)azm * (F.c1 * (double)cur.inv[5] * rzm);
        acc += (double)azp * (F.c1 * (double)cur.inv[6] * rzp);
        const double At = (double)cur.diag * t0 - A.s2 * acc;
        A.z[c] = t0 + F.a2 * t0 + F.c2 * ic * (rc - At);
        A.x[c] = cur.x + alpha * pc;
        rn[c] = rc;
        rr += rc * rc;
        rinf = fmax(rinf, fabs(rc));
        cur = nxt;
    }
    d_block_sum_max(rr, rinf, scratch);
    if (threadIdx.x == 0) {
        A.partial[0 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rr;
        A.partial[11 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rinf;
        if (blockIdx.x == 0) S->it_a = it + 1;
    }
}

// Back-substitution of the dependent rows: diag x_c = b_c - sum_f A(c,n_f) x_n, x_c = 0 for an all-zero row.
// CHECKED: only once the solve has stopped (the launch rides behind a batch of iterations, before the host has seen
// the stop flag).
template <bool CHECKED>
__global__ void __launch_bounds__(FLIP_BLOCK) k_pcg_dependent(PcgArgs A) {
    if (CHECKED && A.scal->done == 0) return;
    int lo, hi;
    d_chunk_range(A, lo, hi);
    for (int q = lo; q < hi; ++q) {
        const long long c = (long long)A.chunk_list[q] * FLIP_BLOCK + threadIdx.x;
        if (c >= A.B) continue;
        const uint8_t fl = A.flags[c];
        if ((fl & (FLIP_FLAG_ROW | FLIP_FLAG_DEP)) != (FLIP_FLAG_ROW | FLIP_FLAG_DEP)) continue;
        const double d = A.diag[c];
        const double s = d_apply_row(A, c, fl, 0.0, [&](long long n) { return A.x[n]; });
        A.x[c] = d != 0 ? (A.rhs[c] - s) / d : 0.0;
    }
}

static void fill_pcg_args(flip_ctx *ctx, PcgArgs &A) {
    memset(&A, 0, sizeof A);
    A.B = ctx->B;
    A.sy = ctx->g.nx;
    A.sz = (long long)ctx->g.nx * ctx->g.ny;
    A.nblk = ctx->pcg_blocks;
    A.s2 = ctx->sys_s2;
    A.chunk_list = ctx->chunk_list;
    A.n_active = ctx->d_counts + 1;
    A.rows = ctx->cg_rows;
    A.n_cg = ctx->d_counts + 6;
    A.flags = ctx->flags;
    A.tx = ctx->term[0]; A.ty = ctx->term[1]; A.tz = ctx->term[2];
    A.diag = ctx->diag; A.invdiag = ctx->invdiag; A.rhs = ctx->rhs;
    A.cgx = ctx->cg_c[0]; A.cgy = ctx->cg_c[1]; A.cgz = ctx->cg_c[2];
    A.reg = ctx->reg;
    A.cg_diag = ctx->cg_diag; A.cg_inv = ctx->cg_inv;
    A.x = ctx->x; A.r = ctx->r; A.z = ctx->z; A.p0 = ctx->pv[0]; A.p1 = ctx->pv[1]; A.ap = ctx->ap;
    A.partial = ctx->partial;
    A.scal = ctx->scal;
    A.coupled = ctx->sys_coupled ? 1 : 0;
    A.cdt = ctx->sys_coupled ? (double)ctx->sys_dt : 0.0; // read by the coupled kernels only
    for (int a = 0; a < 6; ++a) { A.minv[a] = ctx->obj_minv[a]; A.J[a] = ctx->J[a]; }
}

__global__ void k_set_scalars(flip_pcg_scalars *dst, flip_pcg_scalars v) { *dst = v; }

// flip_pcg_solve in three parts, so that flip_solve_pair can interleave two solves:
//   pcg_prepare  validation, scalars, zeroed vectors, coarse operators, r = b, z = M^-1 r
//   pcg_batch    `batch` iterations (direction, update, V-cycle) enqueued on ctx->stream
//   pcg_finish   dependent rows, counters, timings
int flip_pair_streams(flip_ctx *ctx) {
    for (int q = 0; q < 2; ++q)
        if (!ctx->pair_stream[q]) FLIP_CHECK(hipStreamCreateWithFlags(&ctx->pair_stream[q], hipStreamNonBlocking));
    for (int q = 0; q < 3; ++q)
        if (!ctx->pair_event[q]) FLIP_CHECK(hipEventCreateWithFlags(&ctx->pair_event[q], hipEventDisableTiming));
    return 0;
}

struct PcgRun {
    bool use_mg = false, sym = false, fused = false;
    int n_prof = 0, launched = 0, max_iter = 0;
};

static int pcg_prepare(flip_ctx *ctx, double tol, int32_t max_iter, int precond, PcgRun &run, const char *who) {
    FLIP_REQUIRE(ctx->sys_kind >= 0, "the solver needs flip_build_system first");
    FLIP_REQUIRE(precond == FLIP_PRECOND_JACOBI || precond == FLIP_PRECOND_MULTIGRID, "unknown preconditioner");
    // the V-cycle is a symmetric operator for the symmetric CG system; a box too small to coarsen keeps Jacobi
    bool use_mg = precond == FLIP_PRECOND_MULTIGRID;
    FLIP_REQUIRE(!use_mg || ctx->sys_eliminated,
                 "the multigrid preconditioner needs eliminate_dependent_rows = 1 (symmetric system)");
    FLIP_REQUIRE(tol > 0.0, "tol must be positive");
    FLIP_REQUIRE(!ctx->slab_on, "this context owns a slab of a larger system; use the flip_pcg_dist_* entry points");
    FLIP_REQUIRE(ctx->pcg_blocks % 8 == 0 && ctx->pcg_blocks <= FLIP_MAX_PCG_BLOCKS, "bad PCG grid size");
    (void)who;
    hipStream_t st = ctx->stream;
    if (max_iter <= 0) {   // Eigen's default: twice the number of rows
        if (ctx->n_rows < 0) {   // flip_index_cells left the count on the device
            int32_t n = 0;
            FLIP_CHECK(hipMemcpyAsync(&n, ctx->d_counts, 4, hipMemcpyDeviceToHost, st));
            FLIP_CHECK(hipStreamSynchronize(st));
            ctx->n_rows = n;
        }
        max_iter = 2 * (ctx->n_rows > 0 ? ctx->n_rows : 1);
    }

    // cells off the CG rows must read as zero in x and in the vectors the stencil reads at neighbours
    // Warm start: the last converged solution of a system of the same kind on these arrays is the guess (Eigen's
    // solveWithGuess; the stopping rule is the same |r| < tol |b|, so the result meets the same bound).  Only for the
    // plain symmetric system: the rigid-body term and the as-run non-symmetric form start from zero as the reference does.
    const bool warm = ctx->warm_start && ctx->x_kind == ctx->sys_kind && ctx->sys_eliminated && !ctx->sys_coupled;
    ctx->x_kind = -1;
    if (warm) {
        k_warm_mask<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(ctx->B, ctx->cg_inv, ctx->x);
        FLIP_LAUNCH_CHECK();
    }
    // The residual buffers are read at the six neighbours of a row; a cell that was a row of an earlier solve must not
    // hand a stale (possibly non-finite) value to a zero coupling.  Ap is the V-cycle's scratch on level 0.
    if (int rc = flip_zero_solver_vectors(ctx, warm, st)) return rc;
    if (use_mg) {
        if (int rc = flip_mg_setup(ctx)) return rc;
        if (ctx->mg.empty()) use_mg = false;
    }
    static const bool fuse_on = [] { const char *e = getenv("FLIP_FUSED_UPDATE"); return !(e && atoi(e) == 0); }();
    const bool fused = fuse_on && use_mg && ctx->sys_eliminated && !ctx->sys_coupled;
    flip_pcg_scalars h;
    memset(&h, 0, sizeof h);
    h.tol2 = tol * tol;
    h.max_iter = max_iter;
    h.fused = fused ? 1 : 0;
    h.creg = ctx->reg_value;
    h.warm = warm ? 1 : 0;
    h.track_inf = 1;
    h.it_inf = -1;
    k_set_scalars<<<1, 1, 0, st>>>(ctx->scal, h); // by value: no host buffer has to outlive the call
    FLIP_LAUNCH_CHECK();
    PcgArgs A;
    fill_pcg_args(ctx, A);
    while (ctx->events.size() < 2 + 2 * 64) {
        hipEvent_t e;
        FLIP_CHECK(hipEventCreate(&e));
        ctx->events.push_back(e);
    }
    FLIP_CHECK(hipEventRecord(ctx->events[0], st));
    if (warm) k_pcg_init<true><<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(A);
    else k_pcg_init<false><<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(A);
    FLIP_LAUNCH_CHECK();
    if (use_mg)
        if (int rc = flip_mg_vcycle(ctx)) return rc;
    run.use_mg = use_mg;
    run.fused = fused;
    run.sym = ctx->sys_eliminated; // dependent rows eliminated -> symmetric CG system
    run.n_prof = 0;
    run.launched = 0;
    run.max_iter = max_iter;
    return 0;
}

static int pcg_batch_direct(flip_ctx *ctx, PcgRun &run, int batch, bool allow_events);

// Iterations per launch batch of a multigrid solve (the host reads the stop flag after each batch), and how the batches
// are sized: long ones while the solve is sure to need them, short ones around the iteration count the last solve of
// the same kind took (consecutive substeps need nearly the same number), so that few launches run after convergence
// and the stop flag is read once or twice per solve instead of every 8 iterations.
#define FLIP_MG_BATCH 8
#define FLIP_MG_BATCH_SHORT 2
static int mg_next_batch(int launched, int expected) {
    if (expected <= 0) return FLIP_MG_BATCH;
    // the direction kernel of iteration `expected` (counted from 0) is the one that sees the converged residual
    const int target = expected + 1;
    if (launched + FLIP_MG_BATCH <= target) return FLIP_MG_BATCH;
    if (launched < target + 2 * FLIP_MG_BATCH_SHORT) return FLIP_MG_BATCH_SHORT;
    return FLIP_MG_BATCH;
}
// whether the stop flag is worth reading after a batch: not while the solve cannot have converged yet
static bool mg_check_after(int launched, int expected) { return expected <= 0 || launched >= expected + 1; }
static bool pcg_graphs_enabled() {
    static const bool on = [] { const char *e = getenv("FLIP_GRAPHS"); return !(e && atoi(e) == 0); }();
    return on;
}

// `batch` iterations on ctx->stream.  The launches of a batch take the same arguments in every solve of a kind (the
// iteration's state lives in device memory), so they can be captured once into a graph and replayed.  That pays when
// two solves share the device (flip_solve_pair): 2 x 19 launches per iteration pair cost more host time than the
// device needs to run them.  A single solve is faster with plain launches (measured), and only those can be
// bracketed by the profiling events.
static int pcg_batch(flip_ctx *ctx, PcgRun &run, int batch, bool as_graph) {
    hipStream_t st = ctx->stream;
    if (!as_graph || !pcg_graphs_enabled() || ctx->sys_coupled || st == nullptr) return pcg_batch_direct(ctx, run, batch, ctx->profiling);
    PcgArgs A;
    fill_pcg_args(ctx, A);
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t q = 0; q < n; ++q) { key ^= b[q]; key *= 1099511628211ull; }
    };
    mix(&A, sizeof A);
    const int extra[5] = { batch, run.use_mg ? 1 : 0, run.sym ? 1 : 0, (int)ctx->mg.size(), run.fused ? 1 : 0 };
    mix(extra, sizeof extra);
    mix(&ctx->mg_diag_f, sizeof ctx->mg_diag_f);
    const int slot = batch == FLIP_MG_BATCH ? 0 : 1;
    if (!ctx->batch_exec[slot] || ctx->batch_key[slot] != key) {
        if (ctx->batch_exec[slot]) { FLIP_CHECK(hipGraphExecDestroy(ctx->batch_exec[slot])); ctx->batch_exec[slot] = nullptr; }
        hipGraph_t graph = nullptr;
        FLIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        PcgRun scratch = run;
        const int rc = pcg_batch_direct(ctx, scratch, batch, false);
        const hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        FLIP_CHECK(e);
        const hipError_t ei = hipGraphInstantiate(&ctx->batch_exec[slot], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        FLIP_CHECK(ei);
        ctx->batch_key[slot] = key;
    }
    FLIP_CHECK(hipGraphLaunch(ctx->batch_exec[slot], st));
    run.launched += batch;
    return 0;
}

static int pcg_batch_direct(flip_ctx *ctx, PcgRun &run, int batch, bool allow_events) {
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = ctx->pcg_blocks;
    for (int b = 0; b < batch; ++b) {
        const bool prof = allow_events && run.n_prof < 64;
        // the first bracketed iteration of a multigrid solve also has every level-0 kernel timed on its own
        const bool marks = prof && run.n_prof == 0 && run.use_mg;
        if (marks) {
            for (hipEvent_t &e : ctx->kprof)
                if (!e) FLIP_CHECK(hipEventCreate(&e));
            FLIP_CHECK(hipEventRecord(ctx->kprof[0], st));
        }
        if (prof) FLIP_CHECK(hipEventRecord(ctx->events[2 + 2 * run.n_prof], st));
        if (run.sym) k_pcg_direction<true><<<nblk, FLIP_BLOCK, 0, st>>>(A);
        else k_pcg_direction<false><<<nblk, FLIP_BLOCK, 0, st>>>(A);
        if (prof) { FLIP_CHECK(hipEventRecord(ctx->events[3 + 2 * run.n_prof], st)); ++run.n_prof; }
        if (marks) FLIP_CHECK(hipEventRecord(ctx->kprof[1], st));
        ctx->kprof_arm = marks;
        if (run.fused) {
            FusedArgs F;
            F.inv_f = ctx->mg_inv_f; F.diag_f = ctx->mg_diag_f; F.r2 = ctx->r2;
            flip_mg_smoother(&F.c1, &F.a2, &F.c2);
            k_pcg_update_pre<<<nblk, FLIP_BLOCK, 0, st>>>(A, F);
            if (int rc = flip_mg_vcycle(ctx, true)) return rc;
        } else {
            k_pcg_update<<<nblk, FLIP_BLOCK, 0, st>>>(A);
            if (run.use_mg)
                if (int rc = flip_mg_vcycle(ctx)) return rc;
        }
        if (marks) { ctx->kprof_arm = false; ctx->kprof_valid = true; }
    }
    FLIP_LAUNCH_CHECK();
    run.launched += batch;
    return 0;
}

// Enqueues the tail of a solve on ctx->stream: dependent rows, end event, read-back of scalars and counters into
// host memory that must stay alive until the stream has been synchronised; pcg_report then fills `info`.
struct PcgTail { flip_pcg_scalars h; int32_t counts[8]; };
// if_done: the dependent rows are substituted only if the stop flag is set by the time the kernel runs; the caller
// learns that from t.h.done after synchronising and otherwise enqueues the tail again later.
static int pcg_enqueue_tail(flip_ctx *ctx, PcgTail &t, bool if_done = false) {
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    if (ctx->sys_eliminated) {
        if (if_done) k_pcg_dependent<true><<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(A);
        else k_pcg_dependent<false><<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(A);
        FLIP_LAUNCH_CHECK();
    }
    FLIP_CHECK(hipEventRecord(ctx->events[1], st));
    memset(&t, 0, sizeof t);
    FLIP_CHECK(hipMemcpyAsync(&t.h, ctx->scal, sizeof t.h, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipMemcpyAsync(t.counts, ctx->d_counts, 32, hipMemcpyDeviceToHost, st));
    return 0;
}
static int pcg_report(flip_ctx *ctx, const PcgRun &run, const PcgTail &t, flip_pcg_info *info) {
    ctx->n_active_chunks = t.counts[1];
    if (run.use_mg && t.h.done == 1) ctx->last_iterations[ctx->sys_kind & 1] = t.h.it_a;
    ctx->x_kind = t.h.done == 1 ? ctx->sys_kind : -1;
    if (!info) return 0;
    memset(info, 0, sizeof *info);
    info->converged = t.h.done == 1 ? 1 : 0;
    info->iterations = t.h.it_a;
    info->rows = t.counts[0];
    info->active_chunks = t.counts[1];
    info->dependent_rows = t.counts[4];
    info->cg_rows = t.counts[6];
    info->rhs_norm_2 = sqrt(t.h.rhs_norm2);
    info->rel_residual_2 = t.h.rhs_norm2 > 0 ? sqrt(t.h.rr / t.h.rhs_norm2) : 0.0;
    info->iterations_inf = t.h.track_inf ? t.h.it_inf : -1;
    info->rel_residual_inf = (t.h.track_inf && t.h.binf > 0) ? t.h.rinf / t.h.binf : 0.0;
    float ms = 0.f;
    FLIP_CHECK(hipEventElapsedTime(&ms, ctx->events[0], ctx->events[1]));
    info->seconds = (double)ms * 1e-3;
    double sp = 0.0;
    int used = 0;
    for (int q = 0; q < run.n_prof && q < t.h.it_a; ++q) {
        float e = 0.f;
        FLIP_CHECK(hipEventElapsedTime(&e, ctx->events[2 + 2 * q], ctx->events[3 + 2 * q]));
        sp += (double)e * 1e-3;
        ++used;
    }
    info->spmv_seconds = sp;
    info->spmv_launches = used;
    return 0;
}

extern "C" int flip_pcg_solve(flip_ctx *ctx, double tol, int32_t max_iter, int precond, flip_pcg_info *info) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!ctx->pair_on, "flip_pcg_solve: not between flip_pair_begin and flip_pair_end");
    FLIP_CHECK(hipSetDevice(ctx->device));
    PcgRun run;
    if (int rc = pcg_prepare(ctx, tol, max_iter, precond, run, "flip_pcg_solve")) return rc;
    // the iterations run on a stream of the library's own (the caller's may be the null stream, which cannot be
    // captured into a graph), ordered after everything enqueued so far
    hipStream_t main_stream = ctx->stream;
    if (int rc = flip_pair_streams(ctx)) return rc;
    hipStream_t st = ctx->pair_stream[0];
    FLIP_CHECK(hipEventRecord(ctx->pair_event[2], main_stream));
    FLIP_CHECK(hipStreamWaitEvent(st, ctx->pair_event[2], 0));
    ctx->stream = st;
    const int expected = ctx->last_iterations[ctx->sys_kind & 1];
    int32_t done = 0;
    int rc = 0;
    // one extra direction launch decides convergence after the last update
    while (!rc && !done && run.launched <= run.max_iter) {
        rc = pcg_batch(ctx, run, run.use_mg ? mg_next_batch(run.launched, expected) : 32, false);
        if (!rc && run.use_mg && !mg_check_after(run.launched, expected) && run.launched <= run.max_iter) continue;
        if (!rc && hipMemcpyAsync(&done, &ctx->scal->done, 4, hipMemcpyDeviceToHost, st) != hipSuccess) rc = flip_fail(ctx, "hipMemcpyAsync of the stop flag", __FILE__, __LINE__);
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = flip_fail(ctx, "hipStreamSynchronize in the solver loop", __FILE__, __LINE__);
    }
    PcgTail tail;
    if (!rc) rc = pcg_enqueue_tail(ctx, tail);
    ctx->stream = main_stream;
    if (rc) return rc;
    FLIP_CHECK(hipStreamSynchronize(st));
    FLIP_CHECK(hipEventRecord(ctx->pair_event[0], st));
    FLIP_CHECK(hipStreamWaitEvent(main_stream, ctx->pair_event[0], 0));
    return pcg_report(ctx, run, tail, info);
}

// The divergence system (src/FluidSimulator.cpp:505) and the density system (:521) of one substep depend on the same
// labels, level sets and face fractions and on nothing the other solve produces, so both are assembled first and then
// iterated side by side on two streams: the latency-bound coarse levels of one V-cycle run under the bandwidth-bound
// fine level of the other.  Each solve performs exactly the operations flip_build_system + flip_pcg_solve would.
// Afterwards the divergence system is the active one; flip_select_system switches between the two for the projections.
// The arrays of the second system set (allocated on first use).
static int ensure_other_set(flip_ctx *ctx) {
    if (ctx->other.allocated) return 0;
    flip_swap_systems(ctx);
    int rc = flip_alloc_system(ctx);
    if (!rc && hipMalloc((void **)&ctx->cg_scan, (size_t)(ctx->B + 1) * 4) != hipSuccess)
        rc = flip_fail(ctx, "hipMalloc of the second system's scan scratch", __FILE__, __LINE__);
    if (rc) {
        // nothing of a half-allocated set may survive: the next call would allocate over its pointers
        flip_free_system(ctx);
        if (ctx->cg_scan) { (void)hipFree(ctx->cg_scan); ctx->cg_scan = nullptr; }
    }
    flip_swap_systems(ctx);
    if (rc) return rc;
    ctx->other.allocated = true;
    return 0;
}

extern "C" int flip_solve_pair(flip_ctx *ctx, float dt, double tol, int32_t max_iter, int precond, flip_pcg_info *info_divergence,
                               flip_pcg_info *info_density) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!ctx->slab_on, "flip_solve_pair: not available on a slab context");
    FLIP_REQUIRE(!ctx->pair_on, "flip_solve_pair: not between flip_pair_begin and flip_pair_end");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t main_stream = ctx->stream;
    if (int rc0 = flip_pair_streams(ctx)) return rc0;
    if (int rc0 = ensure_other_set(ctx)) return rc0;
    // The density system lives in the other set, the divergence system in the active one.  Assembly and solver set-up of
    // each (coarse operators, row lists: several dozen small launches) already run on the solve's own stream, side by
    // side with the other's; every scratch array they use belongs to the system set.
    PcgRun run[2];
    int rc = 0;
    FLIP_CHECK(hipEventRecord(ctx->pair_event[2], main_stream));
    FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[0], ctx->pair_event[2], 0));
    FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[1], ctx->pair_event[2], 0));
    bool active[2] = { true, true }, capped[2] = { false, false };
    PcgTail tail[2];
    // ctx->stream is the stream of whichever system is active; restored before returning
    auto with_system = [&](int q, auto &&fn) -> int {
        if (q == 1) flip_swap_systems(ctx);
        ctx->stream = ctx->pair_stream[q];
        const int r = fn();
        ctx->stream = main_stream;
        if (q == 1) flip_swap_systems(ctx);
        return r;
    };
    // An error return must not leave work queued on the pair streams unordered against the context's stream: later calls
    // on it touch the same arrays (density, labels, the system sets).  The error text of the failing call is kept.
    auto bail = [&](int code) -> int {
        const std::string text = ctx->err;
        for (int q = 0; q < 2; ++q) (void)hipStreamSynchronize(ctx->pair_stream[q]);
        ctx->err = text;
        return code;
    };
    rc = with_system(1, [&]() -> int {
        // the row count of index_cells sits with the system that was active then
        if (int r = flip_copy_i32(ctx, ctx->d_counts, ctx->other.d_counts, ctx->stream)) return r;
        if (int r = flip_build_system(ctx, FLIP_SOLVE_DENSITY, dt)) return r;
        return pcg_prepare(ctx, tol, max_iter, precond, run[1], "flip_solve_pair");
    });
    if (rc) return bail(rc);
    rc = with_system(0, [&]() -> int {
        if (int r = flip_build_system(ctx, FLIP_SOLVE_DIVERGENCE, dt)) return r;
        return pcg_prepare(ctx, tol, max_iter, precond, run[0], "flip_solve_pair");
    });
    if (rc) return bail(rc);
    if (ctx->profiling) {
        // the direction kernels bracketed by events run with the device to themselves: two iterations of the divergence
        // solve before the density solve's stream is released
        // (the first iteration untimed: its launch follows a host-side pause, and the gap would be charged to the kernel);
        // the set-up of the density system on the other stream must have drained first
        FLIP_CHECK(hipEventRecord(ctx->pair_event[2], ctx->pair_stream[1]));
        FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[0], ctx->pair_event[2], 0));
        if ((rc = with_system(0, [&] { return pcg_batch_direct(ctx, run[0], 1, false); }))) return bail(rc);
        if ((rc = with_system(0, [&] { return pcg_batch_direct(ctx, run[0], 1, true); }))) return bail(rc);
        FLIP_CHECK(hipEventRecord(ctx->pair_event[2], ctx->pair_stream[0]));
        FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[1], ctx->pair_event[2], 0));
    }
    const int expected[2] = { ctx->last_iterations[FLIP_SOLVE_DIVERGENCE], ctx->last_iterations[FLIP_SOLVE_DENSITY] };
    while (active[0] || active[1]) {
        // batches of both solves, alternating, until one of them is due for a look at its stop flag
        for (bool more = true; more;) {
            more = true;
            for (int q = 0; q < 2; ++q)
                if (active[q]) {
                    const int batch = run[q].use_mg ? mg_next_batch(run[q].launched, expected[q]) : 32;
                    if ((rc = with_system(q, [&] { return pcg_batch(ctx, run[q], batch, true); }))) return bail(rc);
                    if (!run[q].use_mg || mg_check_after(run[q].launched, expected[q]) || run[q].launched > run[q].max_iter) more = false;
                }
        }
        // The look at a solve's stop flag brings its whole tail along -- the substitution of the dependent rows (a launch
        // that returns at once while the flag is clear), the end event, the scalars and counters -- so that a solve found
        // stopped needs no second round trip between device and host.
        for (int q = 0; q < 2; ++q)
            if (active[q] && (rc = with_system(q, [&] { return pcg_enqueue_tail(ctx, tail[q], true); }))) return bail(rc);
        for (int q = 0; q < 2; ++q)
            if (active[q]) FLIP_CHECK(hipStreamSynchronize(ctx->pair_stream[q]));
        for (int q = 0; q < 2; ++q)
            if (active[q]) {
                if (tail[q].h.done) active[q] = false;
                else if (run[q].launched > run[q].max_iter) { active[q] = false; capped[q] = true; }
            }
    }
    for (int q = 0; q < 2; ++q)   // stopped by the iteration limit: the tail without the flag's consent
        if (capped[q] && (rc = with_system(q, [&] { return pcg_enqueue_tail(ctx, tail[q]); }))) return bail(rc);
    for (int q = 0; q < 2; ++q) {
        if (capped[q]) FLIP_CHECK(hipStreamSynchronize(ctx->pair_stream[q]));
        // later work on the context's own stream is ordered after both solves
        FLIP_CHECK(hipEventRecord(ctx->pair_event[q], ctx->pair_stream[q]));
        FLIP_CHECK(hipStreamWaitEvent(main_stream, ctx->pair_event[q], 0));
    }
    if ((rc = pcg_report(ctx, run[0], tail[0], info_divergence))) return rc;
    flip_swap_systems(ctx);
    rc = pcg_report(ctx, run[1], tail[1], info_density);
    flip_swap_systems(ctx);
    return rc;
}

extern "C" int flip_solver_kernel_times(flip_ctx *ctx, double us[6]) {
    FLIP_REQUIRE(ctx && us, "flip_solver_kernel_times: null argument");
    FLIP_REQUIRE(ctx->kprof_valid, "flip_solver_kernel_times: no profiled multigrid solve yet (flip_set_profiling, then a solve)");
    FLIP_CHECK(hipSetDevice(ctx->device));
    FLIP_CHECK(hipEventSynchronize(ctx->kprof[6]));
    for (int q = 0; q < 6; ++q) {
        float ms = 0.f;
        FLIP_CHECK(hipEventElapsedTime(&ms, ctx->kprof[q], ctx->kprof[q + 1]));
        us[q] = (double)ms * 1e3;
    }
    return 0;
}

extern "C" int flip_select_system(flip_ctx *ctx, int kind) {
    FLIP_REQUIRE(ctx, "null context");
    if (ctx->sys_kind == kind) return 0;
    FLIP_REQUIRE(ctx->other.allocated && ctx->other.sys_kind == kind, "flip_select_system: no assembled system of this kind");
    flip_swap_systems(ctx);
    return 0;
}

// Two systems in progress side by side under the caller's control (the slab driver interleaves the two solves of a
// substep: the collectives of one run under the kernels of the other).  Between flip_pair_begin and flip_pair_end the
// context holds two system sets, each with a stream of its own; flip_pair_select makes one of them the active set, to
// which every call then refers (flip_build_system, flip_pcg_dist_*, flip_get_stream, the projections).  Work enqueued
// before begin is ordered before both streams; after end the context's own stream is ordered after both.
extern "C" int flip_pair_begin(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(!ctx->pair_on, "flip_pair_begin: already begun");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = flip_pair_streams(ctx)) return rc;
    if (int rc = ensure_other_set(ctx)) return rc;
    if (ctx->dist_sums) {   // a slab context: the second set needs the slab solve's arrays too
        flip_swap_systems(ctx);
        const int rc = slab_alloc_state(ctx);
        flip_swap_systems(ctx);
        if (rc) return rc;
    }
    hipStream_t main_stream = ctx->stream;
    // the row count of flip_index_cells sits with the active set
    if (int rc = flip_copy_i32(ctx, ctx->other.d_counts, ctx->d_counts, main_stream)) return rc;
    FLIP_CHECK(hipEventRecord(ctx->pair_event[2], main_stream));
    FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[0], ctx->pair_event[2], 0));
    FLIP_CHECK(hipStreamWaitEvent(ctx->pair_stream[1], ctx->pair_event[2], 0));
    ctx->pair_main_stream = main_stream;
    ctx->stream = ctx->pair_stream[0];
    ctx->pair_slot = 0;
    ctx->pair_on = true;
    return 0;
}
extern "C" int flip_pair_select(flip_ctx *ctx, int slot) {
    FLIP_REQUIRE(ctx && ctx->pair_on, "flip_pair_select: call flip_pair_begin first");
    FLIP_REQUIRE(slot == 0 || slot == 1, "flip_pair_select: the slot is 0 or 1");
    if (slot != ctx->pair_slot) {
        flip_swap_systems(ctx);
        ctx->pair_slot = slot;
    }
    ctx->stream = ctx->pair_stream[slot];
    return 0;
}
extern "C" int flip_pair_end(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx && ctx->pair_on, "flip_pair_end: call flip_pair_begin first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->pair_slot != 0) { flip_swap_systems(ctx); ctx->pair_slot = 0; }
    ctx->stream = ctx->pair_main_stream;
    ctx->pair_on = false;
    for (int q = 0; q < 2; ++q) {
        FLIP_CHECK(hipEventRecord(ctx->pair_event[q], ctx->pair_stream[q]));
        FLIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->pair_event[q], 0));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Split-phase solver for a z-slab decomposition.  The same kernels as flip_pcg_solve; the caller (one process
// per GPU) supplies the two things that cross ranks:
//   - an all-reduce (sum) of the 8-double buffer flip_pcg_dist_buffer(0) after begin(), after every direction()
//     and after every update();
//   - an exchange of the one z plane on either side of a slab face (buffer 1: send planes slab_k0 and
//     slab_k1-1, receive into planes slab_k0-1 and slab_k1) after begin() and after every update().
// Every rank then sees identical scalars, takes identical branches and stops at the same iteration.  The
// preconditioner (Jacobi, or the V-cycle over the rank's own rows) never reads across a slab face, so it is
// block diagonal over the ranks: symmetric positive definite, no communication.  With a rigid body attached the six
// sums J p of the rank-6 term travel in the same buffer as p.Ap.
//
// The global sums travel through the kernels' own reduction scheme: k_dist_gather folds this rank's
// per-workgroup partials of the named slots into the buffer (other entries zero, so one all-reduce serves any
// phase), k_dist_scatter writes the global value back as partial[slot][0] with zeros behind it, and the next
// kernel's fixed-order re-reduction returns exactly that value.

__global__ void __launch_bounds__(FLIP_BLOCK) k_dist_gather(const double *__restrict__ partial, int nblk, int first, int count,
                                                           double *__restrict__ sums) {
    __shared__ double scratch[4];
    for (int s = 0; s < 16; ++s) {
        double v = 0.0;
        if (s >= first && s < first + count) v = d_reduce_partials(partial + (size_t)s * FLIP_MAX_PCG_BLOCKS, nblk, scratch);
        if (threadIdx.x == 0) sums[s] = v;
    }
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_dist_scatter(double *__restrict__ partial, int nblk, int first, int count,
                                                            const double *__restrict__ sums) {
    for (int s = first; s < first + count; ++s)
        for (int q = threadIdx.x; q < nblk; q += FLIP_BLOCK) partial[(size_t)s * FLIP_MAX_PCG_BLOCKS + q] = q == 0 ? sums[s] : 0.0;
}
// After the direction kernel: the ghost planes' search direction follows the same recurrence from the
// neighbour's z received before it; the received z is then cleared so that the preconditioner sees zeros there.
__global__ void k_dist_ghost_direction(flip_pcg_scalars *S, long long plane, long long lo_ghost, long long hi_ghost,
                                       double *__restrict__ z, double *__restrict__ p0, double *__restrict__ p1) {
    if (S->done) return;
    const int it = S->it_b;
    const double beta = S->beta;
    const double *__restrict__ pold = (it & 1) ? p0 : p1;
    double *__restrict__ pnew = (it & 1) ? p1 : p0;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= plane) return;
    if (lo_ghost >= 0) { const long long c = lo_ghost + q; pnew[c] = z[c] + beta * pold[c]; z[c] = 0.0; }
    if (hi_ghost >= 0) { const long long c = hi_ghost + q; pnew[c] = z[c] + beta * pold[c]; z[c] = 0.0; }
}

// Weights of the overlapping variant: across the 2m planes around a slab face this rank's weight rises linearly
// from 0 (far side) to 1 (own side) and the neighbour's falls likewise, so the two add up to one.  The
// preconditioner is sum_i R_i^T W_i M_i W_i R_i with W_i the square roots: symmetric, and a cell of the
// overlap is no longer corrected twice.
struct SlabWeight { int k0, k1, m, has_lo, has_hi; long long plane; };
static SlabWeight slab_weight(const flip_ctx *ctx) {
    SlabWeight W;
    W.k0 = ctx->slab_k0; W.k1 = ctx->slab_k1; W.m = ctx->slab_overlap;
    W.has_lo = ctx->slab_k0 > 0; W.has_hi = ctx->slab_k1 < ctx->g.nz;
    W.plane = (long long)ctx->g.nx * ctx->g.ny;
    return W;
}
__device__ __forceinline__ double d_slab_weight(const SlabWeight &W, int k) {
    double w = 1.0;
    if (W.has_lo && k < W.k0 + W.m) { const double t = ((double)(k - (W.k0 - W.m)) + 0.5) / (2.0 * W.m); w = t > 0.0 ? sqrt(t) : 0.0; }
    if (W.has_hi && k >= W.k1 - W.m) { const double t = ((double)(W.k1 + W.m - 1 - k) + 0.5) / (2.0 * W.m); w *= t > 0.0 ? sqrt(t) : 0.0; }
    return w;
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_dist_weight_r(const flip_pcg_scalars *S, SlabWeight W, const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ n_rows, const double *__restrict__ r,
                                                             double *__restrict__ rw) {
    if (S->done) return;
    const long long n = *n_rows;
    for (long long q = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x; q < n; q += (long long)gridDim.x * FLIP_BLOCK) {
        const long long c = rows[q];
        rw[c] = d_slab_weight(W, (int)(c / W.plane)) * r[c];
    }
}
__global__ void k_dist_weight_z(const flip_pcg_scalars *S, SlabWeight W, long long base, long long n, double *__restrict__ z) {
    if (S->done) return;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const long long c = base + q;
    z[c] *= d_slab_weight(W, (int)(c / W.plane));
}

// Overlapping variant: the correction a neighbour computed on its copy of these cells is added to the local one
// (cells that are rows only), and r.z is taken over the owned rows once z is final.
__global__ void k_dist_add(const flip_pcg_scalars *S, long long n, const double *__restrict__ inv, const double *__restrict__ recv,
                           double *__restrict__ z) {
    if (S->done) return;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n && inv[q] != 0.0) z[q] += recv[q];
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_dist_rz(PcgArgs A) {
    __shared__ double scratch[4];
    if (A.scal->done) return;
    long long lo, hi;
    d_row_range(A, lo, hi);
    double rz = 0.0;
    for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
        const long long c = A.rows[q];
        rz += A.r[c] * A.z[c];
    }
    rz = d_block_sum(rz, scratch);
    if (threadIdx.x == 0) A.partial[1 * FLIP_MAX_PCG_BLOCKS + blockIdx.x] = rz;
}

extern "C" int flip_pcg_dist_buffer(flip_ctx *ctx, int which, void **ptr, int64_t *count) {
    FLIP_REQUIRE(ctx && ptr && count, "flip_pcg_dist_buffer: null argument");
    FLIP_REQUIRE(ctx->dist_sums, "flip_pcg_dist_buffer: call flip_set_slab first");
    switch (which) {
    case 0: *ptr = ctx->dist_sums; *count = 16; break;
    case 1: *ptr = ctx->z; *count = ctx->B; break;
    case 2: *ptr = ctx->x; *count = ctx->B; break;
    case 3: *ptr = ctx->r; *count = ctx->B; break;
    case 4:
    case 5:
        FLIP_REQUIRE(ctx->slab_overlap > 0, "flip_pcg_dist_buffer: no overlap set (flip_set_slab_overlap)");
        *ptr = ctx->dist_recv[which - 4];
        *count = (int64_t)(ctx->slab_overlap + 1) * ctx->g.nx * ctx->g.ny;
        break;
    default: FLIP_REQUIRE(false, "flip_pcg_dist_buffer: unknown buffer");
    }
    return 0;
}

extern "C" int flip_pcg_dist_begin(flip_ctx *ctx, double tol, int32_t max_iter, int precond) {
    FLIP_REQUIRE(ctx, "null context");
    FLIP_REQUIRE(ctx->sys_kind >= 0, "flip_pcg_dist_begin: call flip_build_system first");
    FLIP_REQUIRE(ctx->dist_sums, "flip_pcg_dist_begin: call flip_set_slab first");
    FLIP_REQUIRE(precond == FLIP_PRECOND_JACOBI || precond == FLIP_PRECOND_MULTIGRID, "flip_pcg_dist_begin: unknown preconditioner");
    FLIP_REQUIRE(ctx->sys_eliminated, "flip_pcg_dist_begin: needs eliminate_dependent_rows = 1 (symmetric system)");
    FLIP_REQUIRE(tol > 0.0 && max_iter > 0, "flip_pcg_dist_begin: tol and max_iter must be positive");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    flip_pcg_scalars h;
    memset(&h, 0, sizeof h);
    h.tol2 = tol * tol;
    h.max_iter = max_iter;
    h.creg = ctx->reg_value;
    k_set_scalars<<<1, 1, 0, st>>>(ctx->scal, h); // by value: no host buffer has to outlive the call
    FLIP_LAUNCH_CHECK();
    ctx->x_kind = -1; // the slab solve starts from zero
    if (int rc = flip_zero_solver_vectors(ctx, false, st)) return rc;
    bool use_mg = precond == FLIP_PRECOND_MULTIGRID;
    if (use_mg) {
        if (int rc = flip_mg_setup(ctx)) return rc;
        if (ctx->mg.empty()) use_mg = false;
    }
    ctx->dist_mg = use_mg;
    ctx->dist_max_iter = max_iter;
    ctx->dist_active = true;
    ctx->dist_prof = 0;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = ctx->pcg_blocks;
    while (ctx->events.size() < 2 + 2 * 64) {
        hipEvent_t e;
        FLIP_CHECK(hipEventCreate(&e));
        ctx->events.push_back(e);
    }
    FLIP_CHECK(hipEventRecord(ctx->events[0], st));
    k_pcg_init<false><<<nblk, FLIP_BLOCK, 0, st>>>(A);
    FLIP_LAUNCH_CHECK();
    if (ctx->slab_overlap > 0) return 0; // the caller continues with the residual exchange, precondition() and combine()
    if (use_mg)
        if (int rc = flip_mg_vcycle(ctx)) return rc;
    k_dist_gather<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 0, 2, ctx->dist_sums);
    FLIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int flip_pcg_dist_direction(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_direction: no solve in progress");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = ctx->pcg_blocks;
    k_dist_scatter<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 0, 2, ctx->dist_sums);
    const bool prof = ctx->profiling && ctx->dist_prof < 64;
    if (prof) FLIP_CHECK(hipEventRecord(ctx->events[2 + 2 * ctx->dist_prof], st));
    k_pcg_direction<true><<<nblk, FLIP_BLOCK, 0, st>>>(A);
    if (prof) { FLIP_CHECK(hipEventRecord(ctx->events[3 + 2 * ctx->dist_prof], st)); ++ctx->dist_prof; }
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const long long lo_ghost = ctx->slab_on && ctx->slab_k0 > 0 ? (ctx->slab_k0 - 1) * plane : -1;
    const long long hi_ghost = ctx->slab_on && ctx->slab_k1 < ctx->g.nz ? ctx->slab_k1 * plane : -1;
    if (lo_ghost >= 0 || hi_ghost >= 0)
        k_dist_ghost_direction<<<flip_grid_blocks(plane), FLIP_BLOCK, 0, st>>>(ctx->scal, plane, lo_ghost, hi_ghost, ctx->z,
                                                                              ctx->pv[0], ctx->pv[1]);
    // p.Ap and, with a rigid body, the six sums J p (the rank-6 term couples rows of every slab)
    k_dist_gather<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 2, ctx->sys_coupled ? 7 : 1, ctx->dist_sums);
    FLIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int flip_pcg_dist_update(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_update: no solve in progress");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = ctx->pcg_blocks;
    k_dist_scatter<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 2, ctx->sys_coupled ? 7 : 1, ctx->dist_sums);
    k_pcg_update<<<nblk, FLIP_BLOCK, 0, st>>>(A);
    FLIP_LAUNCH_CHECK();
    if (ctx->slab_overlap > 0) return 0;
    if (ctx->dist_mg)
        if (int rc = flip_mg_vcycle(ctx)) return rc;
    k_dist_gather<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 0, 2, ctx->dist_sums);
    FLIP_LAUNCH_CHECK();
    return 0;
}

// Overlapping variant, after the neighbours' residuals have arrived in the overlap planes of buffer 3: the local
// preconditioner over the owned planes and the overlap (z = Jacobi of the owned rows already when it is Jacobi).
extern "C" int flip_pcg_dist_precondition(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_precondition: no solve in progress");
    FLIP_REQUIRE(ctx->slab_overlap > 0, "flip_pcg_dist_precondition: no overlap set (flip_set_slab_overlap)");
    FLIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->dist_mg) return 0;
    hipStream_t st = ctx->stream;
    const SlabWeight W = slab_weight(ctx);
    if (ctx->slab_weighted) {
        k_dist_weight_r<<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(ctx->scal, W, ctx->mg_rows, ctx->d_counts + 7, ctx->r, ctx->dist_rw);
        FLIP_LAUNCH_CHECK();
    }
    if (int rc = flip_mg_vcycle(ctx)) return rc;
    if (ctx->slab_weighted) {
        const long long n = 2LL * W.m * W.plane;
        if (W.has_lo) k_dist_weight_z<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(ctx->scal, W, (long long)(W.k0 - W.m) * W.plane, n, ctx->z);
        if (W.has_hi) k_dist_weight_z<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(ctx->scal, W, (long long)(W.k1 - W.m) * W.plane, n, ctx->z);
        FLIP_LAUNCH_CHECK();
    }
    return 0;
}

// Overlapping variant, after the neighbours' corrections have arrived in buffers 4 (from the lower neighbour, for
// planes [slab_k0 - 1, slab_k0 + overlap)) and 5 (from the upper one, for planes [slab_k1 - overlap, slab_k1 + 1)):
// adds them, which also completes z on the first ghost plane of either side, and leaves rr and r.z in buffer 0.
extern "C" int flip_pcg_dist_combine(flip_ctx *ctx) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_combine: no solve in progress");
    FLIP_REQUIRE(ctx->slab_overlap > 0, "flip_pcg_dist_combine: no overlap set (flip_set_slab_overlap)");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = ctx->pcg_blocks;
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const long long n = (long long)(ctx->slab_overlap + 1) * plane;
    if (ctx->slab_k0 > 0) {
        const long long base = (long long)(ctx->slab_k0 - 1) * plane;
        k_dist_add<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(ctx->scal, n, ctx->cg_inv + base, ctx->dist_recv[0], ctx->z + base);
    }
    if (ctx->slab_k1 < ctx->g.nz) {
        const long long base = (long long)(ctx->slab_k1 - ctx->slab_overlap) * plane;
        k_dist_add<<<flip_grid_blocks(n), FLIP_BLOCK, 0, st>>>(ctx->scal, n, ctx->cg_inv + base, ctx->dist_recv[1], ctx->z + base);
    }
    k_dist_rz<<<nblk, FLIP_BLOCK, 0, st>>>(A);
    k_dist_gather<<<1, FLIP_BLOCK, 0, st>>>(ctx->partial, nblk, 0, 2, ctx->dist_sums);
    FLIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int flip_pcg_dist_status(flip_ctx *ctx, int32_t *done, int32_t *iterations) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_status: no solve in progress");
    FLIP_CHECK(hipSetDevice(ctx->device));
    flip_pcg_scalars h;
    FLIP_CHECK(hipMemcpyAsync(&h, ctx->scal, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FLIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (done) *done = h.done;
    if (iterations) *iterations = h.it_a;
    return 0;
}

// The caller has exchanged the ghost planes of x (buffer 2) before this call: the dependent rows next to a slab
// face read their liquid neighbours there.
extern "C" int flip_pcg_dist_finish(flip_ctx *ctx, flip_pcg_info *info) {
    FLIP_REQUIRE(ctx && ctx->dist_active, "flip_pcg_dist_finish: no solve in progress");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PcgArgs A;
    fill_pcg_args(ctx, A);
    k_pcg_dependent<false><<<ctx->pcg_blocks, FLIP_BLOCK, 0, st>>>(A);
    FLIP_LAUNCH_CHECK();
    FLIP_CHECK(hipEventRecord(ctx->events[1], st));
    flip_pcg_scalars h;
    FLIP_CHECK(hipMemcpyAsync(&h, ctx->scal, sizeof h, hipMemcpyDeviceToHost, st));
    int32_t counts[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    FLIP_CHECK(hipMemcpyAsync(counts, ctx->d_counts, 32, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipStreamSynchronize(st));
    ctx->dist_active = false;
    if (info) {
        memset(info, 0, sizeof *info);
        info->converged = h.done == 1 ? 1 : 0;
        info->iterations = h.it_a;
        info->rows = counts[0];
        info->active_chunks = counts[1];
        info->dependent_rows = counts[4];
        info->cg_rows = counts[6];
        info->rhs_norm_2 = sqrt(h.rhs_norm2);
        info->rel_residual_2 = h.rhs_norm2 > 0 ? sqrt(h.rr / h.rhs_norm2) : 0.0;
        info->iterations_inf = -1; // a maximum over the slabs is not exchanged
        float ms = 0.f;
        FLIP_CHECK(hipEventElapsedTime(&ms, ctx->events[0], ctx->events[1]));
        info->seconds = (double)ms * 1e-3;
        double sp = 0.0;
        int used = 0;
        for (int q = 0; q < ctx->dist_prof && q < h.it_a; ++q) {
            float t = 0.f;
            FLIP_CHECK(hipEventElapsedTime(&t, ctx->events[2 + 2 * q], ctx->events[3 + 2 * q]));
            sp += (double)t * 1e-3;
            ++used;
        }
        info->spmv_seconds = sp;
        info->spmv_launches = used;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Parity hooks

__global__ void k_apply_plain(PcgArgs A, long long own_lo, long long own_hi, const double *__restrict__ x, double *__restrict__ y,
                              double *__restrict__ jp_out) {
    __shared__ double scratch[4];
    double jp[6] = { 0, 0, 0, 0, 0, 0 };
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < A.B; c += (long long)gridDim.x * blockDim.x) {
        const uint8_t fl = A.flags[c];
        double s = 0.0;
        if (fl & FLIP_FLAG_ROW) {
            s = d_apply_row(A, c, fl, x[c], [&](long long n) { return x[n]; });
            if (A.coupled && c >= own_lo && c < own_hi) // the rows of a slab's ghost planes belong to the neighbours' sums
#pragma unroll
                for (int a = 0; a < 6; ++a) jp[a] += A.J[a][c] * x[c];
        }
        y[c] = s;
    }
    if (A.coupled)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double v = d_block_sum(jp[a], scratch);
            if (threadIdx.x == 0) jp_out[a * gridDim.x + blockIdx.x] = v;
        }
}
__global__ void k_apply_coupling(PcgArgs A, double g0, double g1, double g2, double g3, double g4, double g5, double *__restrict__ y) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.B || !(A.flags[c] & FLIP_FLAG_ROW)) return;
    y[c] += A.J[0][c] * g0 + A.J[1][c] * g1 + A.J[2][c] * g2 + A.J[3][c] * g3 + A.J[4][c] * g4 + A.J[5][c] * g5;
}

// J * v over the rows for a dense device vector v (uses ctx->ap as scratch-free reduction target)
static int coupling_jv(flip_ctx *ctx, const double *dev_v, double *dev_y, double jv[6]) {
    PcgArgs A;
    fill_pcg_args(ctx, A);
    const int nblk = 256;
    const long long plane = (long long)ctx->g.nx * ctx->g.ny;
    const long long own_lo = ctx->slab_on ? ctx->slab_k0 * plane : 0, own_hi = ctx->slab_on ? ctx->slab_k1 * plane : ctx->B;
    k_apply_plain<<<nblk, FLIP_BLOCK, 0, ctx->stream>>>(A, own_lo, own_hi, dev_v, dev_y, ctx->d_vmax);
    FLIP_LAUNCH_CHECK();
    for (int a = 0; a < 6; ++a) jv[a] = 0.0;
    if (A.coupled) {
        std::vector<double> h(6 * nblk);
        FLIP_CHECK(hipMemcpyAsync(h.data(), ctx->d_vmax, sizeof(double) * 6 * nblk, hipMemcpyDeviceToHost, ctx->stream));
        FLIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < nblk; ++b) jv[a] += h[a * nblk + b];
    }
    return 0;
}

int flip_pcg_coupling_jp(flip_ctx *ctx, double jp[6]) {
    FLIP_REQUIRE(ctx->sys_coupled, "no coupled system assembled");
    return coupling_jv(ctx, ctx->x, ctx->ap, jp);
}

// The rigid body under a slab decomposition: every rank holds the same body; its velocity update V += dt Minv (J p)
// (src/FluidSimulator.cpp:1095-1100) needs J p over the rows of ALL slabs.  flip_solid_coupling_partial returns this
// context's share (rows of its owned planes, after the solve), the caller sums the six numbers over the ranks and hands
// the total to flip_solid_velocity_apply on every rank.  On a slab context flip_project_velocity leaves the body alone.
extern "C" int flip_solid_coupling_partial(flip_ctx *ctx, double jp[6]) {
    FLIP_REQUIRE(ctx && jp, "flip_solid_coupling_partial: null argument");
    FLIP_REQUIRE(ctx->has_obj && ctx->sys_coupled, "flip_solid_coupling_partial: no coupled divergence system is active");
    FLIP_CHECK(hipSetDevice(ctx->device));
    return flip_pcg_coupling_jp(ctx, jp);
}
extern "C" int flip_solid_velocity_apply(flip_ctx *ctx, float dt, const double jp_total[6]) {
    FLIP_REQUIRE(ctx && jp_total, "flip_solid_velocity_apply: null argument");
    FLIP_REQUIRE(ctx->has_obj, "flip_solid_velocity_apply: no rigid body attached");
    for (int a = 0; a < 6; ++a) ctx->obj_vel[a] = ctx->obj_vel[a] + (double)dt * (ctx->obj_minv[a] * jp_total[a]);
    return 0;
}

extern "C" int flip_apply_operator(flip_ctx *ctx, const double *x_host, double *y_host) {
    FLIP_REQUIRE(ctx && x_host && y_host, "null argument");
    FLIP_REQUIRE(ctx->sys_kind >= 0, "flip_apply_operator: call flip_build_system first");
    FLIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // p0 holds x, p1 receives y (both are solver scratch)
    FLIP_CHECK(hipMemcpyAsync(ctx->pv[0], x_host, (size_t)ctx->B * 8, hipMemcpyHostToDevice, st));
    double jv[6];
    if (int rc = coupling_jv(ctx, ctx->pv[0], ctx->pv[1], jv)) return rc;
    if (ctx->sys_coupled) {
        PcgArgs A;
        fill_pcg_args(ctx, A);
        double gq[6];
        for (int a = 0; a < 6; ++a) gq[a] = (double)ctx->sys_dt * ctx->obj_minv[a] * jv[a];
        k_apply_coupling<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, st>>>(A, gq[0], gq[1], gq[2], gq[3], gq[4], gq[5], ctx->pv[1]);
        FLIP_LAUNCH_CHECK();
    }
    FLIP_CHECK(hipMemcpyAsync(y_host, ctx->pv[1], (size_t)ctx->B * 8, hipMemcpyDeviceToHost, st));
    FLIP_CHECK(hipStreamSynchronize(st));
    return 0;
}
