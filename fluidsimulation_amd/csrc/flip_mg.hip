// flip_mg.hip -- geometric multigrid V-cycle used as the CG preconditioner (FLIP_PRECOND_MULTIGRID).
//
// It replaces the sequential MIC(0) factorisation and triangular solves of headers/pcgsolver/pcg_solver.h
// (:80-214) with a preconditioner whose every step is data parallel: the reference has no counterpart.
//
//   hierarchy   : cell-centred, 2x2x2 coarsening of the dense box; level 0 is the CG view of the assembled
//                 system (symmetric couplings of the CG rows).  Coarse operator = w * P^T A P with P the
//                 piecewise-constant prolongation; with w = 1/2 the coupling of two coarse cells is half the sum of
//                 the four fine couplings across their common face, which is also what re-discretising the
//                 finite-volume operator on the coarse grid gives (face area x4, distance x2); w = 0.54 is used
//                 (mg_wgal: measured).
//   smoother    : degree-2 Chebyshev polynomial in D^-1 A (two operator applications) before and after the coarse
//                 correction, the same polynomial both times, so the V-cycle is a symmetric operator as CG
//                 requires; the pre-smoother starts from zero.
//   transfer    : restriction = sum of the eight child residuals (the equations are volume integrals),
//                 prolongation = injection of the parent value.
//   coarsest    : a fixed number of Chebyshev sweeps by one workgroup, vectors in LDS.
// Every level keeps dense, zero-padded arrays and a compact list of its rows, like level 0.
#include "flip_internal.h"

#include <cstdlib>

// Smoother: the degree-2 Chebyshev polynomial in D^-1 A for the eigenvalue interval [lo, 2] (every level's operator
// is weakly diagonally dominant, so the spectrum of D^-1 A lies in (0, 2]; what lies below `lo` is left to the
// coarse levels).  As a three-term recurrence over the iterates,
//   x1 = x0 + c1 D^-1 (b - A x0),   x2 = x1 + a2 (x1 - x0) + c2 D^-1 (b - A x1),
// with theta = (2 + lo)/2, delta = (2 - lo)/2: c1 = 1/theta, rho0 = delta/theta, rho1 = 1/(2 theta/delta - rho0),
// a2 = rho1 rho0, c2 = 2 rho1/delta.  For lo = 0.5 it damps the interval by 0.22 per application where two Jacobi
// sweeps with weight 0.8 reach 0.36, at the same two operator applications.  Pre- and post-smoothing use the same
// polynomial, so the V-cycle stays a symmetric operator.  lo = 0.3 gives the fewest CG iterations on the dam break at
// a time step within the CFL limit (256^3: 0.25: 51.7, 0.3: 50.9, 0.35: 51.05, 0.4: 51.7, 0.5: 53.7, 0.6: 56.6 per
// substep; 512^3: 55.3 / 55.7 / 56.3 for 0.3 / 0.35 / 0.4; 128^3: 44.8 / 44.4 / 44.6); with the five times larger
// step of the earlier rounds it was 0.4 (0.3: 63.7, 0.4: 62, 0.5: 64.3, 0.65: 71, 0.8: 77; two Jacobi sweeps: 78).
// The lower end of the interval can be overridden with FLIP_MG_LAMBDA_LO for tuning.
struct MgCheb { double c1, a2, c2; };
static MgCheb mg_cheb() {
    static MgCheb k = [] {
        double lo = 0.3;
        if (const char *e = getenv("FLIP_MG_LAMBDA_LO")) { const double v = atof(e); if (v > 0.0 && v < 2.0) lo = v; }
        const double theta = 0.5 * (2.0 + lo), delta = 0.5 * (2.0 - lo);
        const double rho0 = delta / theta, rho1 = 1.0 / (2.0 * theta / delta - rho0);
        return MgCheb{ 1.0 / theta, rho1 * rho0, 2.0 * rho1 / delta };
    }();
    return k;
}
void flip_mg_smoother(double *c1, double *a2, double *c2) { const MgCheb k = mg_cheb(); *c1 = k.c1; *a2 = k.a2; *c2 = k.c2; }
#define MG_COARSEST_CELLS 4096   // one workgroup of 1024 threads, at most 4 cells per thread
#define MG_COARSEST_LDS (60 * 1024) // its two padded vectors must fit the default dynamic LDS limit
#define MG_COARSEST_SWEEPS 12          // 24 gave the same CG iteration count at 256^3 (61.8 against 61.0 per substep), 8 costs 3 more
#define MG_COARSEST_KAPPA 100.0  // Chebyshev target interval of the coarsest solve: [2/kappa, 2]

template <class CT>
struct MgDev {
    long long B, sy, sz;
    int nx, ny, nz;
    const CT *cx, *cy, *cz;   // coupling toward the low x/y/z neighbour (0 = none)
    double scale;             // couplings are multiplied by this (dx^2 of the divergence system on level 0)
    double c1, a2, c2;        // smoother coefficients
    const CT *diag, *inv;     // diagonal and its inverse (0 off the rows); level 0 reads float copies: the
                              // preconditioner does not need the last bits and these are a sixth of its traffic
    double *x, *b, *t;
    const uint8_t *reg;              // level 0 only: rows whose six couplings all equal S->creg (nullptr: none)
    const double *b_alt;             // level 0 only: the second residual buffer of the fused update kernel
    const flip_pcg_scalars *S;       // level 0 only: says which of b / b_alt is current (nullptr: always b)
    const int32_t *rows;
    const int32_t *n_rows;
};

// The right-hand side of a level: on level 0 the CG residual, which alternates between two buffers when the update
// kernel is the fused one (the buffer written by update number it_a - 1 is number it_a & 1).
template <class CT>
__device__ __forceinline__ const double *d_mg_rhs(const MgDev<CT> &L) {
    return (L.S && L.S->fused && (L.S->it_a & 1)) ? L.b_alt : L.b;
}

// The six couplings of cell c (toward -x, +x, -y, +y, -z, +z): one common value for a row flagged regular, which
// spares the three array reads, else from the arrays.
template <class CT>
__device__ __forceinline__ void d_mg_couplings(const MgDev<CT> &L, long long c, double (&a)[6]) {
    if (L.reg && L.reg[c]) {
        const double s = (double)L.S->creg;
        a[0] = a[1] = a[2] = a[3] = a[4] = a[5] = s;
    } else {
        a[0] = (double)L.cx[c]; a[1] = (double)L.cx[c + 1];
        a[2] = (double)L.cy[c]; a[3] = (double)L.cy[c + L.sy];
        a[4] = (double)L.cz[c]; a[5] = (double)L.cz[c + L.sz];
    }
}

template <class CT>
__device__ __forceinline__ double d_mg_apply(const MgDev<CT> &L, long long c, const double *__restrict__ v) {
    double a[6];
    d_mg_couplings(L, c, a);
    double acc = a[0] * v[c - 1];
    acc += a[1] * v[c + 1];
    acc += a[2] * v[c - L.sy];
    acc += a[3] * v[c + L.sy];
    acc += a[4] * v[c - L.sz];
    acc += a[5] * v[c + L.sz];
    return (double)L.diag[c] * v[c] - L.scale * acc;
}

// Pre-smoothing, both steps of the smoother from a zero guess in one kernel:
//   t = C1 D^-1 b  (evaluated on the fly at the 7 stencil points),  out = t + A2 t + C2 D^-1 (b - A t)
template <class CT>
__device__ __forceinline__ void d_mg_pre(const MgDev<CT> &L, double *__restrict__ out, long long first, long long stride, long long n) {
    const CT *__restrict__ inv = L.inv;
    const double *__restrict__ b = d_mg_rhs(L);
    // the row list is read one step ahead: a row's loads then start without waiting for its index
    long long c_next = first < n ? L.rows[first] : 0;
#pragma unroll 2
    for (long long q = first; q < n; q += stride) {
        const long long c = c_next;
        if (q + stride < n) c_next = L.rows[q + stride];
        const long long sy = L.sy, sz = L.sz;
        const double bc = b[c], ic = (double)inv[c];
        const double t0 = L.c1 * ic * bc;
        double a[6];
        d_mg_couplings(L, c, a);
        double acc = a[0] * (L.c1 * (double)inv[c - 1] * b[c - 1]);
        acc += a[1] * (L.c1 * (double)inv[c + 1] * b[c + 1]);
        acc += a[2] * (L.c1 * (double)inv[c - sy] * b[c - sy]);
        acc += a[3] * (L.c1 * (double)inv[c + sy] * b[c + sy]);
        acc += a[4] * (L.c1 * (double)inv[c - sz] * b[c - sz]);
        acc += a[5] * (L.c1 * (double)inv[c + sz] * b[c + sz]);
        const double At = (double)L.diag[c] * t0 - L.scale * acc;
        out[c] = t0 + L.a2 * t0 + L.c2 * ic * (bc - At);
    }
}
// TOP in the list-driven kernels: the level is level 0 of a solve (it has the regular-row flags, the alternating
// residual buffers and the solver's scalars); the coarse levels that read float operator copies are the same code
// under another name, which also keeps the two apart in a profile.
template <class CT, bool TOP>
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_pre(MgDev<CT> L, const int32_t *done, double *__restrict__ out) {
    // the stop flag and the row count are independent loads: both are in flight before either is looked at
    const int stop = done ? *done : 0;
    const long long n = *L.n_rows;
    if (stop) return;
    if (!TOP) { L.reg = nullptr; L.S = nullptr; }   // known at compile time below level 0: the branches on them fold away
    d_mg_pre(L, out, (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x, (long long)gridDim.x * FLIP_BLOCK, n);
}

// Second post-smoothing step: out = in + A2 (in - v) + C2 D^-1 (b - A in), where v = out_old + P xc is the corrected
// iterate the first step started from (rebuilt at the row itself: `out` still holds the pre-smoothed value there).
// RZ: also the per-workgroup partial of b.out into `partial`.
template <class CT>
__device__ __forceinline__ double d_mg_second_step(const MgDev<CT> &L, const MgDev<double> &C, long long c, const double *__restrict__ in,
                                                   const double *__restrict__ out, double b) {
    // 32-bit index arithmetic: a level has fewer than 2^31 cells, and a 64-bit division costs several times a 32-bit one
    const unsigned cu = (unsigned)c, tq = cu / (unsigned)L.nx;
    const int i = (int)(cu - tq * (unsigned)L.nx);
    const int k = (int)(tq / (unsigned)L.ny), j = (int)(tq - (unsigned)k * (unsigned)L.ny);
    const double v0 = out[c] + C.x[(i >> 1) + C.sy * (j >> 1) + C.sz * (k >> 1)];
    const double x1 = in[c];
    return x1 + L.a2 * (x1 - v0) + L.c2 * (double)L.inv[c] * (b - d_mg_apply(L, c, in));
}
template <class CT, bool RZ, bool TOP>
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_smooth(MgDev<CT> L, MgDev<double> C, const int32_t *done,
                                                         const double *__restrict__ in, double *out, double *__restrict__ partial) {
    __shared__ double scratch[4];
    // the stop flag and the row count are independent loads: both are in flight before either is looked at
    const int stop = done ? *done : 0;
    const long long n = *L.n_rows;
    if (stop) return;
    if (!TOP) { L.reg = nullptr; L.S = nullptr; }
    const double *__restrict__ rhs = d_mg_rhs(L);
    double acc = 0.0;
    if (RZ) {
        // same split of the row list as the CG kernels, so the partial layout matches their reduction
        const int per = gridDim.x >> 3;
        const int rank = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        const long long lo = (n * rank) / gridDim.x, hi = (n * (rank + 1)) / gridDim.x;
        long long c_next = lo + threadIdx.x < hi ? L.rows[lo + threadIdx.x] : 0;
#pragma unroll 2
        for (long long q = lo + threadIdx.x; q < hi; q += FLIP_BLOCK) {
            const long long c = c_next;
            if (q + FLIP_BLOCK < hi) c_next = L.rows[q + FLIP_BLOCK];
            const double b = rhs[c];
            const double v = d_mg_second_step(L, C, c, in, out, b);
            out[c] = v;
            acc += b * v;
        }
        acc = d_block_sum(acc, scratch);
        if (threadIdx.x == 0) partial[blockIdx.x] = acc;
    } else {
        const long long first = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x, stride = (long long)gridDim.x * FLIP_BLOCK;
        long long c_next = first < n ? L.rows[first] : 0;
#pragma unroll 2
        for (long long q = first; q < n; q += stride) {
            const long long c = c_next;
            if (q + stride < n) c_next = L.rows[q + stride];
            out[c] = d_mg_second_step(L, C, c, in, out, rhs[c]);
        }
    }
}
// Coarse correction + first post-smoothing step in one kernel: with v = in + P xc (the parent's value
// added on the fly at the 7 stencil points), out = v + C1 D^-1 (b - A v).
template <class CT>
__device__ __forceinline__ void d_mg_post(const MgDev<CT> &F, const MgDev<double> &C, const double *__restrict__ in, double *__restrict__ out,
                                          long long first, long long stride, long long n) {
    const double *__restrict__ xc = C.x;
    const double *__restrict__ rhs = d_mg_rhs(F);
    long long c_next = first < n ? F.rows[first] : 0;
#pragma unroll 2
    for (long long q = first; q < n; q += stride) {
        const long long c = c_next;
        if (q + stride < n) c_next = F.rows[q + stride];
        const unsigned cu = (unsigned)c, tq = cu / (unsigned)F.nx;
        const int i = (int)(cu - tq * (unsigned)F.nx);
        const int k = (int)(tq / (unsigned)F.ny), j = (int)(tq - (unsigned)k * (unsigned)F.ny);
        const long long pj = C.sy * (j >> 1), pk = C.sz * (k >> 1);
        const long long pc = (i >> 1) + pj + pk;
        // parents of the six neighbours (an arithmetic shift keeps -1 at -1: padded, and its coupling is 0)
        const long long pxm = ((i - 1) >> 1) + pj + pk, pxp = ((i + 1) >> 1) + pj + pk;
        const long long pym = (i >> 1) + C.sy * ((j - 1) >> 1) + pk, pyp = (i >> 1) + C.sy * ((j + 1) >> 1) + pk;
        const long long pzm = (i >> 1) + pj + C.sz * ((k - 1) >> 1), pzp = (i >> 1) + pj + C.sz * ((k + 1) >> 1);
        const long long sy = F.sy, sz = F.sz;
        const double v0 = in[c] + xc[pc];
        double a[6];
        d_mg_couplings(F, c, a);
        double acc = a[0] * (in[c - 1] + xc[pxm]);
        acc += a[1] * (in[c + 1] + xc[pxp]);
        acc += a[2] * (in[c - sy] + xc[pym]);
        acc += a[3] * (in[c + sy] + xc[pyp]);
        acc += a[4] * (in[c - sz] + xc[pzm]);
        acc += a[5] * (in[c + sz] + xc[pzp]);
        const double Av = (double)F.diag[c] * v0 - F.scale * acc;
        out[c] = v0 + F.c1 * (double)F.inv[c] * (rhs[c] - Av);
    }
}
template <class CT, bool TOP>
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_post(MgDev<CT> F, MgDev<double> C, const int32_t *done,
                                                       const double *__restrict__ in, double *__restrict__ out) {
    const int stop = done ? *done : 0;
    const long long n = *F.n_rows;
    if (stop) return;
    if (!TOP) { F.reg = nullptr; F.S = nullptr; }
    d_mg_post(F, C, in, out, (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x, (long long)gridDim.x * FLIP_BLOCK, n);
}

// coarse b = sum over the eight children of (b - A v) on the fine level.  Eight lanes per coarse row, one per
// child, summed by three exchange steps in a fixed order.
template <class CT>
__device__ __forceinline__ void d_mg_restrict(const MgDev<CT> &F, const MgDev<double> &C, const double *__restrict__ v, long long first_thread,
                                              long long total_threads, long long n) {
    const double *__restrict__ rhs = d_mg_rhs(F);
    const int child = threadIdx.x & 7;
    const int di = child & 1, dj = (child >> 1) & 1, dk = child >> 2;
    long long cc_next = (first_thread >> 3) < n ? C.rows[first_thread >> 3] : 0;
    for (long long q = first_thread >> 3; q < ((n + 31) & ~31LL); q += total_threads >> 3) {
        double s = 0.0;
        long long cc = 0;
        const long long cc_this = cc_next;
        if (q + (total_threads >> 3) < n) cc_next = C.rows[q + (total_threads >> 3)];
        if (q < n) {
            cc = cc_this;
            const unsigned cu = (unsigned)cc, tq = cu / (unsigned)C.nx;
            const int I = (int)(cu - tq * (unsigned)C.nx);
            const int K = (int)(tq / (unsigned)C.ny), J = (int)(tq - (unsigned)K * (unsigned)C.ny);
            const int i = 2 * I + di, j = 2 * J + dj, k = 2 * K + dk;
            if (i < F.nx && j < F.ny && k < F.nz) {
                const long long c = (long long)i + F.sy * j + F.sz * k;
                if (F.inv[c] != (CT)0) s = rhs[c] - d_mg_apply(F, c, v);
            }
        }
        s += __shfl_xor(s, 1, FLIP_WAVE);
        s += __shfl_xor(s, 2, FLIP_WAVE);
        s += __shfl_xor(s, 4, FLIP_WAVE);
        if (q < n && child == 0) C.b[cc] = s;
    }
}
template <class CT, bool TOP>
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_restrict(MgDev<CT> F, MgDev<double> C, const int32_t *done,
                                                           const double *__restrict__ v) {
    const int stop = done ? *done : 0;
    const long long n = *C.n_rows;
    if (stop) return;
    if (!TOP) { F.reg = nullptr; F.S = nullptr; }
    d_mg_restrict(F, C, v, (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x, (long long)gridDim.x * FLIP_BLOCK, n);
}

// Coarsest level: one workgroup, vectors in LDS (dense over the level's cells, zero-padded so that the
// stencil needs no bounds test), a fixed number of Chebyshev sweeps on D^-1 A from zero.  The spectrum of
// D^-1 A lies in (0, 2] because every level's operator is weakly diagonally dominant; the sweeps target
// [2 / MG_COARSEST_KAPPA, 2].  A fixed polynomial in D^-1 A applied to D^-1 b is a symmetric positive
// definite operator, as CG requires of its preconditioner.  Each thread keeps the coefficients of its
// (at most 4) cells in registers:
//   x_{k+1} = x_k + a_k (x_k - x_{k-1}) + c_k D^-1 (b - A x_k)
__device__ __forceinline__ void d_mg_coarsest(const MgDev<double> &L, int pad, double *lds) {
    const int B = (int)L.B, sy = (int)L.sy, sz = (int)L.sz;
    double *xa = lds + pad, *xb = lds + (B + 2 * pad) + pad;
    for (int q = threadIdx.x; q < 2 * (B + 2 * pad); q += blockDim.x) lds[q] = 0.0;
    double k0[4], kd[4], kc[4][6];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = threadIdx.x + m * 1024;
        k0[m] = 0.0; kd[m] = 0.0;
#pragma unroll
        for (int f = 0; f < 6; ++f) kc[m][f] = 0.0;
        if (c < B) {
            const double wi = L.inv[c];
            k0[m] = wi * L.b[c];
            kd[m] = wi * L.diag[c];
            kc[m][0] = wi * L.cx[c]; kc[m][1] = wi * L.cx[c + 1];
            kc[m][2] = wi * L.cy[c]; kc[m][3] = wi * L.cy[c + sy];
            kc[m][4] = wi * L.cz[c]; kc[m][5] = wi * L.cz[c + sz];
        }
    }
    const double lmax = 2.0, lmin = 2.0 / MG_COARSEST_KAPPA;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    __syncthreads();
    // x_0 = 0 (xb), x_1 = D^-1 b / theta (xa)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = threadIdx.x + m * 1024;
        if (c < B) xa[c] = k0[m] / theta;
    }
    __syncthreads();
    for (int s = 1; s < MG_COARSEST_SWEEPS; ++s) {
        const double rho_new = 1.0 / (2.0 * sigma - rho);
        const double ak = rho_new * rho, ck = 2.0 * rho_new / delta;
        rho = rho_new;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c = threadIdx.x + m * 1024;
            if (c < B) {
                const double xc = xa[c];
                double res = k0[m] - kd[m] * xc;
                res += kc[m][0] * xa[c - 1];
                res += kc[m][1] * xa[c + 1];
                res += kc[m][2] * xa[c - sy];
                res += kc[m][3] * xa[c + sy];
                res += kc[m][4] * xa[c - sz];
                res += kc[m][5] * xa[c + sz];
                xb[c] = xc + ak * (xc - xb[c]) + ck * res; // overwrites x_{k-1}, which only this thread reads
            }
        }
        __syncthreads();
        double *tmp = xa; xa = xb; xb = tmp;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = threadIdx.x + m * 1024;
        if (c < B) L.x[c] = xa[c];
    }
}
__global__ void __launch_bounds__(1024) k_mg_coarsest(MgDev<double> L, const int32_t *done, int pad) {
    extern __shared__ double lds[];
    if (done && *done) return;
    d_mg_coarsest(L, pad, lds);
}

// The small levels cost launch latency plus the latency of every *dependent* load, not bandwidth: the list-driven
// kernels above read the stop flag, then the row count, then the row's cell, then the cell's data -- four round trips
// for a few thousand rows.  The forms below take one thread per cell of the dense level instead (the cell index is the
// thread index), issue every load at once, and test the stop flag and whether the cell is a row only before the store:
// one round trip.  A cell that is not a row has zero diagonal, inverse, couplings and right-hand side (k_mg_coarsen,
// k_mg_prune, the memsets of mg_coarsen_into), so the loads are harmless there.  Same expressions per row as the
// list-driven forms, hence the same results bit for bit.
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_pre_dense(MgDev<double> L, const int32_t *done, double *__restrict__ out) {
    const int stop = *done;
    const long long c = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x;
    if (c >= L.B) return;
    const long long sy = L.sy, sz = L.sz;
    const double *__restrict__ b = L.b, *__restrict__ inv = L.inv;
    const double bc = b[c], ic = inv[c];
    const double t0 = L.c1 * ic * bc;
    double acc = L.cx[c] * (L.c1 * inv[c - 1] * b[c - 1]);
    acc += L.cx[c + 1] * (L.c1 * inv[c + 1] * b[c + 1]);
    acc += L.cy[c] * (L.c1 * inv[c - sy] * b[c - sy]);
    acc += L.cy[c + sy] * (L.c1 * inv[c + sy] * b[c + sy]);
    acc += L.cz[c] * (L.c1 * inv[c - sz] * b[c - sz]);
    acc += L.cz[c + sz] * (L.c1 * inv[c + sz] * b[c + sz]);
    const double At = L.diag[c] * t0 - L.scale * acc;
    const double v = t0 + L.a2 * t0 + L.c2 * ic * (bc - At);
    if (!stop && ic != 0.0) out[c] = v;
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_post_dense(MgDev<double> F, MgDev<double> C, const int32_t *done,
                                                             const double *__restrict__ in, double *__restrict__ out) {
    const int stop = *done;
    const long long c = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x;
    if (c >= F.B) return;
    const double *__restrict__ xc = C.x;
    const unsigned cu = (unsigned)c, tq = cu / (unsigned)F.nx;
    const int i = (int)(cu - tq * (unsigned)F.nx);
    const int k = (int)(tq / (unsigned)F.ny), j = (int)(tq - (unsigned)k * (unsigned)F.ny);
    const long long pj = C.sy * (j >> 1), pk = C.sz * (k >> 1);
    const long long pc = (i >> 1) + pj + pk;
    const long long pxm = ((i - 1) >> 1) + pj + pk, pxp = ((i + 1) >> 1) + pj + pk;
    const long long pym = (i >> 1) + C.sy * ((j - 1) >> 1) + pk, pyp = (i >> 1) + C.sy * ((j + 1) >> 1) + pk;
    const long long pzm = (i >> 1) + pj + C.sz * ((k - 1) >> 1), pzp = (i >> 1) + pj + C.sz * ((k + 1) >> 1);
    const long long sy = F.sy, sz = F.sz;
    const double ic = F.inv[c];
    const double v0 = in[c] + xc[pc];
    double acc = F.cx[c] * (in[c - 1] + xc[pxm]);
    acc += F.cx[c + 1] * (in[c + 1] + xc[pxp]);
    acc += F.cy[c] * (in[c - sy] + xc[pym]);
    acc += F.cy[c + sy] * (in[c + sy] + xc[pyp]);
    acc += F.cz[c] * (in[c - sz] + xc[pzm]);
    acc += F.cz[c + sz] * (in[c + sz] + xc[pzp]);
    const double Av = F.diag[c] * v0 - F.scale * acc;
    const double v = v0 + F.c1 * ic * (F.b[c] - Av);
    if (!stop && ic != 0.0) out[c] = v;
}
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_smooth_dense(MgDev<double> L, MgDev<double> C, const int32_t *done,
                                                               const double *__restrict__ in, double *out) {
    const int stop = *done;
    const long long c = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x;
    if (c >= L.B) return;
    const double ic = L.inv[c];
    const double v = d_mg_second_step(L, C, c, in, out, L.b[c]);
    if (!stop && ic != 0.0) out[c] = v;
}
// eight lanes per cell of the coarse level C, one per child
__global__ void __launch_bounds__(FLIP_BLOCK) k_mg_restrict_dense(MgDev<double> F, MgDev<double> C, const int32_t *done,
                                                                 const double *__restrict__ v) {
    const int stop = *done;
    const long long t = (long long)blockIdx.x * FLIP_BLOCK + threadIdx.x;
    const long long cc = t >> 3;
    const int child = threadIdx.x & 7;
    double s = 0.0;
    if (cc < C.B) {
        const unsigned cu = (unsigned)cc, tq = cu / (unsigned)C.nx;
        const int I = (int)(cu - tq * (unsigned)C.nx);
        const int K = (int)(tq / (unsigned)C.ny), J = (int)(tq - (unsigned)K * (unsigned)C.ny);
        const int i = 2 * I + (child & 1), j = 2 * J + ((child >> 1) & 1), k = 2 * K + (child >> 2);
        if (i < F.nx && j < F.ny && k < F.nz) {
            const long long c = (long long)i + F.sy * j + F.sz * k;
            const double ic = F.inv[c];
            const double res = F.b[c] - d_mg_apply(F, c, v);
            s = ic != 0.0 ? res : 0.0;
        }
    }
    s += __shfl_xor(s, 1, FLIP_WAVE);
    s += __shfl_xor(s, 2, FLIP_WAVE);
    s += __shfl_xor(s, 4, FLIP_WAVE);
    // a coarse cell none of whose children is a row is not a row itself and receives the zero it already holds
    if (!stop && cc < C.B && child == 0) C.b[cc] = s;
}

// Galerkin coarsening with piecewise-constant transfer, scaled by wgal (see the header comment and mg_wgal).
template <class CT>
__global__ void k_mg_coarsen(MgDev<CT> F, const double *__restrict__ fine_diag, int cnx, int cny, int cnz, long long csy, long long csz, double *__restrict__ cx,
                             double *__restrict__ cy, double *__restrict__ cz, double *__restrict__ diag,
                             double *__restrict__ inv, int32_t *__restrict__ mark, double *__restrict__ vx, double *__restrict__ vb,
                             double *__restrict__ vt, double wgal) {
    const long long cB = (long long)cnx * cny * cnz;
    const long long cc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cc > cB) return;
    if (cc == cB) { mark[cc] = 0; return; }
    // the level's vectors start a solve at zero (a previous solve may have left values on cells that are no rows now;
    // their pads are never written)
    vx[cc] = 0.0; vb[cc] = 0.0; vt[cc] = 0.0;
    const int I = (int)(cc % cnx);
    const long long tq = cc / cnx;
    const int J = (int)(tq % cny), K = (int)(tq / cny);
    double sx = 0, sy_ = 0, sz_ = 0, sd = 0, internal = 0;
    for (int dk = 0; dk < 2; ++dk)
        for (int dj = 0; dj < 2; ++dj)
            for (int di = 0; di < 2; ++di) {
                const int i = 2 * I + di, j = 2 * J + dj, k = 2 * K + dk;
                if (i >= F.nx || j >= F.ny || k >= F.nz) continue;
                const long long c = (long long)i + F.sy * j + F.sz * k;
                // a child that is not a row contributes nothing; in slab mode the ghost cells are such children and,
                // unlike every other non-row, still carry their x and y couplings
                if (fine_diag[c] == 0.0) continue;
                sd += fine_diag[c]; // the exact diagonal: a coarse cell is a row iff this balance is positive
                const double ax = F.scale * (double)F.cx[c], ay = F.scale * (double)F.cy[c], az = F.scale * (double)F.cz[c];
                if (di == 0) sx += ax; else internal += ax;   // low face of the coarse cell / face between two children
                if (dj == 0) sy_ += ay; else internal += ay;
                if (dk == 0) sz_ += az; else internal += az;
            }
    const double d = wgal * (sd - 2.0 * internal);
    const bool row = d > 0.0;
    cx[cc] = row ? wgal * sx : 0.0;
    cy[cc] = row ? wgal * sy_ : 0.0;
    cz[cc] = row ? wgal * sz_ : 0.0;
    diag[cc] = row ? d : 0.0;
    inv[cc] = row ? 1.0 / d : 0.0;
    mark[cc] = row ? 1 : 0;
}
// a coupling must vanish when the neighbour on its low side is not a row (its own arrays are zero there,
// but the coefficient is stored with the high cell)
__global__ void k_mg_prune(int cnx, int cny, int cnz, long long csy, long long csz, const double *__restrict__ inv,
                           double *__restrict__ cx, double *__restrict__ cy, double *__restrict__ cz) {
    const long long cB = (long long)cnx * cny * cnz;
    const long long cc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cc >= cB) return;
    const int I = (int)(cc % cnx);
    const long long tq = cc / cnx;
    const int J = (int)(tq % cny), K = (int)(tq / cny);
    if (I == 0 || inv[cc - 1] == 0.0) cx[cc] = 0.0;
    if (J == 0 || inv[cc - csy] == 0.0) cy[cc] = 0.0;
    if (K == 0 || inv[cc - csz] == 0.0) cz[cc] = 0.0;
}
// The row list of a level from the scan of its marks, the row count, and (where the level has them: cx_f != null) the
// float copies of its pruned operator.
__global__ void k_mg_compact(long long B, const int32_t *__restrict__ scan, int32_t *__restrict__ rows, int32_t *__restrict__ n_rows,
                             const double *__restrict__ cx, const double *__restrict__ cy, const double *__restrict__ cz,
                             const double *__restrict__ diag, const double *__restrict__ inv, float *__restrict__ cx_f,
                             float *__restrict__ cy_f, float *__restrict__ cz_f, float *__restrict__ diag_f, float *__restrict__ inv_f) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    if (c == 0) *n_rows = scan[B];
    if (scan[c + 1] > scan[c]) rows[scan[c]] = (int32_t)c;
    if (cx_f) {
        cx_f[c] = (float)cx[c]; cy_f[c] = (float)cy[c]; cz_f[c] = (float)cz[c];
        diag_f[c] = (float)diag[c]; inv_f[c] = (float)inv[c];
    }
}

__global__ void k_mg_narrow(long long B, const double *__restrict__ diag, const double *__restrict__ inv, float *__restrict__ diag_f,
                            float *__restrict__ inv_f) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    diag_f[c] = (float)diag[c];
    inv_f[c] = (float)inv[c];
}

// ---------------------------------------------------------------------------------------------

static hipError_t mg_alloc_padded(double **p, int64_t n, int64_t pad) {
    hipError_t e = hipMalloc((void **)p, (size_t)(n + 2 * pad) * sizeof(double));
    if (e != hipSuccess) return e;
    e = flip_zero_new(*p, (size_t)(n + 2 * pad) * sizeof(double));
    *p += pad;
    return e;
}

void flip_mg_free(flip_ctx *ctx) {
    if (ctx->mg_diag_f) { (void)hipFree(ctx->mg_diag_f - ctx->pad); ctx->mg_diag_f = ctx->mg_inv_f = nullptr; }
    for (flip_mg_level &L : ctx->mg) {
        double **dp[] = { &L.cx, &L.cy, &L.cz, &L.diag, &L.inv, &L.x };   // x, b, t are one allocation
        for (double **q : dp)
            if (*q) { (void)hipFree(*q - L.pad); *q = nullptr; }
        L.b = L.t = nullptr;
        float **fp[] = { &L.cx_f, &L.cy_f, &L.cz_f, &L.diag_f, &L.inv_f };
        for (float **q : fp)
            if (*q) { (void)hipFree(*q - L.pad); *q = nullptr; }
        if (L.rows) (void)hipFree(L.rows);
        if (L.scan) (void)hipFree(L.scan);
        if (L.n_rows) (void)hipFree(L.n_rows);
    }
    ctx->mg.clear();
}

// levels of at most this many cells use the one-thread-per-cell kernels (FLIP_MG_DENSE_CELLS: tuning aid, 0 = never)
static int64_t mg_dense_cells() {
    static const int64_t v = [] { const char *e = getenv("FLIP_MG_DENSE_CELLS"); return e ? (int64_t)atoll(e) : (int64_t)131072; }();
    return v;
}

static int mg_coarsest_pad(int nx, int ny) { return (int)(((int64_t)nx * ny + 1 + 15) / 16 * 16); }
// can one workgroup solve a level of this size in LDS?
static bool mg_coarsest_ok(int nx, int ny, int nz) {
    const int64_t B = (int64_t)nx * ny * nz;
    return B <= MG_COARSEST_CELLS && 2 * (B + 2 * mg_coarsest_pad(nx, ny)) * (int64_t)sizeof(double) <= MG_COARSEST_LDS;
}

// Allocates the level arrays once per context (sizes depend on the box only).  No level is created when the
// box itself is small enough for the coarsest solver: multigrid has nothing to offer there.
static int mg_ensure_levels(flip_ctx *ctx) {
    if (!ctx->mg.empty()) return 0;
    int nx = ctx->g.nx, ny = ctx->g.ny, nz = ctx->g.nz;
    for (;;) {
        if (mg_coarsest_ok(nx, ny, nz) || (nx <= 2 && ny <= 2 && nz <= 2)) break;
        nx = (nx + 1) / 2; ny = (ny + 1) / 2; nz = (nz + 1) / 2;
        flip_mg_level L;
        L.nx = nx; L.ny = ny; L.nz = nz;
        L.B = (int64_t)nx * ny * nz;
        L.pad = (((int64_t)nx * ny + FLIP_BLOCK + 63) / 64) * 64;
        hipError_t e = hipSuccess;
        double **dp[] = { &L.cx, &L.cy, &L.cz, &L.diag, &L.inv };
        for (double **q : dp)
            if (e == hipSuccess) e = mg_alloc_padded(q, L.B, L.pad);
        if (e == hipSuccess) {   // the level's three vectors in one allocation: one fill zeroes them before a solve
            const int64_t span = L.B + 2 * L.pad;
            e = hipMalloc((void **)&L.x, (size_t)(3 * span) * sizeof(double));
            if (e == hipSuccess) {
                e = flip_zero_new(L.x, (size_t)(3 * span) * sizeof(double));
                L.x += L.pad; L.b = L.x + span; L.t = L.b + span;
            }
        }
        if (L.B > mg_dense_cells() && !getenv("FLIP_MG_WIDE_LEVELS")) {
            float **fp[] = { &L.cx_f, &L.cy_f, &L.cz_f, &L.diag_f, &L.inv_f };
            for (float **q : fp)
                if (e == hipSuccess) {
                    e = hipMalloc((void **)q, (size_t)(L.B + 2 * L.pad) * sizeof(float));
                    if (e == hipSuccess) e = flip_zero_new(*q, (size_t)(L.B + 2 * L.pad) * sizeof(float));
                    if (e == hipSuccess) *q += L.pad;
                }
        }
        if (e == hipSuccess) e = hipMalloc((void **)&L.rows, (size_t)L.B * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&L.scan, (size_t)(L.B + 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&L.n_rows, 4);
        ctx->mg.push_back(L);
        if (e != hipSuccess) {
            flip_mg_free(ctx);
            return flip_fail_hip(ctx, e, "hipMalloc of the multigrid levels", __FILE__, __LINE__);
        }
    }
    return 0;
}

static MgDev<float> mg_level0(flip_ctx *ctx) {
    MgDev<float> L;
    // with an overlapping slab preconditioner: the rows of the widened region, and the weighted residual if weights are on
    const bool wide = ctx->slab_on && ctx->slab_overlap > 0;
    L.B = ctx->B; L.sy = ctx->g.nx; L.sz = (long long)ctx->g.nx * ctx->g.ny;
    L.nx = ctx->g.nx; L.ny = ctx->g.ny; L.nz = ctx->g.nz;
    // in slab mode the preconditioner is local to the rank: no coupling across the slab faces
    L.cx = ctx->cg_c[0]; L.cy = ctx->cg_c[1]; L.cz = ctx->slab_on ? ctx->mg_cz : ctx->cg_c[2];
    L.scale = ctx->sys_s2;
    { const MgCheb k = mg_cheb(); L.c1 = k.c1; L.a2 = k.a2; L.c2 = k.c2; }
    L.diag = ctx->mg_diag_f; L.inv = ctx->mg_inv_f;
    L.x = ctx->z; L.b = (wide && ctx->slab_weighted) ? ctx->dist_rw : ctx->r; L.t = ctx->ap;
    L.b_alt = ctx->r2; L.S = ctx->scal;
    L.reg = ctx->reg;   // z = M^-1 r; Ap is free between the update and direction kernels
    L.rows = wide ? ctx->mg_rows : ctx->cg_rows; L.n_rows = ctx->d_counts + (wide ? 7 : 6);
    return L;
}
static MgDev<double> mg_level(const flip_mg_level &S) {
    MgDev<double> L;
    L.B = S.B; L.sy = S.nx; L.sz = (long long)S.nx * S.ny;
    L.nx = S.nx; L.ny = S.ny; L.nz = S.nz;
    L.cx = S.cx; L.cy = S.cy; L.cz = S.cz;
    L.scale = 1.0;
    { const MgCheb k = mg_cheb(); L.c1 = k.c1; L.a2 = k.a2; L.c2 = k.c2; }
    L.diag = S.diag; L.inv = S.inv;
    L.x = S.x; L.b = S.b; L.t = S.t;
    L.b_alt = nullptr; L.S = nullptr;
    L.reg = nullptr;
    L.rows = S.rows; L.n_rows = S.n_rows;
    return L;
}

// The same level through the float copies of its operator (only where they exist).
static MgDev<float> mg_level_narrow(const flip_mg_level &S) {
    MgDev<float> L;
    L.B = S.B; L.sy = S.nx; L.sz = (long long)S.nx * S.ny;
    L.nx = S.nx; L.ny = S.ny; L.nz = S.nz;
    L.cx = S.cx_f; L.cy = S.cy_f; L.cz = S.cz_f;
    L.scale = 1.0;
    { const MgCheb k = mg_cheb(); L.c1 = k.c1; L.a2 = k.a2; L.c2 = k.c2; }
    L.diag = S.diag_f; L.inv = S.inv_f;
    L.x = S.x; L.b = S.b; L.t = S.t;
    L.b_alt = nullptr; L.S = nullptr;
    L.reg = nullptr;
    L.rows = S.rows; L.n_rows = S.n_rows;
    return L;
}

// Weight of the Galerkin product in the coarse operators: w P^T A P.  1/2 is the re-discretisation (see the header);
// 0.54, an operator 8 % stiffer and so a coarse correction 8 % smaller on every level, takes fewer CG iterations
// (256^3: 0.45: 57.1, 0.5: 50.9, 0.52: 49.9, 0.54: 49.6, 0.56: 50.0, 0.6: 53.6 per substep; 512^3: 55.3 -> 54.3; 64^3:
// 39.7 -> 39.4; 128^3: 44.8 -> 44.95).  Any positive weight leaves the V-cycle symmetric positive definite.
// FLIP_MG_WGAL overrides it for tuning.
static double mg_wgal() { static const double w = [] { const char *e = getenv("FLIP_MG_WGAL"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.54; }(); return w; }
template <class CT>
static int mg_coarsen_into(flip_ctx *ctx, const MgDev<CT> &F, const double *fine_diag, flip_mg_level &S) {
    hipStream_t st = ctx->stream;
    const long long csy = S.nx, csz = (long long)S.nx * S.ny;
    k_mg_coarsen<CT><<<flip_grid_blocks(S.B + 1), FLIP_BLOCK, 0, st>>>(F, fine_diag, S.nx, S.ny, S.nz, csy, csz, S.cx, S.cy, S.cz, S.diag,
                                                                      S.inv, S.scan, S.x, S.b, S.t, mg_wgal());
    FLIP_LAUNCH_CHECK();
    k_mg_prune<<<flip_grid_blocks(S.B), FLIP_BLOCK, 0, st>>>(S.nx, S.ny, S.nz, csy, csz, S.inv, S.cx, S.cy, S.cz);
    FLIP_LAUNCH_CHECK();
    if (int rc = flip_exclusive_scan(ctx, S.scan, S.B + 1)) return rc;
    k_mg_compact<<<flip_grid_blocks(S.B), FLIP_BLOCK, 0, st>>>(S.B, S.scan, S.rows, S.n_rows, S.cx, S.cy, S.cz, S.diag, S.inv, S.cx_f, S.cy_f,
                                                              S.cz_f, S.diag_f, S.inv_f);
    FLIP_LAUNCH_CHECK();
    return 0;
}

// Builds the coarse operators for the system currently assembled (called once per solve).
int flip_mg_setup(flip_ctx *ctx) {
    if (int rc = mg_ensure_levels(ctx)) return rc;
    if (ctx->mg.empty()) return 0; // box too small to coarsen: the caller keeps the Jacobi preconditioner
    {
        const flip_mg_level &S = ctx->mg.back();
        FLIP_REQUIRE(mg_coarsest_ok(S.nx, S.ny, S.nz), "multigrid: the coarsest level does not fit one workgroup");
    }
    // float copies of the level-0 diagonal and its inverse, padded like every array the stencil kernels read
    if (!ctx->mg_diag_f) {
        float *base = nullptr;
        FLIP_CHECK(hipMalloc((void **)&base, 2 * (size_t)(ctx->B + 2 * ctx->pad) * sizeof(float)));
        FLIP_CHECK(flip_zero_new(base, 2 * (size_t)(ctx->B + 2 * ctx->pad) * sizeof(float)));
        ctx->mg_diag_f = base + ctx->pad;
        ctx->mg_inv_f = base + (ctx->B + 2 * ctx->pad) + ctx->pad;
    }
    if (!ctx->mg_narrowed) {   // otherwise k_build_system has written them along with the doubles
        k_mg_narrow<<<flip_grid_blocks(ctx->B), FLIP_BLOCK, 0, ctx->stream>>>(ctx->B, ctx->cg_diag, ctx->cg_inv, ctx->mg_diag_f, ctx->mg_inv_f);
        FLIP_LAUNCH_CHECK();
    }
    if (int rc = mg_coarsen_into(ctx, mg_level0(ctx), ctx->cg_diag, ctx->mg[0])) return rc;
    for (size_t l = 1; l < ctx->mg.size(); ++l)
        if (int rc = mg_coarsen_into(ctx, mg_level(ctx->mg[l - 1]), ctx->mg[l - 1].diag, ctx->mg[l])) return rc;
    return 0;
}

static int mg_blocks(const flip_ctx *ctx, int64_t B) {
    int64_t nb = (B + FLIP_BLOCK - 1) / FLIP_BLOCK;
    if (nb > ctx->pcg_blocks) nb = ctx->pcg_blocks;
    return (int)(nb < 1 ? 1 : nb);
}

// One V-cycle: z = M^-1 r on level 0, and the partials of r.z in slot 1 for the next direction kernel.
int flip_mg_vcycle(flip_ctx *ctx, bool pre_smoothed) {
    hipStream_t st = ctx->stream;
    const int32_t *done = &ctx->scal->done;
    const int nblk = ctx->pcg_blocks;
    const size_t nl = ctx->mg.size();
    MgDev<float> L0 = mg_level0(ctx);
    // ---- down: pre-smooth (result in x), restrict the residual ----
    const bool mark = ctx->kprof_arm;
    if (!pre_smoothed) k_mg_pre<float, true><<<nblk, FLIP_BLOCK, 0, st>>>(L0, done, L0.x);
    if (mark) FLIP_CHECK(hipEventRecord(ctx->kprof[2], st));
    k_mg_restrict<float, true><<<mg_blocks(ctx, ctx->mg[0].B * 8), FLIP_BLOCK, 0, st>>>(L0, mg_level(ctx->mg[0]), done, L0.x);
    if (mark) FLIP_CHECK(hipEventRecord(ctx->kprof[3], st));
    const int64_t dense_max = mg_dense_cells();
    for (size_t l = 0; l + 1 < nl; ++l) {
        MgDev<double> F = mg_level(ctx->mg[l]);
        const int64_t B = ctx->mg[l].B, Bc = ctx->mg[l + 1].B;
        if (ctx->mg[l].cx_f && Bc * 8 > dense_max) {
            const MgDev<float> Fn = mg_level_narrow(ctx->mg[l]);
            k_mg_pre<float, false><<<mg_blocks(ctx, B), FLIP_BLOCK, 0, st>>>(Fn, done, Fn.x);
            k_mg_restrict<float, false><<<mg_blocks(ctx, Bc * 8), FLIP_BLOCK, 0, st>>>(Fn, mg_level(ctx->mg[l + 1]), done, Fn.x);
            continue;
        }
        if (B <= dense_max) k_mg_pre_dense<<<(unsigned)((B + FLIP_BLOCK - 1) / FLIP_BLOCK), FLIP_BLOCK, 0, st>>>(F, done, F.x);
        else k_mg_pre<double, false><<<mg_blocks(ctx, B), FLIP_BLOCK, 0, st>>>(F, done, F.x);
        if (Bc * 8 <= dense_max) k_mg_restrict_dense<<<(unsigned)((Bc * 8 + FLIP_BLOCK - 1) / FLIP_BLOCK), FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.x);
        else k_mg_restrict<double, false><<<mg_blocks(ctx, Bc * 8), FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.x);
    }
    // ---- coarsest ----
    {
        const flip_mg_level &S = ctx->mg[nl - 1];
        const int pad = mg_coarsest_pad(S.nx, S.ny);
        const size_t lds = 2 * (size_t)(S.B + 2 * pad) * sizeof(double);
        k_mg_coarsest<<<1, 1024, lds, st>>>(mg_level(S), done, pad);
    }
    // ---- up: correction + post-smooth 1 (x -> t), post-smooth 2 (t -> x) ----
    for (size_t l = nl - 1; l-- > 0;) {
        MgDev<double> F = mg_level(ctx->mg[l]);
        const int nb = mg_blocks(ctx, ctx->mg[l].B);
        if (ctx->mg[l].B <= dense_max) {
            const unsigned nd = (unsigned)((ctx->mg[l].B + FLIP_BLOCK - 1) / FLIP_BLOCK);
            k_mg_post_dense<<<nd, FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.x, F.t);
            k_mg_smooth_dense<<<nd, FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.t, F.x);
            continue;
        }
        if (ctx->mg[l].cx_f && ctx->mg[l + 1].B * 8 > dense_max) {
            const MgDev<float> Fn = mg_level_narrow(ctx->mg[l]);
            k_mg_post<float, false><<<nb, FLIP_BLOCK, 0, st>>>(Fn, mg_level(ctx->mg[l + 1]), done, Fn.x, Fn.t);
            k_mg_smooth<float, false, false><<<nb, FLIP_BLOCK, 0, st>>>(Fn, mg_level(ctx->mg[l + 1]), done, Fn.t, Fn.x, nullptr);
            continue;
        }
        k_mg_post<double, false><<<nb, FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.x, F.t);
        k_mg_smooth<double, false, false><<<nb, FLIP_BLOCK, 0, st>>>(F, mg_level(ctx->mg[l + 1]), done, F.t, F.x, nullptr);
    }
    if (mark) FLIP_CHECK(hipEventRecord(ctx->kprof[4], st));
    k_mg_post<float, true><<<nblk, FLIP_BLOCK, 0, st>>>(L0, mg_level(ctx->mg[0]), done, L0.x, L0.t);
    if (mark) FLIP_CHECK(hipEventRecord(ctx->kprof[5], st));
    if (ctx->slab_on && ctx->slab_overlap > 0) // r.z waits for the neighbours' share of z (flip_pcg_dist_combine)
        k_mg_smooth<float, false, true><<<nblk, FLIP_BLOCK, 0, st>>>(L0, mg_level(ctx->mg[0]), done, L0.t, L0.x, nullptr);
    else
        k_mg_smooth<float, true, true><<<nblk, FLIP_BLOCK, 0, st>>>(L0, mg_level(ctx->mg[0]), done, L0.t, L0.x, ctx->partial + 1 * FLIP_MAX_PCG_BLOCKS);
    if (mark) FLIP_CHECK(hipEventRecord(ctx->kprof[6], st));
    FLIP_LAUNCH_CHECK();
    return 0;
}
