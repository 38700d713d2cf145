// Tuning aid: times variants of the fused direction + SpMV kernel on a synthetic liquid block, to find what bounds
// the row-list stencil kernels of the solver.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o stencil_probe stencil_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define BLK 256
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    long long n, sy, sz;
    int nblk;
    double beta, s2;
    const int32_t *rows;
    const double *z, *pold, *diag;
    const float *cx, *cy, *cz;
    double *pnew, *ap, *partial;
};
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ void row_range(const Args &A, long long &lo, long long &hi) {
    const int per = A.nblk >> 3;
    const int rank = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    lo = (A.n * rank) / A.nblk;
    hi = (A.n * (rank + 1)) / A.nblk;
}
#define BODY(c)                                                                                                            \
    {                                                                                                                      \
        const double z0 = z[c], zxm = z[c - 1], zxp = z[c + 1], zym = z[c - sy], zyp = z[c + sy], zzm = z[c - sz],         \
                     zzp = z[c + sz];                                                                                      \
        const double p0 = pold[c], pxm = pold[c - 1], pxp = pold[c + 1], pym = pold[c - sy], pyp = pold[c + sy],           \
                     pzm = pold[c - sz], pzp = pold[c + sz];                                                               \
        const float axm = cx[c], axp = cx[c + 1], aym = cy[c], ayp = cy[c + sy], azm = cz[c], azp = cz[c + sz];            \
        const double d = dg[c];                                                                                            \
        const double pc = z0 + beta * p0;                                                                                  \
        double acc = (double)axm * (zxm + beta * pxm);                                                                     \
        acc += (double)axp * (zxp + beta * pxp);                                                                           \
        acc += (double)aym * (zym + beta * pym);                                                                           \
        acc += (double)ayp * (zyp + beta * pyp);                                                                           \
        acc += (double)azm * (zzm + beta * pzm);                                                                           \
        acc += (double)azp * (zzp + beta * pzp);                                                                           \
        const double s = d * pc - A.s2 * acc;                                                                              \
        A.pnew[c] = pc;                                                                                                    \
        A.ap[c] = s;                                                                                                       \
        pap += pc * s;                                                                                                     \
    }
#define PROLOGUE                                                                                                           \
    __shared__ double scratch[4];                                                                                          \
    const double *__restrict__ z = A.z, *__restrict__ pold = A.pold, *__restrict__ dg = A.diag;                            \
    const float *__restrict__ cx = A.cx, *__restrict__ cy = A.cy, *__restrict__ cz = A.cz;                                 \
    const long long sy = A.sy, sz = A.sz;                                                                                  \
    const double beta = A.beta;                                                                                            \
    long long lo, hi;                                                                                                      \
    row_range(A, lo, hi);                                                                                                  \
    double pap = 0.0;
#define EPILOGUE                                                                                                           \
    pap = block_sum(pap, scratch);                                                                                         \
    if (threadIdx.x == 0) A.partial[blockIdx.x] = pap;

// K0: the production form
__global__ void __launch_bounds__(BLK) k0(Args A) {
    PROLOGUE
#pragma unroll 2
    for (long long q = lo + threadIdx.x; q < hi; q += BLK) {
        const long long c = A.rows[q];
        BODY(c)
    }
    EPILOGUE
}
// K1: row index of the next trip loaded before the current one is processed
__global__ void __launch_bounds__(BLK) k1(Args A) {
    PROLOGUE
    long long q = lo + threadIdx.x;
    int cn = q < hi ? A.rows[q] : 0;
    for (; q < hi; q += BLK) {
        const long long c = cn;
        if (q + BLK < hi) cn = A.rows[q + BLK];
        BODY(c)
    }
    EPILOGUE
}
// K2: SpMV only (7 loads of one vector): what the gathers of the second vector cost
__global__ void __launch_bounds__(BLK) k2(Args A) {
    PROLOGUE
    (void)pold; (void)beta;
#pragma unroll 2
    for (long long q = lo + threadIdx.x; q < hi; q += BLK) {
        const long long c = A.rows[q];
        const double z0 = z[c], zxm = z[c - 1], zxp = z[c + 1], zym = z[c - sy], zyp = z[c + sy], zzm = z[c - sz], zzp = z[c + sz];
        const float axm = cx[c], axp = cx[c + 1], aym = cy[c], ayp = cy[c + sy], azm = cz[c], azp = cz[c + sz];
        double acc = (double)axm * zxm;
        acc += (double)axp * zxp; acc += (double)aym * zym; acc += (double)ayp * zyp; acc += (double)azm * zzm; acc += (double)azp * zzp;
        const double s = dg[c] * z0 - A.s2 * acc;
        A.pnew[c] = z0;
        A.ap[c] = s;
        pap += z0 * s;
    }
    EPILOGUE
}
// K3: 32-bit cell ids and strides
__global__ void __launch_bounds__(BLK) k3(Args A) {
    __shared__ double scratch[4];
    const double *__restrict__ z = A.z, *__restrict__ pold = A.pold, *__restrict__ dg = A.diag;
    const float *__restrict__ cx = A.cx, *__restrict__ cy = A.cy, *__restrict__ cz = A.cz;
    const int sy = (int)A.sy, sz = (int)A.sz;
    const double beta = A.beta;
    long long lo, hi;
    row_range(A, lo, hi);
    double pap = 0.0;
    const int n = (int)(hi - lo);
    const int32_t *__restrict__ rows = A.rows + lo;
#pragma unroll 2
    for (int q = threadIdx.x; q < n; q += BLK) {
        const int c = rows[q];
        BODY(c)
    }
    EPILOGUE
}
// K4: streaming floor: own-cell loads only (z, p, diag, 3 couplings), same stores
__global__ void __launch_bounds__(BLK) k4(Args A) {
    PROLOGUE
#pragma unroll 2
    for (long long q = lo + threadIdx.x; q < hi; q += BLK) {
        const long long c = A.rows[q];
        const double pc = z[c] + beta * pold[c];
        const double s = dg[c] * pc - A.s2 * ((double)cx[c] + (double)cy[c] + (double)cz[c]);
        A.pnew[c] = pc;
        A.ap[c] = s;
        pap += pc * s;
    }
    (void)sy; (void)sz;
    EPILOGUE
}
// K5: x neighbours from the neighbouring lanes when the list is contiguous there
__global__ void __launch_bounds__(BLK) k5(Args A) {
    PROLOGUE
#pragma unroll 2
    for (long long q0 = lo; q0 < hi; q0 += BLK) {
        const long long q = q0 + threadIdx.x;
        const bool live = q < hi;
        const int c = live ? A.rows[q] : -2;
        double pc = 0.0;
        if (live) pc = z[c] + beta * pold[c];
        const int cl = __shfl_up(c, 1, 64), cr = __shfl_down(c, 1, 64);
        double pl = __shfl_up(pc, 1, 64), pr = __shfl_down(pc, 1, 64);
        const int lane = threadIdx.x & 63;
        if (live) {
            if (lane == 0 || cl != c - 1) pl = z[c - 1] + beta * pold[c - 1];
            if (lane == 63 || cr != c + 1) pr = z[c + 1] + beta * pold[c + 1];
            const double zym = z[c - sy], zyp = z[c + sy], zzm = z[c - sz], zzp = z[c + sz];
            const double pym = pold[c - sy], pyp = pold[c + sy], pzm = pold[c - sz], pzp = pold[c + sz];
            const float axm = cx[c], axp = cx[c + 1], aym = cy[c], ayp = cy[c + sy], azm = cz[c], azp = cz[c + sz];
            double acc = (double)axm * pl;
            acc += (double)axp * pr;
            acc += (double)aym * (zym + beta * pym);
            acc += (double)ayp * (zyp + beta * pyp);
            acc += (double)azm * (zzm + beta * pzm);
            acc += (double)azp * (zzp + beta * pzp);
            const double s = dg[c] * pc - A.s2 * acc;
            A.pnew[c] = pc;
            A.ap[c] = s;
            pap += pc * s;
        }
    }
    EPILOGUE
}

// K6: k4's work as a one-shot grid (one row per thread, no loop, no reduction): cost of the persistent form
__global__ void __launch_bounds__(BLK) k6(Args A) {
    const long long q = (long long)blockIdx.x * BLK + threadIdx.x;
    if (q >= A.n) return;
    const long long c = A.rows[q];
    const double pc = A.z[c] + A.beta * A.pold[c];
    const double s = A.diag[c] * pc - A.s2 * ((double)A.cx[c] + (double)A.cy[c] + (double)A.cz[c]);
    A.pnew[c] = pc;
    A.ap[c] = s;
}
// K7: the production stencil as a one-shot grid, without the reduction
__global__ void __launch_bounds__(BLK) k7(Args A) {
    const long long q = (long long)blockIdx.x * BLK + threadIdx.x;
    if (q >= A.n) return;
    const double *__restrict__ z = A.z, *__restrict__ pold = A.pold, *__restrict__ dg = A.diag;
    const float *__restrict__ cx = A.cx, *__restrict__ cy = A.cy, *__restrict__ cz = A.cz;
    const long long sy = A.sy, sz = A.sz;
    const double beta = A.beta;
    double pap = 0.0;
    const long long c = A.rows[q];
    BODY(c)
    (void)pap;
}
// K8: one-shot, two rows per thread
__global__ void __launch_bounds__(BLK) k8(Args A) {
    const long long q0 = ((long long)blockIdx.x * BLK) * 2 + threadIdx.x;
    const double *__restrict__ z = A.z, *__restrict__ pold = A.pold, *__restrict__ dg = A.diag;
    const float *__restrict__ cx = A.cx, *__restrict__ cy = A.cy, *__restrict__ cz = A.cz;
    const long long sy = A.sy, sz = A.sz;
    const double beta = A.beta;
    double pap = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long long q = q0 + u * BLK;
        if (q < A.n) {
            const long long c = A.rows[q];
            BODY(c)
        }
    }
    (void)pap;
}

// K9: the production form when a per-row flag says that all six couplings have the common value (interior rows):
// the three coupling arrays are not read for such rows
__global__ void __launch_bounds__(BLK) k9(Args A, const uint8_t *__restrict__ reg, float S) {
    PROLOGUE
#pragma unroll 2
    for (long long q = lo + threadIdx.x; q < hi; q += BLK) {
        const long long c = A.rows[q];
        const double z0 = z[c], zxm = z[c - 1], zxp = z[c + 1], zym = z[c - sy], zyp = z[c + sy], zzm = z[c - sz], zzp = z[c + sz];
        const double p0 = pold[c], pxm = pold[c - 1], pxp = pold[c + 1], pym = pold[c - sy], pyp = pold[c + sy], pzm = pold[c - sz],
                     pzp = pold[c + sz];
        float axm = S, axp = S, aym = S, ayp = S, azm = S, azp = S;
        if (!reg[c]) { axm = cx[c]; axp = cx[c + 1]; aym = cy[c]; ayp = cy[c + sy]; azm = cz[c]; azp = cz[c + sz]; }
        const double d = dg[c];
        const double pc = z0 + beta * p0;
        double acc = (double)axm * (zxm + beta * pxm);
        acc += (double)axp * (zxp + beta * pxp);
        acc += (double)aym * (zym + beta * pym);
        acc += (double)ayp * (zyp + beta * pyp);
        acc += (double)azm * (zzm + beta * pzm);
        acc += (double)azp * (zzp + beta * pzp);
        const double s = d * pc - A.s2 * acc;
        A.pnew[c] = pc;
        A.ap[c] = s;
        pap += pc * s;
    }
    EPILOGUE
}

template <class K>
static void run(const char *name, K kern, Args A, int nblk, double bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 5; ++w) kern<<<nblk, BLK>>>(A);
    CK(hipDeviceSynchronize());
    const int reps = 200;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) kern<<<nblk, BLK>>>(A);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-28s nblk %5d  %8.2f us  %6.2f TB/s (53 B/row)\n", name, nblk, us, bytes / us * 1e-6);
}

int main(int argc, char **argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;   // 1: 256^3-like, 2: 512^3-like
    const int nx = 160 * scale, ny = 200 * scale, nz = 160 * scale;
    const int lx = 128 * scale, ly = 94 * scale, lz = 128 * scale;
    const long long B = (long long)nx * ny * nz, sy = nx, sz = (long long)nx * ny, pad = sz + 256;
    std::vector<int32_t> rows;
    for (int k = 1; k <= lz; ++k) for (int j = 1; j <= ly; ++j) for (int i = 1; i <= lx; ++i) rows.push_back((int32_t)(i + sy * j + sz * k));
    const long long n = (long long)rows.size();
    printf("box %d x %d x %d = %lld cells, %lld rows\n", nx, ny, nz, B, n);
    auto dalloc = [&](size_t bytes) { void *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
    double *z = (double *)dalloc((B + 2 * pad) * 8) + pad, *pold = (double *)dalloc((B + 2 * pad) * 8) + pad;
    double *pnew = (double *)dalloc((B + 2 * pad) * 8) + pad, *ap = (double *)dalloc((B + 2 * pad) * 8) + pad;
    double *dg = (double *)dalloc((B + 2 * pad) * 8) + pad;
    float *cx = (float *)dalloc((B + 2 * pad) * 4) + pad, *cy = (float *)dalloc((B + 2 * pad) * 4) + pad, *cz = (float *)dalloc((B + 2 * pad) * 4) + pad;
    int32_t *drows = (int32_t *)dalloc(n * 4);
    CK(hipMemcpy(drows, rows.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<double> hz(B);
    for (long long c = 0; c < B; ++c) hz[c] = (double)((c * 2654435761u) % 1000) * 1e-3;
    CK(hipMemcpy(z, hz.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pold, hz.data(), B * 8, hipMemcpyHostToDevice));
    double *partial = (double *)dalloc(4096 * 8);
    Args A{ n, sy, sz, 1024, 0.37, 1.0, drows, z, pold, dg, cx, cy, cz, pnew, ap, partial };
    const double bytes = 53.0 * (double)n;
    for (int nblk : { 1024, 1280, 2048 }) {
        A.nblk = nblk;
        run("k0 production", k0, A, nblk, bytes);
        run("k1 index prefetch", k1, A, nblk, bytes);
        run("k2 spmv only (7 loads)", k2, A, nblk, bytes);
        run("k3 32-bit ids", k3, A, nblk, bytes);
        run("k4 own cell only", k4, A, nblk, bytes);
        run("k5 x neighbours by shuffle", k5, A, nblk, bytes);
    }
    {
        uint8_t *reg = (uint8_t *)dalloc(B);
        CK(hipMemset(reg, 1, B));
        A.nblk = 1024;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 5; ++w) k9<<<1024, BLK>>>(A, reg, 0.5f);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 200; ++r) k9<<<1024, BLK>>>(A, reg, 0.5f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k9 couplings by flag (all regular)  %8.2f us  %6.2f TB/s (53 B/row)\n", ms * 1e3 / 200, bytes / (ms * 1e3 / 200) * 1e-6);
    }
    {
        const int g1 = (int)((n + BLK - 1) / BLK);
        run("k6 own cell, one-shot grid", k6, A, g1, bytes);
        run("k7 stencil, one-shot grid", k7, A, g1, bytes);
        run("k8 stencil, one-shot x2", k8, A, (g1 + 1) / 2, bytes);
    }
    // the production form on two streams at once, each with its own vectors: does a pair of such kernels overlap?
    {
        double *z2 = (double *)dalloc((B + 2 * pad) * 8) + pad, *pold2 = (double *)dalloc((B + 2 * pad) * 8) + pad;
        double *pnew2 = (double *)dalloc((B + 2 * pad) * 8) + pad, *ap2 = (double *)dalloc((B + 2 * pad) * 8) + pad;
        double *partial2 = (double *)dalloc(4096 * 8);
        CK(hipMemcpy(z2, hz.data(), B * 8, hipMemcpyHostToDevice));
        Args A2 = A;
        A2.z = z2; A2.pold = pold2; A2.pnew = pnew2; A2.ap = ap2; A2.partial = partial2;
        hipStream_t s[2];
        for (int q = 0; q < 2; ++q) CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
        for (int nblk : { 512, 1024 }) {
            A.nblk = A2.nblk = nblk;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 200;
            for (int two = 0; two < 2; ++two) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r) {
                    k0<<<nblk, BLK, 0, s[0]>>>(A);
                    if (two) k0<<<nblk, BLK, 0, s[1]>>>(A2);
                }
                CK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / reps;
                printf("k0 nblk %d on %s: %.2f us per launch%s\n", nblk, two ? "two streams" : "one stream", us, two ? " pair" : "");
            }
        }
    }
    // a chain of full-size stencil kernels on one stream and a chain of tiny ones (a coarse multigrid level: 3000 rows,
    // 12 workgroups) on another: do the tiny ones hide under the large ones?
    {
        hipStream_t s[2];
        for (int q = 0; q < 2; ++q) CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
        Args big = A; big.nblk = 1024;
        Args tiny = A; tiny.n = 3000; tiny.nblk = 16;
        tiny.pnew = (double *)dalloc((B + 2 * pad) * 8) + pad; tiny.ap = (double *)dalloc((B + 2 * pad) * 8) + pad;
        tiny.partial = (double *)dalloc(4096 * 8);
        const int reps = 100;
        auto timeit = [&](bool run_big, bool run_tiny, int tiny_per_big) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) {
                if (run_big) k0<<<1024, BLK, 0, s[0]>>>(big);
                if (run_tiny) for (int q = 0; q < tiny_per_big; ++q) k0<<<16, BLK, 0, s[1]>>>(tiny);
            }
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / reps;
        };
        for (int tpb : { 1, 4 }) {
            timeit(true, true, tpb);
            const double a = timeit(true, false, tpb), b = timeit(false, true, tpb), c = timeit(true, true, tpb);
            printf("per round: 1 large kernel %.1f us; %d tiny kernels %.1f us; both on two streams %.1f us\n", a, tpb, b, c);
        }
    }
    return 0;
}
