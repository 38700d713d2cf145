// Do two chains of dependent kernels, one per stream, overlap on this device?  Each chain mimics a multigrid CG
// iteration: a few bandwidth-bound sweeps followed by a run of tiny launch-latency-bound kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) k_big(const double *__restrict__ a, double *__restrict__ b, long long n) {
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long long)gridDim.x * 256) b[q] = a[q] * 1.0000001 + 1.0;
}
__global__ void __launch_bounds__(256) k_small(const double *__restrict__ a, double *__restrict__ b, int n) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < n) b[q] = a[q] + 1.0;
}
static void chain(hipStream_t st, double *a, double *b, long long n, int iters, int bigs, int smalls, int big_blocks) {
    for (int it = 0; it < iters; ++it) {
        for (int q = 0; q < bigs; ++q) { k_big<<<big_blocks, 256, 0, st>>>(a, b, n); double *t = a; a = b; b = t; }
        for (int q = 0; q < smalls; ++q) { k_small<<<42, 256, 0, st>>>(a, b, 10752); double *t = a; a = b; b = t; }
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const long long n = 5100744; // doubles per vector: 41 MB, a pair fits the last-level cache
    double *buf[4];
    for (double *&p : buf) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    hipStream_t s[2];
    for (hipStream_t &q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    const int iters = 8;
    for (int big_blocks : { 1024, 512 }) {
        for (int mode = 0; mode < 2; ++mode) { // 0: plain launches, 1: graphs
            hipGraphExec_t ex[2] = { nullptr, nullptr };
            if (mode == 1)
                for (int q = 0; q < 2; ++q) {
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(s[q], hipStreamCaptureModeThreadLocal));
                    chain(s[q], buf[2 * q], buf[2 * q + 1], n, iters, 5, 13, big_blocks);
                    CK(hipStreamEndCapture(s[q], &g));
                    CK(hipGraphInstantiate(&ex[q], g, nullptr, nullptr, 0));
                    CK(hipGraphDestroy(g));
                }
            for (int streams = 1; streams <= 2; ++streams) {
                double best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    for (int round = 0; round < 4; ++round)
                        for (int q = 0; q < 2; ++q) {
                            hipStream_t st = s[streams == 2 ? q : 0];
                            if (mode == 1) CK(hipGraphLaunch(ex[q], st));
                            else chain(st, buf[2 * q], buf[2 * q + 1], n, iters, 5, 13, big_blocks);
                        }
                    CK(hipDeviceSynchronize());
                    const double t = now() - t0;
                    best = t < best ? t : best;
                }
                printf("big kernels of %4d workgroups, %s, %d stream(s): %7.1f us per iteration of one chain (2 chains x 32 iterations)\n", big_blocks,
                       mode ? "graphs" : "plain launches", streams, best * 1e6 / 64);
            }
            for (hipGraphExec_t e : ex)
                if (e) CK(hipGraphExecDestroy(e));
        }
    }
    // the two parts on their own
    for (int part = 0; part < 2; ++part) {
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            chain(s[0], buf[0], buf[1], n, 32, part == 0 ? 5 : 0, part == 0 ? 0 : 13, 1024);
            CK(hipDeviceSynchronize());
            const double t = now() - t0;
            best = t < best ? t : best;
        }
        printf("%s alone: %7.1f us per iteration\n", part == 0 ? "5 big kernels" : "13 small kernels", best * 1e6 / 32);
    }
    // big chain on one stream, small chain on the other
    {
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            chain(s[0], buf[0], buf[1], n, 32, 5, 0, 1024);
            chain(s[1], buf[2], buf[3], n, 32, 0, 13, 1024);
            CK(hipDeviceSynchronize());
            const double t = now() - t0;
            best = t < best ? t : best;
        }
        printf("big chain and small chain on two streams: %7.1f us per iteration\n", best * 1e6 / 32);
    }
    return 0;
}
