// Do kernels of two streams run at the same time on this device?  Small (1 workgroup) and large (1024 workgroups)
// busy kernels, launched directly and as graphs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[2];
    for (int q = 0; q < 2; ++q) CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
    int *sink; CK(hipMalloc(&sink, 4));
    const long long cyc = 100000; // wall_clock64 ticks at 100 MHz: 1 ms
    spin<<<1, 64, 0, s[0]>>>(1000, sink); CK(hipDeviceSynchronize());
    for (int grid : { 1, 256, 1024, 4096 }) {
        double t0 = now();
        spin<<<grid, 256, 0, s[0]>>>(cyc, sink); CK(hipStreamSynchronize(s[0]));
        const double one = now() - t0;
        t0 = now();
        spin<<<grid, 256, 0, s[0]>>>(cyc, sink);
        spin<<<grid, 256, 0, s[1]>>>(cyc, sink);
        CK(hipDeviceSynchronize());
        const double two = now() - t0;
        // a chain of short kernels on each stream: 50 x (big 20 us + small 5 us)
        printf("grid %5d: one kernel %.3f ms, one per stream on two streams %.3f ms\n", grid, one * 1e3, two * 1e3);
    }
    // chains: alternating large/small kernels, as a V-cycle does
    for (int mode = 0; mode < 2; ++mode) {
        auto chain = [&](hipStream_t st) {
            for (int k = 0; k < 100; ++k) {
                spin<<<1024, 256, 0, st>>>(2000, nullptr);  // 20 us, fills the device
                spin<<<8, 256, 0, st>>>(1000, nullptr);     // 10 us, 8 workgroups
            }
        };
        hipGraphExec_t ex[2] = { nullptr, nullptr };
        if (mode == 1)
            for (int q = 0; q < 2; ++q) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(s[q], hipStreamCaptureModeThreadLocal));
                chain(s[q]);
                CK(hipStreamEndCapture(s[q], &g));
                CK(hipGraphInstantiate(&ex[q], g, nullptr, nullptr, 0));
            }
        auto go = [&](int q) { if (mode == 1) CK(hipGraphLaunch(ex[q], s[q])); else chain(s[q]); };
        go(0); CK(hipDeviceSynchronize());
        double t0 = now();
        go(0); CK(hipDeviceSynchronize());
        const double one = now() - t0;
        t0 = now();
        go(0); go(1); CK(hipDeviceSynchronize());
        const double two = now() - t0;
        printf("%s chain of 100 x (20 us full-device + 10 us 8-workgroup): one stream %.3f ms, two streams %.3f ms (ideal 3.0 / ~4.0)\n",
               mode ? "graph " : "direct", one * 1e3, two * 1e3);
    }
    return 0;
}
