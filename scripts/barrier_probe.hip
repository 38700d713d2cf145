// Cost of a grid-wide barrier built from one device-scope counter, for G workgroups of 1024 threads that are all
// resident: the price of folding several small multigrid kernels into one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) { ok = false; break; } // never spin for ever
        }
    }
    __syncthreads();
    return ok;
}
__global__ void __launch_bounds__(1024) k_barriers(unsigned *counter, int n, double *data, int len) {
    for (int b = 0; b < n; ++b) {
        // a little work between barriers: every thread touches a few elements written by others in the previous phase
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < len; q += gridDim.x * blockDim.x) data[q] = data[(q + 977) % len] + 1.0;
        if (!grid_barrier(counter, (unsigned)(b + 1) * gridDim.x)) return;
    }
}
int main() {
    unsigned *counter; double *data;
    const int len = 24000;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&data, len * 8)); CK(hipMemset(data, 0, len * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : { 8, 16, 32, 64 }) {
        for (int n : { 1, 9, 33 }) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4));
                CK(hipEventRecord(e0));
                k_barriers<<<G, 1024>>>(counter, n, data, len);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("G %3d workgroups, %2d phases (24000-element sweep + barrier): %7.2f us\n", G, n, best * 1e3);
        }
    }
    return 0;
}
