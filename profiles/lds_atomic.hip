// lds_atomic.hip -- LDS fp64 atomic add (with return) at random slots: the flush of a register partial into an LDS table
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// MODE 0: 4 atomics per update (rtn on hi, no-rtn on lo, for two sums); MODE 1: lock-free plain RMW of 32 B (no atomicity; upper bound of a locked RMW)
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int steps, int n_slots) {
    extern __shared__ double tab[];   // [n_slots][4]
    for (int i = threadIdx.x; i < n_slots * 4; i += 1024) tab[i] = 0.0;
    __syncthreads();
    unsigned long long s = mix(blockIdx.x * 1024ull + threadIdx.x);
    double chk = 0.0;
    for (int it = 0; it < steps; it++) {
        s = mix(s);
        const int slot = (int)(s % (unsigned)n_slots);
        const double v = (double)(s & 1023) * 0.001, c = 0.5;
        double *a = tab + slot * 4;
        if (MODE == 0) {
            const double o = atomicAdd(a, v);
            const double sum = o + v, bb = sum - o;
            const double err = (o - (sum - bb)) + (v - bb);
            atomicAdd(a + 1, err);
            const double o2 = atomicAdd(a + 2, c);
            const double sum2 = o2 + c, bb2 = sum2 - o2;
            const double err2 = (o2 - (sum2 - bb2)) + (c - bb2);
            atomicAdd(a + 3, err2);
        } else {
            double h = a[0], l = a[1], h2 = a[2], l2 = a[3];
            double sum = h + v, bb = sum - h; double e = (h - (sum - bb)) + (v - bb); e += l; double hh = sum + e; l = e - (hh - sum); h = hh;
            double sum2 = h2 + c, bb2 = sum2 - h2; double e2 = (h2 - (sum2 - bb2)) + (c - bb2); e2 += l2; double hh2 = sum2 + e2; l2 = e2 - (hh2 - sum2); h2 = hh2;
            a[0] = h; a[1] = l; a[2] = h2; a[3] = l2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_slots * 4; i += 1024) chk += tab[i];
    if (chk == 12345.678) out[0] = chk;
}
int main() {
    double *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 4096, blocks = 256 * 4;
    for (int n_slots : {512, 2048, 4096}) {
        const size_t lds = (size_t)n_slots * 32;
        CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int mode = 0; mode < 2; mode++)
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) k<0><<<blocks, 1024, lds>>>(out, steps, n_slots); else k<1><<<blocks, 1024, lds>>>(out, steps, n_slots);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double n = (double)blocks * 1024 * steps;
                if (rep) printf("slots %5d mode %d (%s): %7.2f ms  %.3e updates/s\n", n_slots, mode, mode ? "plain 32-B RMW" : "4 atomics + 2 two-sums", ms, n / (ms * 1e-3));
            }
    }
    return 0;
}
