// rand_rmw.hip -- what the HBM of an MI355X sustains for the access pattern of k_paths2's row updates: 32-byte
// read-modify-writes at random 32-byte-aligned places of a region far larger than the caches (here 64 GiB), every wave
// updating 51 places per step (one per lane), as many waves in flight as k_paths2 runs (5 per SIMD).
//   hipcc --offload-arch=gfx950 -O3 rand_rmw.hip -o rand_rmw && ./rand_rmw
// Variants: "dep"  -- a wave waits for its update before issuing the next (a row update of k_paths2 follows the reduction
//                     of the column's records: one update in flight per wave)
//           "indep"-- 4 updates in flight per wave (the rate the memory system takes when latency is hidden)
//           "read" -- the loads alone, "write" -- the stores alone
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int MODE, int INFLIGHT, int GRP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_rmw(double *buf, unsigned long long n_entries,
                                                                                         int steps, int lanes) {
    const unsigned long long wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (lane >= lanes) return;
    unsigned long long s = mix(wave * 64 + lane / GRP);
    for (int it = 0; it < steps; it += INFLIGHT) {
        double2 a[INFLIGHT], b[INFLIGHT];
        double2 *p[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) { s = mix(s); p[j] = reinterpret_cast<double2 *>(buf + ((s % (n_entries / GRP)) * GRP + lane % GRP) * 4); }
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) { a[j] = p[j][0]; b[j] = p[j][1]; }
        } else {
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) { a[j] = make_double2(1.0, 2.0); b[j] = a[j]; }
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) { a[j].x += 1.0; a[j].y += 0.5; b[j].x += 1.0; b[j].y += 0.25; }
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) { p[j][0] = a[j]; p[j][1] = b[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < INFLIGHT; j++) if (a[j].x == 12345.678 && b[j].y == 1.5) p[j][0] = a[j];   // keeps the loads alive
        }
    }
}

int main(int argc, char **argv) {
    const unsigned long long mib = argc > 1 ? strtoull(argv[1], 0, 10) : 65536;
    const unsigned long long n_entries = (mib << 20) / 32;
    double *buf;
    CK(hipMalloc(&buf, n_entries * 32));
    CK(hipMemset(buf, 0, n_entries * 32));
    const int waves = 256 * 4 * 5, blocks = waves / 4, steps = 2048, lanes = 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"rmw 32-B entries, 1 per 32-B block", "rmw 4 lanes per 128-B line", "rmw 8 lanes per 256 B", "rmw 2 lanes per 64 B"};
    for (int v = 0; v < 4; v++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (v == 0) k_rmw<0, 1, 1><<<blocks, 256>>>(buf, n_entries, steps, lanes);
            if (v == 1) k_rmw<0, 1, 4><<<blocks, 256>>>(buf, n_entries, steps, lanes);
            if (v == 2) k_rmw<0, 1, 8><<<blocks, 256>>>(buf, n_entries, steps, lanes);
            if (v == 3) k_rmw<0, 1, 2><<<blocks, 256>>>(buf, n_entries, steps, lanes);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)waves * lanes * steps;
            if (rep) printf("%-34s %6.1f ms  %.3e entries/s  (%.2f TB/s of 32-byte accesses%s)\n", names[v], ms, n / (ms * 1e-3),
                            n * 64.0 / (ms * 1e-3) / 1e12, ", read + write");
        }
    }
    return 0;
}
