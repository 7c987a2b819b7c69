// stream_append.hip -- the two phases of the plan of DESIGN.md 7.2c for the row updates of k_paths4, as micro-benchmarks:
// instead of a random 32-byte read-modify-write per (start, column, end), (1) APPEND a 40-byte record (end rank + the two
// (value, error) pairs) to one of P streams chosen by the end's range, (2) later take (start, range) tasks, hold the range's
// accumulators in LDS and add its stream up there.
//   hipcc --offload-arch=gfx950 -O3 stream_append.hip -o stream_append && ./stream_append [P] [ends]
// Phase 1 as measured here: 1280 workgroups of 4 waves (5 waves per SIMD like k_paths4), every step 22 lanes of a wave emit
// a record for a pseudo-random end; the workgroup shares P cursors in LDS (ds_add_rtn) and P streams in HBM.
// Phase 2: one workgroup of 1024 threads per (workgroup of phase 1, range): the stream is read back and added into LDS
// accumulators [ends / P][4] with the per-slot serialisation k_pair_tri uses for its double-double sums.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rec { double a, b, c, d; long long end; };     // 40 bytes

template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
void k_append(Rec *streams, unsigned *counts, long long cap, int n_ends, int steps, int lanes) {
    __shared__ unsigned cur[P];
    for (int t = threadIdx.x; t < P; t += 256) cur[t] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long s = mix(blockIdx.x * 256ull + threadIdx.x);
    Rec *base = streams + (size_t)blockIdx.x * P * cap;
    if (lane < lanes)
        for (int it = 0; it < steps; it++) {
            s = mix(s);
            const int u = (int)(s % (unsigned long long)n_ends);
            const int p = (int)((long long)u * P / n_ends);
            const unsigned pos = atomicAdd(&cur[p], 1u);
            if (pos < cap) {
                Rec r; r.a = 1.0; r.b = 0.5; r.c = 2.0; r.d = 0.25; r.end = u;
                base[(size_t)p * cap + pos] = r;
            }
        }
    __syncthreads();
    for (int t = threadIdx.x; t < P; t += 256) counts[blockIdx.x * P + t] = cur[t] < cap ? cur[t] : (unsigned)cap;
}

__device__ __forceinline__ void two_sum(double &hi, double &lo, double x) {
    const double s = hi + x, bb = s - hi;
    lo += (hi - (s - bb)) + (x - bb);
    hi = s;
}

template <int P>
__global__ __launch_bounds__(1024) void k_reduce(const Rec *streams, const unsigned *counts, long long cap, int n_ends, double *out) {
    extern __shared__ double acc[];                        // [slots][4]
    const int slots = (n_ends + P - 1) / P;
    __shared__ unsigned claim[2048];
    const int task = blockIdx.x;                           // (workgroup of phase 1, range)
    const int p = task % P;
    const int e0 = (int)((long long)p * n_ends / P);
    for (int t = threadIdx.x; t < slots * 4; t += 1024) acc[t] = 0.0;
    for (int t = threadIdx.x; t < 2048; t += 1024) claim[t] = 0xffffffffu;
    __syncthreads();
    const Rec *src = streams + (size_t)task * cap;
    const unsigned n = counts[task];
    for (unsigned t = threadIdx.x; t < ((n + 1023u) & ~1023u); t += 1024) {
        Rec r; r.a = r.b = r.c = r.d = 0.0; r.end = -1;
        if (t < n) r = src[t];
        bool pending = r.end >= 0;
        const int sl = pending ? (int)r.end - e0 : 0;
        volatile unsigned *vc = claim;
        while (__syncthreads_or(pending)) {                 // records of one slot take turns (rare: 1024 records, 4500 slots)
            if (pending) vc[sl & 2047] = threadIdx.x;
            __syncthreads();
            if (pending && vc[sl & 2047] == threadIdx.x) {
                volatile double *a = acc + (size_t)sl * 4;
                double h0 = a[0], l0 = a[1], h1 = a[2], l1 = a[3];
                two_sum(h0, l0, r.a); l0 += r.b;
                two_sum(h1, l1, r.c); l1 += r.d;
                a[0] = h0; a[1] = l0; a[2] = h1; a[3] = l1;
                pending = false;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    double s = 0.0;                                          // "finalise": here just a checksum so that nothing is optimised away
    for (int t = threadIdx.x; t < slots; t += 1024) s += acc[t * 4] + acc[t * 4 + 2];
    if (s == 123.456) out[task] = s;
}

template <int P>
int run(int n_ends) {
    const int groups = 1280, steps = 4096, lanes = 22;
    const long long per_stream = (long long)4 * lanes * steps / P;
    const long long cap = per_stream + per_stream / 4 + 256;
    Rec *streams; unsigned *counts; double *out;
    CK(hipMalloc(&streams, sizeof(Rec) * (size_t)groups * P * cap));
    CK(hipMalloc(&counts, sizeof(unsigned) * groups * P));
    CK(hipMalloc(&out, sizeof(double) * groups * P));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double n = (double)groups * 4 * lanes * steps;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        k_append<P><<<groups, 256>>>(streams, counts, cap, n_ends, steps, lanes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("P = %3d  append : %7.2f ms  %.3e records/s  (%.2f TB/s of 40-byte records, %.1f GB)\n", P, ms, n / (ms * 1e-3),
                        n * 40.0 / (ms * 1e-3) / 1e12, n * 40.0 / 1e9);
    }
    const int slots = (n_ends + P - 1) / P;
    const size_t lds = sizeof(double) * 4 * (size_t)slots;
    CK(hipFuncSetAttribute((const void *)k_reduce<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        k_reduce<P><<<groups * P, 1024, lds>>>(streams, counts, cap, n_ends, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("P = %3d  reduce : %7.2f ms  %.3e records/s  (%d accumulators = %zu KB of LDS per workgroup)\n", P, ms, n / (ms * 1e-3),
                        slots, lds >> 10);
    }
    CK(hipFree(streams)); CK(hipFree(counts)); CK(hipFree(out));
    return 0;
}

int main(int argc, char **argv) {
    const int n_ends = argc > 1 ? atoi(argv[1]) : 145772;
    if (run<32>(n_ends)) return 1;
    if (run<64>(n_ends)) return 1;
    if (run<128>(n_ends)) return 1;
    return 0;
}
