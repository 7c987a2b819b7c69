// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate with NACC independent accumulator chains per wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float x) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float a = x + threadIdx.x, b = x - threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    if (s == 12345.f) out[0] = s;
}
template <int NACC>
void run(int wgs_per_cu, float *d) {
    int iters = 20000 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g(256 * wgs_per_cu), b(256);
    k<NACC><<<g, b>>>(d, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<g, b>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)g.x * 4 * iters * 16 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, ms, flop / ms / 1e9);
}
int main() {
    float *d; hipMalloc(&d, 1024);
    run<1>(1, d); run<1>(2, d); run<1>(4, d); run<2>(1, d); run<2>(2, d); run<4>(1, d); run<4>(2, d);
    return 0;
}
