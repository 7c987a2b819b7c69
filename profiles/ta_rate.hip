// ta_rate.hip -- what a CU's vector-memory path (TA / TCP) takes per instruction, for the load shapes of k_paths4's column
// loop.  All addresses fall into a small L2-resident region, so neither HBM nor the L2 capacity is in the way: the rate is
// that of the address / tag / return path of the CU, shared by all its waves.  16 waves per CU (4 per SIMD), every wave
// issues a chain-free stream of loads (results consumed once at the end).
//   hipcc --offload-arch=gfx950 -O3 ta_rate.hip -o ta_rate && ./ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16; return z; }

// MODE 0: dwordx4, 64 lanes, every lane its own random 64-byte record (the merged records today: 4 such per column)
// MODE 1: dwordx4, 11 lanes random records, 53 lanes the SAME address (dummy lanes of today's record loads)
// MODE 2: dwordx4, 11 lanes active (exec-masked), random records
// MODE 3: dwordx4, 44 lanes = 11 records x 4 consecutive 16-byte pieces (cooperative load of the same bytes), 20 lanes one address
// MODE 4: dword,   64 lanes random
// MODE 5: dwordx4, 64 lanes consecutive (1 KiB contiguous)
// MODE 6: dwordx4, 44 lanes pieces, 20 lanes masked off
// MODE 7: dwordx2, 64 lanes random
template <int MODE>
__global__ __launch_bounds__(256) void k_ta(const uint4 *buf, unsigned n16, int iters, unsigned long long *sink) {
    const int lane = threadIdx.x & 63;
    unsigned s = mix(blockIdx.x * 256 + threadIdx.x + 1);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = mix(s + u);
            const unsigned wave_r = __builtin_amdgcn_readfirstlane(s);
            unsigned idx;
            bool act = true;
            if (MODE == 0 || MODE == 4 || MODE == 7) idx = (s % (n16 / 4)) * 4;
            else if (MODE == 1) idx = lane < 11 ? (s % (n16 / 4)) * 4 : (wave_r % (n16 / 4)) * 4;
            else if (MODE == 2) { idx = (s % (n16 / 4)) * 4; act = lane < 11; }
            else if (MODE == 3 || MODE == 6) {
                const unsigned rs = mix(wave_r + (lane >> 2));       // the quad's record
                idx = lane < 44 ? (rs % (n16 / 4)) * 4 + (lane & 3) : (wave_r % (n16 / 4)) * 4;
                if (MODE == 6) act = lane < 44;
            } else idx = (wave_r % (n16 / 64)) * 64 + lane;
            if (MODE == 4) { if (act) acc.x += reinterpret_cast<const unsigned *>(buf)[idx * 4]; }
            else if (MODE == 7) { if (act) { const uint2 v = reinterpret_cast<const uint2 *>(buf)[idx * 2]; acc.x += v.x; acc.y += v.y; } }
            else if (act) { const uint4 v = buf[idx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = acc.x;
}

int main() {
    const unsigned n16 = (8u << 20) / 16;            // 8 MiB region: 1 MiB per XCD-L2-slice worth, L2 / MALL resident
    uint4 *buf; unsigned long long *sink;
    CK(hipMalloc(&buf, (size_t)n16 * 16)); CK(hipMemset(buf, 1, (size_t)n16 * 16)); CK(hipMalloc(&sink, 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 4, iters = 2000;      // 4 blocks x 4 waves = 16 waves per CU
    const double ghz = prop.clockRate * 1e-6;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[8] = {"dwordx4 64 lanes random records", "dwordx4 11 random + 53 lanes one address", "dwordx4 11 lanes active (masked)",
                            "dwordx4 44 lanes pieces (11 x 64 B) + 20 one address", "dword 64 lanes random", "dwordx4 64 lanes contiguous",
                            "dwordx4 44 lanes pieces, 20 masked off", "dwordx2 64 lanes random"};
    for (int m = 0; m < 8; m++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            switch (m) {
                case 0: k_ta<0><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 1: k_ta<1><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 2: k_ta<2><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 3: k_ta<3><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 4: k_ta<4><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 5: k_ta<5><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 6: k_ta<6><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 7: k_ta<7><<<blocks, 256>>>(buf, n16, iters, sink); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double instr_per_cu = 16.0 * iters * 8;
        printf("%-56s %7.2f ms  %6.1f cycles per wave-instruction and CU (%.2f GHz)\n", names[m], ms, ms * 1e-3 * ghz * 1e9 / instr_per_cu, ghz);
    }
    return 0;
}
