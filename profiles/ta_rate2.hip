// ta_rate2.hip -- the load / store GROUPS of one column of k_paths4 as the CU's memory path sees them (companion of
// ta_rate.hip: there the cost turned out to be ~5 cycles per distinct 128-byte line of an instruction, whatever its width).
// Same set-up: addresses inside an 8 MiB region, 16 waves per CU, no dependent chains.
//   G0: merged records today    -- 4 x dwordx4 at +0/+16/+32/+48 of 11 random 64-byte records (53 lanes: one dummy address)
//   G1: merged records, pieces  -- 1 x dwordx4, 44 lanes = 11 records x 4 pieces
//   G2: row entries today       -- 2 x dwordx4 at +16/+0 of 22 random 32-byte entries, both slice lanes load (44 lanes)
//   G3: row entries, split      -- 1 x dwordx4: the two slice lanes of an end load one half of its entry each
//   G4: row stores today        -- 2 x dwordx4 stores by 22 lanes (+ 42 lanes to one junk line)
//   G5: row stores, split       -- 1 x dwordx4 store by 44 lanes (halves), 20 lanes junk
//   G6: end records today       -- dwordx4 + dwordx3 of 22 consecutive 32-byte records (clamped: 42 lanes repeat the last)
//   G7: end records, pieces     -- 1 x dwordx4, 44 lanes consecutive 16-byte pieces
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned mix(unsigned z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16; return z; }

template <int G>
__global__ __launch_bounds__(256) void k_grp(uint4 *buf, unsigned n16, int iters, unsigned long long *sink) {
    const int lane = threadIdx.x & 63;
    unsigned s = mix(blockIdx.x * 256 + (threadIdx.x & ~63) + 1);
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 *junk = buf + n16 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s = mix(s + u);
            const unsigned w = __builtin_amdgcn_readfirstlane(s);
            if (G == 0) {
                const unsigned r = (lane < 11 ? mix(w + lane) : w) % (n16 / 4) * 4;
#pragma unroll
                for (int q = 0; q < 4; q++) { const uint4 v = buf[r + q]; acc.x += v.x; acc.y += v.w; }
            } else if (G == 1) {
                const unsigned r = (lane < 44 ? mix(w + (lane >> 2)) % (n16 / 4) * 4 + (lane & 3) : w % (n16 / 4) * 4);
                const uint4 v = buf[r]; acc.x += v.x; acc.y += v.w;
            } else if (G == 2) {
                const unsigned r = (lane < 44 ? mix(w + (lane >> 1)) : w) % (n16 / 2) * 2;
                const uint4 v1 = buf[r + 1]; const uint4 v0 = buf[r]; acc.x += v0.x + v1.x; acc.y += v0.w + v1.w;
            } else if (G == 3) {
                const unsigned r = (lane < 44 ? mix(w + (lane >> 1)) % (n16 / 2) * 2 + (lane & 1) : w % (n16 / 2) * 2);
                const uint4 v = buf[r]; acc.x += v.x; acc.y += v.w;
            } else if (G == 4) {
                uint4 *p = (lane < 44 && !(lane & 1)) ? buf + mix(w + (lane >> 1)) % (n16 / 2) * 2 : junk;
                p[0] = make_uint4(s, 1, 2, 3); p[1] = make_uint4(s, 4, 5, 6);
            } else if (G == 5) {
                uint4 *p = lane < 44 ? buf + mix(w + (lane >> 1)) % (n16 / 2) * 2 + (lane & 1) : junk;
                p[0] = make_uint4(s, 1, 2, 3);
            } else if (G == 6) {
                const unsigned r = (w % (n16 / 128)) * 128 + (lane < 22 ? lane : 21) * 2;
                const uint4 v0 = buf[r]; const uint4 v1 = buf[r + 1]; acc.x += v0.x + v1.x; acc.y += v0.w + v1.z;
            } else {
                const unsigned r = (w % (n16 / 128)) * 128 + (lane < 44 ? lane : 43);
                const uint4 v = buf[r]; acc.x += v.x; acc.y += v.w;
            }
        }
    }
    if (acc.x + acc.y == 0x12345678u) sink[0] = acc.x;
}

int main() {
    const unsigned n16 = (8u << 20) / 16;
    uint4 *buf; unsigned long long *sink;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 4, iters = 2000;
    CK(hipMalloc(&buf, (size_t)n16 * 16 + (size_t)blocks * 4 * 128)); CK(hipMemset(buf, 1, (size_t)n16 * 16)); CK(hipMalloc(&sink, 8));
    const double ghz = prop.clockRate * 1e-6;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[8] = {"merged records today (4 x dwordx4, 11 records)", "merged records as pieces (1 x dwordx4, 44 lanes)",
                            "row entries today (2 x dwordx4, 22 entries x 2 lanes)", "row entries split over the slice lanes (1 x dwordx4)",
                            "row stores today (2 x dwordx4, 22 lanes)", "row stores split (1 x dwordx4, 44 lanes)",
                            "end records today (dwordx4 + dwordx4, 22 consecutive)", "end records as pieces (1 x dwordx4, 44 lanes)"};
    for (int m = 0; m < 8; m++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            switch (m) {
                case 0: k_grp<0><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 1: k_grp<1><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 2: k_grp<2><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 3: k_grp<3><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 4: k_grp<4><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 5: k_grp<5><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 6: k_grp<6><<<blocks, 256>>>(buf, n16, iters, sink); break;
                case 7: k_grp<7><<<blocks, 256>>>(buf, n16, iters, sink); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-58s %7.2f ms  %6.1f cycles per group and CU\n", names[m], ms, ms * 1e-3 * ghz * 1e9 / (16.0 * iters * 4));
    }
    return 0;
}
