// common.h -- shared device/host helpers of libxmap_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
// The kernels are written for gfx950 (MI355X): 160 KB of LDS per CU (k_cbs_scatter holds 147 KB, k_ts_bin 76 KB), wave64, its
// DPP / MFMA / LDS-DMA forms.  A device pass for another target is refused here rather than failing late in a kernel.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libxmap_hip is written for gfx950 (MI355X): build with --offload-arch=gfx950"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/xmap_hip.h"

namespace xmap {

void set_error(const char *fmt, ...);

#define XM_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            xmap::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return XMAP_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define XM_LAUNCH_CHECK() XM_HIP(hipGetLastError())

#define XM_ARG(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            xmap::set_error("%s:%d bad argument: %s", __FILE__, __LINE__, #cond); \
            return XMAP_ERR_ARG;                                            \
        }                                                                   \
    } while (0)

// Temporaries of one entry-point call: the library's own per-stream arenas (util.hip), not hipMallocAsync.
hipError_t xm_malloc_async(void **p, size_t bytes, hipStream_t st);
hipError_t xm_free_async(void *p, hipStream_t st);
// Every entry point that takes temporaries opens a scope first: whatever path the call leaves by (XM_HIP / XM_ARG /
// `if (rc) return rc`), the arena's count of live temporaries is back to what it was at entry, so an early return cannot
// leave the arena growing for the rest of the process.
struct XmScope {
    int dev; hipStream_t st; size_t live0; bool ok;
    explicit XmScope(hipStream_t st);
    ~XmScope();
    XmScope(const XmScope &) = delete;
    XmScope &operator=(const XmScope &) = delete;
};
#define XM_SCOPE(stream) xmap::XmScope xm_scope_((hipStream_t)(stream))
// plan.hip: stable LSD radix sort of (key, value) pairs by the low `bits` bits of the key; tmp buffers of n entries
int radix_sort_pairs(hipStream_t st, unsigned long long *keys, int *vals, unsigned long long *keys_tmp, int *vals_tmp, long long n,
                     int bits);

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// readlane with a wave-uniform lane index (SGPR)
__device__ __forceinline__ int rl32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlf(float v, int l) { return __int_as_float(rl32(__float_as_int(v), l)); }
__device__ __forceinline__ long long rl64(long long v, int l) {
    int lo = rl32((int)(v & 0xffffffffll), l), hi = rl32((int)(v >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double rld(double v, int l) {
    return __longlong_as_double(rl64(__double_as_longlong(v), l));
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Error-free accumulation (Knuth two-sum + renormalisation, double-double running sum): the result is
// the exact sum of the added fp64 values to ~2^-104, hence independent of the order of addition.
__device__ __forceinline__ void dd_add(double &hi, double &lo, double x) {
    double s = hi + x;
    double bb = s - hi;
    double e = (hi - (s - bb)) + (x - bb);
    e += lo;
    double h2 = s + e;
    lo = e - (h2 - s);
    hi = h2;
}

// murmur3 finaliser: partition hash of an item index
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

}  // namespace xmap
