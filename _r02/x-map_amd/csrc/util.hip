// util.hip -- error plumbing and the prefix sums every count->fill pair needs.
#include <stdarg.h>
#include <stdlib.h>
#include <vector>

#include "common.h"

namespace xmap {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- temporaries of the library ----------------------------------------------------------------------------------------
// Scratch that lives for one entry-point call (scan tile sums, sort histograms, planning arrays, the column end table).
// Round 1 took it from the stream-ordered allocator (hipMallocAsync / hipFreeAsync).  Under the ROCm 7.2 runtime of this
// image that gave wrong results from the second pass of a process on (a block handed out again while kernels of the
// same stream still read it; not with the 7.0 runtime PyTorch bundles -- INTEGRATION.md, "HIP runtime versions"), so
// the library keeps its own arenas instead: one per (thread, device, stream), bump allocation inside a call, everything
// recycled when the last temporary of the call is released.  Reuse is ordered by the stream itself: a later call on the
// same stream runs after the kernels of the earlier one.
namespace {
struct Chunk { char *base; size_t size, used; };
struct Arena { int dev; hipStream_t st; std::vector<Chunk> chunks; size_t live; };
thread_local std::vector<Arena> g_arenas;
Arena *arena_of(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (Arena &a : g_arenas) if (a.dev == dev && a.st == st) return &a;
    g_arenas.push_back(Arena{dev, st, {}, 0});
    return &g_arenas.back();
}
}  // namespace

hipError_t xm_malloc_async(void **p, size_t bytes, hipStream_t st) {
    Arena *a = arena_of(st);
    if (!a) return hipErrorInvalidDevice;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (Chunk &c : a->chunks)
        if (c.size - c.used >= bytes) { *p = c.base + c.used; c.used += bytes; a->live++; return hipSuccess; }
    Chunk c;
    c.size = bytes > ((size_t)32 << 20) ? bytes : ((size_t)32 << 20);
    c.used = bytes;
    hipError_t e = hipMalloc((void **)&c.base, c.size);
    if (e != hipSuccess) return e;
    a->chunks.push_back(c);
    *p = c.base;
    a->live++;
    return hipSuccess;
}
hipError_t xm_free_async(void *p, hipStream_t st) {
    (void)p;
    Arena *a = arena_of(st);
    if (!a || a->live == 0) return hipErrorInvalidValue;
    if (--a->live == 0) for (Chunk &c : a->chunks) c.used = 0;
    return hipSuccess;
}

// ---- three-kernel exclusive scan: tile sums -> scan of tile sums (one block) -> tile scans ----
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ long long block_exclusive_scan(long long v, long long *total, long long *smem) {
    // inclusive scan inside the wave, then across the 4 waves
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    long long base = 0, tot = 0;
    for (int k = 0; k < SCAN_THREADS / 64; k++) {
        long long s = smem[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <typename TIn>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_sums(const TIn *in, long long n, long long *tile_sum) {
    __shared__ long long smem[4];
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) s += (long long)in[base + k];
    long long tot;
    block_exclusive_scan(s, &tot, smem);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(long long *tile_sum, long long n_tiles, long long *grand) {
    __shared__ long long smem[4];
    long long carry = 0;
    for (long long b = 0; b < n_tiles; b += SCAN_THREADS) {
        long long i = b + threadIdx.x;
        long long v = i < n_tiles ? tile_sum[i] : 0;
        long long tot;
        long long ex = block_exclusive_scan(v, &tot, smem);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand = carry;
}

template <typename TIn>
__global__ __launch_bounds__(SCAN_THREADS) void k_tile_scan(const TIn *in, long long n, const long long *tile_off,
                                                            long long *out) {
    __shared__ long long smem[4];
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    long long v[SCAN_ITEMS];
    long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? (long long)in[base + k] : 0;
        s += v[k];
    }
    long long tot;
    long long ex = block_exclusive_scan(s, &tot, smem) + tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

template <typename TIn>
static int exclusive_scan(hipStream_t st, const TIn *in, int64_t *out, int64_t n, int64_t *h_total) {
    XM_ARG(n >= 0);
    // out[n] doubles as the grand total; tile offsets live in a small temporary
    int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    long long *tile = nullptr;
    XM_HIP(xm_malloc_async((void **)&tile, sizeof(long long) * (size_t)(n_tiles + 1), st));
    if (n_tiles > 0) {
        k_tile_sums<TIn><<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, st>>>(in, n, tile);
        XM_LAUNCH_CHECK();
    }
    k_scan_tile_sums<<<dim3(1), dim3(SCAN_THREADS), 0, st>>>(tile, n_tiles, (long long *)(out + n));
    XM_LAUNCH_CHECK();
    if (n_tiles > 0) {
        k_tile_scan<TIn><<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, st>>>(in, n, tile, (long long *)out);
        XM_LAUNCH_CHECK();
    }
    XM_HIP(xm_free_async(tile, st));
    if (h_total) {
        XM_HIP(hipMemcpyAsync(h_total, out + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
    }
    return XMAP_OK;
}
}  // namespace xmap

extern "C" {
const char *xmap_last_error(void) { return xmap::g_err; }
int xmap_version(void) { return 100; }

int xmap_exclusive_scan_i64(void *stream, const int64_t *in, int64_t *out, int64_t n, int64_t *h_total) {
    return xmap::exclusive_scan<long long>((hipStream_t)stream, (const long long *)in, out, n, h_total);
}
int xmap_exclusive_scan_i32_to_i64(void *stream, const int32_t *in, int64_t *out, int64_t n, int64_t *h_total) {
    return xmap::exclusive_scan<int>((hipStream_t)stream, in, out, n, h_total);
}
}
