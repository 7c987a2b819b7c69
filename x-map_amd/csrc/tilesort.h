// tilesort.h -- transposition of sparse records by key when every key's count is known in advance.
//
// Stage A transposes twice: the ratings by item (the rater records of the pair kernel) and the kept pairs by their
// heavier item (the mirrored half of item2item_simRDD; reference core/baselinerSim.py:182-183 emits both directions).
// Round 1 / 2 placed every record with a returning atomic on its key's cursor and a scattered write: both run at the
// memory system's rate of random 64-byte transactions (~1e10 / s), 4x the bytes they carry.  Here the final position
// space is known before a record moves (prefix sum of the per-key counts), so the records are routed towards it in two
// binning levels whose writes are runs of records, and placed exactly in LDS:
//
//   measure of key k   m[k] = ptr[k] + k * KW        (its first position + a weight per key, so that a tile bounds both
//                                                      the records and the keys it holds)
//   tile of key k      m[k] >> ts_log                  (T tiles, a level-A bucket = 2^NB_LOG consecutive tiles)
//   large key          count >= tile measure: always the last key of its tile, at most one per tile; its records are
//                      routed straight to its own position range at level B and converted in slices (k_*_large)
//   level A / B        a workgroup bins a chunk of CH records: the records are loaded whole (coalesced) into registers,
//                      bucket histogram and ranks by LDS atomics, one returning global atomic per occupied bucket for the
//                      fragment's place in the bucket's range, the records staged in LDS in bucket order and copied out
//                      (consecutive lanes write consecutive records of a fragment)
//   level C            one workgroup per tile: the small keys' records get their rank from LDS cursors, are laid out in
//                      final order in LDS and leave as whole rows
//
// Any order inside a key is a correct result (every consumer sums exactly or re-sorts), which is what makes ranks by
// atomics admissible.  No host synchronisation: chunk and slice lists are built on the device, grids are upper bounds.
#pragma once
#include "common.h"

namespace xmap {
namespace ts {

#ifndef EXP_BT
#define EXP_BT 256
#endif
constexpr int BT = EXP_BT;       // threads of a binning workgroup
// records per workgroup of a binning level: the chunk is staged in LDS in bucket order (16-byte records: 64 KB, 24-byte
// records: 48 KB, 32-byte records: 64 KB; two workgroups per CU either way)
#ifndef EXP_CH3       // (tuning builds)
#define EXP_CH3 2048
#endif
template <int RW> struct Chunk { static constexpr int CH = RW == 2 ? 4096 : (RW == 3 ? EXP_CH3 : 2048); static constexpr int EPT = CH / BT; };
constexpr int NB_LOG = 7;
constexpr int NB = 1 << NB_LOG;   // fine tiles per level-A bucket
constexpr int NA_MAX = 256;       // level-A buckets (LDS histogram; two staged chunks of 16-byte records per CU need the rest)
constexpr int CAP = 3072;         // records of a tile's small part that are laid out in LDS (more: placed directly)
constexpr int CT = 512;           // threads of a level-C workgroup
constexpr int NK_MAX = 520;       // keys of a tile's small part (the host chooses KW so that tile measure / KW + 1 fits)
constexpr int SL = 2048;          // records per slice of a large key
constexpr int LT = 256;           // threads of a slice workgroup

struct Geo {
    int K;                        // keys
    long long M;                  // records (= ptr[K])
    int KW, ts_log, T, NA;
    int ch;                       // records per chunk of the binning levels (Chunk<RW>::CH)
    const long long *ptr;         // [K + 1] first position of every key
    unsigned *tk;                 // [K] tile | large << 31
    int *tile_key0;               // [T + 1] first key of a tile
    long long *tile_pos0;         // [T + 1] = ptr[tile_key0]
    int *tile_large;              // [T] the tile's large key, or -1
    unsigned long long *curA;     // [NA] fragment cursors of the level-A buckets
    unsigned long long *curB;     // [2 T] ... of the tiles' small parts, then of their large keys
    unsigned *counters;           // [0] level-B chunks listed, [1] slices of large keys listed
    int2 *clist;                  // level-B chunks (bucket, chunk)
    int2 *slist;                  // slices (large key, slice)
    long long clist_cap, slist_cap;
};

__device__ __forceinline__ long long measure(const Geo &G, int k, long long p) { return p + (long long)k * G.KW; }

// one thread per key: tile, tile boundaries, large keys and their slices
__global__ __launch_bounds__(256) void k_ts_plan(Geo G) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G.K) return;
    const long long p = G.ptr[k], cnt = G.ptr[k + 1] - p;
    const int t = (int)(measure(G, k, p) >> G.ts_log);
    const bool large = cnt >= (1ll << G.ts_log);
    G.tk[k] = (unsigned)t | (large ? 0x80000000u : 0u);
    const int tp = k > 0 ? (int)(measure(G, k - 1, G.ptr[k - 1]) >> G.ts_log) : -1;
    for (int x = tp + 1; x <= t; x++) { G.tile_key0[x] = k; G.tile_pos0[x] = p; }
    if (k == G.K - 1)
        for (int x = t + 1; x <= G.T; x++) { G.tile_key0[x] = G.K; G.tile_pos0[x] = G.ptr[G.K]; }
    if (large) {
        G.tile_large[t] = k;
        const int nsl = (int)((cnt + SL - 1) / SL);
        const unsigned base = atomicAdd(&G.counters[1], (unsigned)nsl);
        for (int x = 0; x < nsl; x++)
            if ((long long)base + x < G.slist_cap) G.slist[base + x] = make_int2(k, x);
    }
}

// one thread per level-A bucket: its chunks of CH records for level B
__global__ __launch_bounds__(256) void k_ts_chunks(Geo G) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= G.NA) return;
    const int t0 = a << NB_LOG, t1 = min(G.T, (a + 1) << NB_LOG);
    const long long size = G.tile_pos0[t1] - G.tile_pos0[t0];
    const int nch = (int)((size + G.ch - 1) / G.ch);
    if (nch == 0) return;
    const unsigned base = atomicAdd(&G.counters[0], (unsigned)nch);
    for (int x = 0; x < nch; x++)
        if ((long long)base + x < G.clist_cap) G.clist[base + x] = make_int2(a, x);
}

// exclusive scan of cnt[0 .. n) (n <= 2 * BT) into off[0 .. n], all BT threads call
__device__ __forceinline__ void block_scan_2(const unsigned *cnt, unsigned *off, int n, unsigned *wsum) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned a = (2 * tid < n) ? cnt[2 * tid] : 0u, b = (2 * tid + 1 < n) ? cnt[2 * tid + 1] : 0u;
    unsigned inc = a + b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int x = 0; x < w; x++) base += wsum[x];
    const unsigned ex = base + inc - (a + b);
    if (2 * tid < n) off[2 * tid] = ex;
    if (2 * tid + 1 < n) off[2 * tid + 1] = ex + a;
    if (tid == BT - 1) off[n] = base + inc;
    __syncthreads();
}

// records of RW 64-bit words, key = low 32 bits of word 0
template <int RW>
struct RecLoader {
    const unsigned long long *__restrict__ in;
    // level A: chunk c = records [c CH, (c + 1) CH)
    __device__ __forceinline__ bool chunk(long long n_in, long long &i0, long long &i1) const {
        i0 = (long long)blockIdx.x * Chunk<RW>::CH;
        i1 = min(n_in, i0 + Chunk<RW>::CH);
        return true;
    }
    __device__ __forceinline__ void load(long long idx, unsigned long long (&w)[RW]) const {
#pragma unroll
        for (int x = 0; x < RW; x++) w[x] = in[idx * RW + x];
    }
    // a record the level does not route (it still reaches extra): none here
    __device__ __forceinline__ bool keep(const unsigned long long (&)[RW]) const { return true; }
    // what else the workgroup does with its chunk (w[r] = record i0 + r BT + tid, on[r]: there is one)
    __device__ __forceinline__ void extra(long long, const unsigned long long (&)[Chunk<RW>::EPT][RW], const bool (&)[Chunk<RW>::EPT]) const {}
};

// One binning level.  LEVEL_B = false: a chunk of the loader's index space (L.chunk), buckets = level-A buckets.
// LEVEL_B = true: a listed chunk of a level-A bucket's range in the input (position space), buckets = the bucket's fine
// tiles (small part) + their large keys.
template <int RW, bool LEVEL_B, typename Loader>
__global__ __launch_bounds__(BT) void k_ts_bin(Geo G, Loader L, long long n_in, unsigned long long *__restrict__ out) {
    constexpr int CH = Chunk<RW>::CH, EPT = Chunk<RW>::EPT;
    constexpr int NBK = LEVEL_B ? 2 * NB : NA_MAX;
    __shared__ unsigned hist[NBK];
    __shared__ unsigned off[NBK + 1];
    __shared__ long long gbase[NBK];
    __shared__ unsigned wsum[BT / 64];
    __shared__ unsigned long long stage[CH * RW];
    __shared__ unsigned short sbk[CH];
    const int tid = threadIdx.x;
    long long i0, i1;
    int a = 0;
    if (LEVEL_B) {
        if (blockIdx.x >= G.counters[0]) return;
        const int2 c = G.clist[blockIdx.x];
        a = c.x;
        const int t0 = a << NB_LOG, t1 = min(G.T, (a + 1) << NB_LOG);
        i0 = G.tile_pos0[t0] + (long long)c.y * CH;
        i1 = min(G.tile_pos0[t1], i0 + CH);
    } else {
        if (!L.chunk(n_in, i0, i1)) return;
    }
    const int nbk = LEVEL_B ? 2 * NB : G.NA;
    for (int b = tid; b < nbk; b += BT) hist[b] = 0u;
    __syncthreads();
    unsigned long long w[EPT][RW];
    bool on[EPT], rt[EPT];  // there is a record; it is routed
    unsigned br[EPT];       // tile word, then bucket << 16 | rank
#pragma unroll
    for (int r = 0; r < EPT; r++) {
        const long long idx = i0 + r * BT + tid;
        on[r] = idx < i1;
        if (on[r]) L.load(idx, w[r]);
        else {
#pragma unroll
            for (int x = 0; x < RW; x++) w[r][x] = 0ull;
        }
        rt[r] = on[r] && L.keep(w[r]);
    }
#pragma unroll
    for (int r = 0; r < EPT; r++) br[r] = rt[r] ? G.tk[(unsigned)w[r][0]] : 0u;
#pragma unroll
    for (int r = 0; r < EPT; r++) {
        if (!rt[r]) continue;
        const unsigned tkv = br[r];
        const int t = (int)(tkv & 0x7fffffffu);
        const int b = LEVEL_B ? ((t & (NB - 1)) | ((tkv >> 31) ? NB : 0)) : (t >> NB_LOG);
        br[r] = ((unsigned)b << 16) | atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    block_scan_2(hist, off, nbk, wsum);
    for (int b = tid; b < nbk; b += BT) {
        const unsigned c = hist[b];
        if (!c) continue;
        long long start;
        unsigned long long *cur;
        if (LEVEL_B) {
            const int t = (a << NB_LOG) | (b & (NB - 1));
            if (b >= NB) { start = G.ptr[G.tile_large[t]]; cur = &G.curB[(size_t)G.T + t]; }
            else { start = G.tile_pos0[t]; cur = &G.curB[t]; }
        } else {
            start = G.tile_pos0[b << NB_LOG];
            cur = &G.curA[b];
        }
        gbase[b] = start + (long long)atomicAdd(cur, (unsigned long long)c);
    }
#pragma unroll
    for (int r = 0; r < EPT; r++) {
        if (!rt[r]) continue;
        const unsigned b = br[r] >> 16, s = off[b] + (br[r] & 0xffffu);
#pragma unroll
        for (int x = 0; x < RW; x++) stage[s * RW + x] = w[r][x];
        sbk[s] = (unsigned short)b;
    }
    L.extra(i0, w, on);
    __syncthreads();
    const int n = (int)off[nbk];
    for (int s = tid; s < n; s += BT) {
        const unsigned b = sbk[s];
        unsigned long long *o = out + (size_t)(gbase[b] + (long long)(s - off[b])) * RW;
#pragma unroll
        for (int x = 0; x < RW; x++) o[x] = stage[s * RW + x];
    }
}

// what a level-C workgroup knows about its tile
struct TileHead { int k0, nk, large; long long pos0; int n; };
__device__ __forceinline__ TileHead tile_head(const Geo &G, int t) {
    TileHead h;
    h.k0 = G.tile_key0[t];
    const int k1 = G.tile_key0[t + 1];
    h.large = G.tile_large[t];
    const int ks1 = h.large >= 0 ? h.large : k1;
    h.nk = ks1 - h.k0;
    h.pos0 = G.tile_pos0[t];
    const long long pos1 = h.large >= 0 ? G.ptr[h.large] : G.tile_pos0[t + 1];
    h.n = (int)(pos1 - h.pos0);
    return h;
}

}  // namespace ts
}  // namespace xmap
