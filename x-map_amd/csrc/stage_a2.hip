// stage_a2.hip -- stage A, second formulation: every unordered item pair is computed ONCE, in the row of
// its lighter item, and mirrored into the CSR afterwards (baseliner_calculate_sim_pipeline, reference
// utils/assist.py:66-77; core/baselinerSim.py:176-216).  sim, mutu and n_ij are symmetric in the reference
// bit for bit, and both directions are emitted by produce_pairwise_items (:182-183), so
// computing the pair once and writing it twice is the same result.
//
// "Weight" of an item = (number of raters, index); the per-user private copy of each profile is sorted
// heaviest first, so the partners a rater contributes to item i are exactly the PREFIX of its sorted
// profile in front of i: no filtering in the inner loop, half the reads, and the heavier an item is the
// fewer partners its row has (the heaviest rows, which dominate the first formulation's cost, become tiny).
//
//   k_hist / k_threshold / k_mark_heavy : rater-count histogram -> #items at least as heavy (bound on a row's
//                                         distinct partners) and the set H of at most HMAX items with more
//                                         than CH raters (dense ids)
//   k_sort_profiles : per-user sort by weight (4 short profiles per wave)
//   k_rater_records : the rater records of every item; W+_i = number of contributions of row i (k_plan2)
//   k_plan2 / k_fill_units2 : light units (item, hash partition) listed by LDS table class, heavy units (item in H,
//                     chunk of CH raters)
//   k_pair_tri      : light rows, one launch per table class (on forked streams); LDS hash table sized to the row's
//                     partner bound and shared by 1 / 2 / 4 / 16 waves, rater records instead of a dependent row_ptr
//                     hop, prefix-only profile reads; appends kept pairs to a half-COO.  LS = true is the
//                     RecommenderSim variant (core/recommenderSim.py:65-133): no filter, self pairs, second walk for
//                     the leave-one-out local sensitivity
//   k_pair_heavy    : rows of H, raters in chunks (one rater per lane), DENSE LDS table over H, partial tables to HBM
//   k_heavy_merge   : double-double merge of the chunk partials (4 waves per row), finalise, append
//   k_scatter       : mirror the half-COO into the CSR rows (atomic cursors)
// Sums are exact (double-double, or integer-exact in cosine mode), so neither the order of raters nor the
// chunking changes a bit of the result.
#include <type_traits>

#include "common.h"
#include "item_stats.h"
#include "tilesort.h"

namespace xmap {

constexpr int T_LOG_SLOTS = 10;
constexpr int T_SLOTS = 1 << T_LOG_SLOTS;
constexpr int HMAX = 1024;              // |H| <= HMAX: dense LDS table of the heavy kernel
constexpr uint32_t T_EMPTY = 0xFFFFFFFFu;
constexpr int SMALL_BOUND = 96;          // rows with at most this many partners use the 128-slot table
constexpr int MID_BOUND = 384;           // ... at most this many: the 512-slot table

constexpr int N_CLASSES = 5;
constexpr int WIDE_MIN = 2048;           // light rows with at least this many raters: 16 waves on one 1024-slot table
// table class (1 = 128 slots, 3 = 256, 2 = 512, 0 = 1024, 4 = 1024 "wide") -> position in the class-major unit list
__host__ __device__ __forceinline__ int class_rank(int cls) {
    return cls == 4 ? 0 : (cls == 0 ? 1 : (cls == 2 ? 2 : (cls == 3 ? 3 : 4)));
}

__device__ __forceinline__ unsigned long long wkey(int n, int item) {
    return ((unsigned long long)(unsigned)n << 32) | (unsigned)item;
}

// ---------------------------------------------------------------------------------------------
// most items have a handful of raters: the low bins are counted per workgroup in LDS first
constexpr int HIST_LDS = 2048;
constexpr int HIST_PER = 16;   // items per thread
__global__ __launch_bounds__(256) void k_hist(int I, const long long *iptr, int HB, int *hist) {
    __shared__ int loc[HIST_LDS];
    for (int t = threadIdx.x; t < HIST_LDS; t += 256) loc[t] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * 256 * HIST_PER;
    for (int q = 0; q < HIST_PER; q++) {
        const long long i = base + (long long)q * 256 + threadIdx.x;
        if (i < I) {
            long long n = iptr[i + 1] - iptr[i];
            int bin = n < HB - 1 ? (int)n : HB - 1;
            if (bin < HIST_LDS) atomicAdd(&loc[bin], 1); else atomicAdd(&hist[bin], 1);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < HIST_LDS && t < HB; t += 256)
        if (loc[t]) atomicAdd(&hist[t], loc[t]);
}

// pre[v] = #{items with n < v}.  CH = smallest v >= ch_min with #{n > v} <= HMAX.
__global__ __launch_bounds__(256) void k_threshold(int I, int HB, const long long *pre, int ch_min, int *CH) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= HB - 1 || v < ch_min) return;
    long long gt_v = I - pre[v + 1];
    long long gt_prev = (v == ch_min) ? (long long)HMAX + 1 : I - pre[v];
    if (gt_v <= HMAX && (v == ch_min || gt_prev > HMAX)) atomicMin(CH, v);
}

__global__ __launch_bounds__(256) void k_mark_heavy(int I, const long long *iptr, const int *CH, int *hid, int *hlist,
                                                    int *n_heavy) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    long long n = iptr[i + 1] - iptr[i];
    int h = -1;
    if (n > *CH) {
        h = atomicAdd(n_heavy, 1);
        if (h < HMAX) hlist[h] = i;
    }
    hid[i] = h;
}

struct RaterRec { int e0; int pos_ge; float rating; int user; };   // 16 B: one rater of an item
// fp64 ratings (the RecommenderSim variant, LS: AlterEgo ratings are np.float64 means, core/generator.py:123-138 ->
// core/recommenderSim.py:64-133): 16-byte profile entries and rater records of their own
struct UbWide { int item_ge; int pad; double rating; };      // 16 B: one entry of a sorted profile, fp64 rating
struct RaterRecWide { int e0; int pos_ge; double rating; };  // 16 B: one rater of an item, fp64 rating (no user: its average is 0)

// private copy of every profile sorted heaviest first, (index | flag, rating) interleaved.  One wave per 4 users:
// profiles of up to 16 ratings (90 % at BASELINE configs[1]) are sorted four at a time, one per 16-lane group, by a
// bitonic network cut off at the longest of the four (the xor-shuffles never leave a group); the others follow one
// by one on the whole wave (network cut off at the profile's length), profiles above 64 ratings by counting ranks.
__device__ __forceinline__ void sort_entry(const int *uitem, const float *urating, const long long *iptr,
                                           const double *info, long long e, unsigned long long &key, int &px, int &py) {
    const int it = uitem[e];
    const float r = urating[e];
    const double2 an = *(const double2 *)(info + (size_t)it * 4), nn = *(const double2 *)(info + (size_t)it * 4 + 2);
    key = wkey((int)nn.y, it);                                      // info[it] = (avg, norm, adjnorm, n): one 32-B record
    const unsigned ge = ((double)r >= an.x) ? 0x80000000u : 0u;   // rating >= item average
    px = (int)((unsigned)it | ge);
    py = __float_as_int(r);
}

// descending bitonic network over groups of `width` lanes (width a power of two <= 64, uniform); pos = lane in group
__device__ __forceinline__ void bitonic_desc(int width, int pos, unsigned long long &key, int &px, int &py) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
        if (k2 <= width)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long ok = __shfl_xor(key, j, 64);
            const int ox = __shfl_xor(px, j, 64), oy = __shfl_xor(py, j, 64);
            const bool desc = (pos & k2) == 0;
            const bool lower = (pos & j) == 0;
            const bool take_other = (lower == desc) ? (ok > key) : (ok < key);
            if (take_other) { key = ok; px = ox; py = oy; }
        }
    }
}

constexpr int SORT_LDS = 256;    // keys of a long profile staged in LDS (2 KB per wave: 8 KB per block leaves the kernel its full occupancy; 1024 had held it to 5 waves per SIMD for the sake of the few profiles of 257..1024 ratings, which now rank from the global scratch)

__device__ __forceinline__ int pow2_at_least(int d) {
    int w = 2;
    while (w < d) w <<= 1;
    return w;
}

__global__ __launch_bounds__(256) void k_sort_profiles(long long U, const long long *uptr, const int *uitem,
                                                       const float *urating, const long long *iptr, const double *info,
                                                       unsigned long long *ub_key, int2 *ub) {
    __shared__ unsigned long long lkeys[4][SORT_LDS];
    const long long u0 = ((long long)blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6))) * 4;
    if (u0 >= U) return;
    const int lane = lane_id();
    const int g = lane >> 4, gl = lane & 15;
    {   // the short profiles, one per 16-lane group
        const long long u = u0 + g;
        long long a = 0;
        int d = 0;
        if (u < U) { a = uptr[u]; d = (int)(uptr[u + 1] - a); }
        const bool small = d <= 16;
        int wmax = small ? d : 0;
#pragma unroll
        for (int m = 32; m >= 16; m >>= 1) wmax = max(wmax, __shfl_xor(wmax, m, 64));
        wmax = rl32(wmax, 0);
        if (wmax > 0) {
            unsigned long long key = 0ull;   // pads sort last
            int px = 0, py = 0;
            if (small && gl < d) sort_entry(uitem, urating, iptr, info, a + gl, key, px, py);
            bitonic_desc(pow2_at_least(wmax), gl, key, px, py);
            if (small && gl < d) ub[a + gl] = make_int2(px, py);
        }
    }
    for (int q = 0; q < 4; q++) {   // the longer ones on the whole wave
        const long long u = u0 + q;
        if (u >= U) break;
        const long long a = uptr[u];
        const int d = (int)(uptr[u + 1] - a);
        if (d <= 16) continue;
        if (d <= 64) {
            unsigned long long key = 0ull;
            int px = 0, py = 0;
            if (lane < d) sort_entry(uitem, urating, iptr, info, a + lane, key, px, py);
            bitonic_desc(pow2_at_least(d), lane, key, px, py);
            if (lane < d) ub[a + lane] = make_int2(px, py);
            continue;
        }
        // longer than a wave: rank by counting.  The keys are staged in LDS
        // (or, past SORT_LDS of them, in the ub_key scratch) and every entry counts the heavier ones.
        unsigned long long *keys = d <= SORT_LDS ? lkeys[uniform((int)(threadIdx.x >> 6))] : ub_key + a;
        for (int p = lane; p < d; p += 64) {
            unsigned long long key;
            int px, py;
            sort_entry(uitem, urating, iptr, info, a + p, key, px, py);
            keys[p] = key;
        }
        __threadfence_block();
        for (int p = lane; p < d; p += 64) {
            unsigned long long key;
            int px, py;
            sort_entry(uitem, urating, iptr, info, a + p, key, px, py);
            int rank = 0;   // equal keys (an item twice in one profile: AlterEgo rows) keep their order
            for (int o = 0; o < d; o++) rank += (keys[o] > key) || (keys[o] == key && o < p);
            ub[a + rank] = make_int2(px, py);
        }
    }
}

// one thread per CSC entry (item i, its p-th rater u): position of i in u's sorted profile = number of heavier
// co-rated items = length of the prefix this rater contributes.  No atomics; raters stay in ascending user order.
__global__ __launch_bounds__(256) void k_rater_records(int I, long long nnz, const long long *iptr, const int *iuser,
                                                       const long long *uptr, const int2 *ub, RaterRec *rc, int *taken,
                                                       unsigned long long *Wp) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < nnz;
    int i = -1;
    long long pos_sum = 0;
    if (valid) {
        // item of CSC entry p: binary search in iptr
        int lo = 0, hi = I;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (iptr[mid] <= p) lo = mid; else hi = mid;
        }
        i = lo;
        const int u = iuser[p];
        const long long a = uptr[u], b = uptr[u + 1];
        RaterRec r;
        r.e0 = (int)a; r.pos_ge = 0; r.rating = 0.f; r.user = u;
        for (long long e = a; e < b; e++) {
            int2 v = ub[e];
            if ((v.x & 0x7fffffff) == i) {
                // A profile may hold the item more than once (AlterEgo rows: a pass-through and a mapped rating); the
                // copies are adjacent in the sorted profile and the item then has as many CSC entries for this user:
                // each takes one copy (`taken`, zeroed, non-NULL only when the caller allows duplicates).
                if (taken && e + 1 < b && (ub[e + 1].x & 0x7fffffff) == i) {
                    e += atomicAdd(&taken[e], 1);
                    v = ub[e];
                }
                const int pos = (b - a >= 2) ? (int)(e - a) : 0;   // users with >= 2 ratings only (baselinerSim.py:184-185)
                r.pos_ge = (int)((unsigned)pos | ((unsigned)v.x & 0x80000000u));
                r.rating = __int_as_float(v.y);
                pos_sum = pos;
                break;
            }
        }
        rc[p] = r;
    }
    // W+ of the item = sum of its raters' prefix lengths (the contributions of its row): the entries of an item are
    // consecutive, so a wave adds up its runs and issues one atomic per run (exact integers: order does not matter).
    // k_plan2 had walked the rater records of every item for this sum, the popular items' 1e5 records with one wave.
    const int lane = lane_id();
    const int i_prev = __shfl_up(i, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || i != i_prev);
    long long incl = pos_sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const bool seg_end = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
    const int h = 63 - __clzll((long long)(heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull))));
    const long long before = __shfl(incl, h > 0 ? h - 1 : 0, 64);
    if (seg_end && i >= 0) {
        const long long seg = incl - (h > 0 ? before : 0);
        if (seg) atomicAdd(&Wp[i], (unsigned long long)seg);
    }
}
__device__ __forceinline__ void plan_item(int i, long long w, int I, const long long *iptr, const long long *pre, int HB,
                                          const int *hid, const int *CH, int target, int dups, int *Q, int *C,
                                          uint8_t *small, unsigned long long *Wp, int *Qcat) {
    const long long n = iptr[i + 1] - iptr[i];
    long long ge = I - pre[n < HB - 1 ? n : HB - 1];   // #{items with at least as many raters}
    const long long others = ge - 1 + (dups ? 1 : 0);   // with duplicate items a row can pair with itself
    long long bound = w < others ? w : others;
    int q = 0, c = 0;
    if (w > 0) {
        if (hid[i] >= 0) c = (int)((n + *CH - 1) / *CH);
        else q = (int)((bound + target - 1) / target);
    }
    Q[i] = q;
    C[i] = c;
    // rows with very many raters (popular items below the heavy threshold, or every popular item when there is no
    // heavy set: RecommenderSim) are bound by the walk over their raters, not by the table: class 4
    const int cls = (q >= 1 && n >= WIDE_MIN) ? 4
                    : ((q != 1) ? 0 : (bound <= SMALL_BOUND ? 1 : (bound <= 2 * SMALL_BOUND ? 3 : (bound <= MID_BOUND ? 2 : 0))));
    small[i] = (uint8_t)cls;
    Wp[i] = (unsigned long long)w;
    // the light units are listed class-major (largest tables first: their units run longest), so that each table
    // class is one contiguous range of units: Qcat[rank][i] (zero-initialised) is what the unit scan runs over
    Qcat[(size_t)class_rank(cls) * I + i] = q;
}

__global__ __launch_bounds__(256) void k_plan2(int I, const long long *iptr, const RaterRec *rc, const long long *pre,
                                               int HB, const int *hid, const int *CH, int target, int dups, int *Q, int *C,
                                               uint8_t *small, unsigned long long *Wp, int *Qcat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // W+ comes summed from k_rater_records
    if (i >= I) return;
    plan_item(i, (long long)Wp[i], I, iptr, pre, HB, hid, CH, target, dups, Q, C, small, Wp, Qcat);
}

// light unit u: uq_item[u] and the record uq_q[4 u ..] = (partition, first rater, end of raters, partitions of the row) --
// what k_pair_tri needs to start, in one round trip
__global__ __launch_bounds__(256) void k_fill_units2(int I, const long long *iptr, const int *Qcat, const long long *uq_ptr,
                                                     int *uq_item, int *uq_q, const int *C, const long long *uc_ptr, int *uc_item,
                                                     int *uc_c, long long cap_light, long long cap_heavy) {
    const long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (long long)N_CLASSES * I) return;
    const int i = (int)(x % I);
    long long b = uq_ptr[x];
    const int nq = Qcat[x];
    if (nq > 0) {
        const int p0 = (int)iptr[i], p1 = (int)iptr[i + 1];
        for (int k = 0; k < nq && b + k < cap_light; k++) {
            uq_item[b + k] = i;
            ((int4 *)uq_q)[b + k] = make_int4(k, p0, p1, nq);
        }
    }
    if (x >= I) return;
    b = uc_ptr[i];
    for (int k = 0; k < C[i] && b + k < cap_heavy; k++) { uc_item[b + k] = i; uc_c[b + k] = k; }
}

// ---------------------------------------------------------------------------------------------
struct TriArgs {
    const long long *iptr;
    const RaterRec *rc;      // [nnz] rater records in CSC order
    const int2 *ub;          // [nnz] weight-sorted profiles: (item | flag, rating bits)
    const double *u_avg; const double *nrm;   // nrm: dense [I] norm of the method (norm2 | adjnorm2)
    int cap;
    // light
    const int *Q; const uint8_t *small; const int *uq_item; const int *uq_q; long long unit_lo, unit_hi;
    // heavy
    const int *hid; const int *hlist; const int *CH; const int *uc_item; const int *uc_c;
    const long long *uc_ptr; const int *C;
    double *hp_hi; double *hp_lo; int *hp_cnt; int *hp_mut;    // [heavy units][HMAX]
    // output: half COO + per-row counts
    long long shard_cap;            // COO entries per shard
    unsigned long long *shard_cur;  // [COO_SHARDS] cursors
    unsigned long long *shard_occ;  // [COO_SHARDS] unordered pairs evaluated
    int *coo_i; int *coo_j; double *coo_sim; int *coo_mutu; int *coo_nij;
    double *coo_aux;                // optional 6th column (RecommenderSim: local sensitivity)
    int *rowcnt;                    // pairs a row computed itself
    int *mircnt;                    // (host side only: NULL = the mirrored counts are added to rowcnt after the kernels)
    int *rowcnt_h;                  // (unused)
    unsigned long long *counters;   // [2] table overflow, [3] COO overflow
    int heavy_mod, heavy_rem;       // the rows of H this call computes: item index % heavy_mod == heavy_rem (item-sharded ranks
                                    // deal the heavy rows round-robin; 1, 0: all of them)
    int raw;                        // user-sharded input: emit every pair's partial sums (dot as (value, error) in coo_sim /
                                    // coo_aux, n_ij, mutuality) unfinished and unfiltered -- xmap_sim2_merge finishes them
};

// cosine (:91-95), significance weighting (:84-89), zero filter (:198,:207) for one accumulated pair
template <int METHOD>
__device__ __forceinline__ bool finish_pair(const TriArgs &A, int i, int j, int n, int m, double dot, double &simv) {
#ifdef EXP_NONRM      // attribution builds (profiles/tools/a_variants.sh), never shipped
    const double np = A.nrm[i] * A.nrm[i];
#else
    const double np = A.nrm[i] * A.nrm[j];
#endif
    const double cs = (np != 0.0) ? 1.0 * dot / np : 0.0;
    const int mn = n < A.cap ? n : A.cap;
    simv = 1.0 * cs * (double)mn / (double)A.cap;
    return (simv != 0.0) && (m != 0);
}

// append the kept pairs of one wave's table (callback gives slot -> pair) to the half COO
// The COO is cut into COO_SHARDS segments with a cursor each (a single cursor word would serialise the ~4e5
// appending waves: one word sustains only ~90 atomics/us); unused entries keep coo_i = -1.
constexpr int COO_SHARDS = 4096;

// (the mirrored row counts are not taken here any more -- one device atomic per kept pair, with HEAVY_SHARDS replicas for
// the heavy partners, was what the pair kernels waited for: xmap_sim3_mircount / mirror_counts take them from the COO;
// rowcnt_h stays in the signature of xmap_sim2_pairs, unused)

// finalise(s, j, n, m, sim, occupied) -> keep.  Pass 1 finalises every slot once (the result is parked by `park`),
// pass 2 writes the kept ones.
template <typename Fin, typename Park, typename Get, typename Aux>
__device__ __forceinline__ void append_pairs(const TriArgs &A, int i, int s_begin, int n_slots, Fin fin, Park park, Get get,
                                             Aux aux) {
    const int lane = lane_id();
    const int shard = (blockIdx.x * (blockDim.x >> 6) + uniform((int)(threadIdx.x >> 6))) & (COO_SHARDS - 1);
    int kept = 0, occ = 0;
    for (int s0 = s_begin; s0 < n_slots; s0 += 64) {
        int j, n, m; double sv; bool o;
        bool keep = fin(s0 + lane, j, n, m, sv, o);
        park(s0 + lane, o, keep, sv);
        kept += __popcll(__ballot(keep));
        occ += __popcll(__ballot(o));
    }
    if (lane == 0 && occ) atomicAdd(&A.shard_occ[shard], (unsigned long long)occ);
    if (!kept) return;
    unsigned long long base = 0;
    if (lane == 0) {
        base = atomicAdd(&A.shard_cur[shard], (unsigned long long)kept);
        atomicAdd(&A.rowcnt[i], kept);
    }
    base = ((unsigned long long)(unsigned)rl32((int)(base >> 32), 0) << 32) | (unsigned)rl32((int)(base & 0xffffffffull), 0);
    if ((long long)(base + kept) > A.shard_cap) {
        if (lane == 0) atomicOr(&A.counters[3], 1ull);
        return;
    }
    base += (unsigned long long)shard * (unsigned long long)A.shard_cap;
    for (int s0 = s_begin; s0 < n_slots; s0 += 64) {
        int j, n, m; double sv;
        bool keep = get(s0 + lane, j, n, m, sv);
        unsigned long long km = __ballot(keep);
        if (keep) {
            long long p = (long long)base + __popcll(km & lanemask_lt());
            A.coo_i[p] = i; A.coo_j[p] = j;
#ifndef EXP_NOCOO     // (indices still written: the mirror reads them)
            A.coo_sim[p] = sv; A.coo_mutu[p] = m; A.coo_nij[p] = n;
            if (A.coo_aux) A.coo_aux[p] = aux(s0 + lane);
#endif
        }
        base += __popcll(km);
    }
}

// sums of the shard cursors (kept pairs) and of the evaluated-pair counters -> counters[4], counters[5]: what the host reads
// after the pair kernels, in one copy with the overflow flags
__global__ __launch_bounds__(256) void k_shard_sums(const unsigned long long *shards, unsigned long long *counters) {
    unsigned long long a = 0ull, b = 0ull;
    for (int s = threadIdx.x; s < COO_SHARDS; s += 256) { a += shards[s]; b += shards[COO_SHARDS + s]; }
    a = (unsigned long long)wave_sum_ll((long long)a);
    b = (unsigned long long)wave_sum_ll((long long)b);
    if (lane_id() == 0) { atomicAdd(&counters[4], a); atomicAdd(&counters[5], b); }
}

// Light rows.  The co-ratings of a block of raters are walked as one flat list, one per lane (k_pair_tri: walk).  Lanes
// of different raters may meet on one partner: the counters use LDS atomics, the fp64 sum is either an LDS atomic add
// (cosine: integer-exact, order irrelevant) or, for the double-double sum of adjusted cosine, serialised per slot
// (conflicts are rare): through a claim word inside one wave, through a lock bit in the slot's key when several waves
// share the table.

// The table size is a template parameter: rows whose partner bound is <= SMALL_BOUND (the vast majority: items
// with a handful of raters) run with 128 slots (3.5 KB of LDS, full occupancy, 8x cheaper init/finalise), rows up
// to 2 SMALL_BOUND with 256, up to MID_BOUND with 512, the others with 1024.  The units are listed class-major, one
// launch per class.  The big tables are few per CU (5 of 1024 slots fit in LDS) and their rows have the most
// raters (158 on average at BASELINE configs[1], against 6 in the smallest class): NW waves share one table there
// (4 for 1024 slots, 2 for 512), each taking every NW-th block of 64 raters, which keeps 20 waves per CU in flight
// instead of 5.
// LS (RecommenderSim, core/recommenderSim.py:90-133): nothing is filtered, a row may pair with itself (an item twice
// in one profile), and every pair also gets its leave-one-out local sensitivity, which needs the FINAL inner product
// and count of the pair: after the accumulation pass the slots are finalised in LDS and the raters are walked a
// second time, each co-rating looking its slot up and raising the slot's maximum (bit pattern of a non-negative
// double, NaN above everything: np.max propagates NaN).
__device__ __forceinline__ double weighted(double cs, int n, int cap) {
    const int mn = n < cap ? n : cap;
    return 1.0 * cs * (double)mn / (double)cap;
}
__device__ __forceinline__ unsigned long long ls_key(double d) {
    return (d != d) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(d);
}

// -DA_TRACE: per-unit time stamps (profiles/tools/trace_a.py reads them): a unit of the smallest class lives ~14 us -- 2.0 us
// until its item / partition / rater range are read, 3.4 us until the first rater records and prefixes are in, 9.7 us
// until its (single) step of 8 raters is in the table, 4.3 us of finalisation and appends -- and holds its LDS table
// all that time; LDS capacity x unit lifetime (79 GB us over 41 MB of LDS = 1.9 ms) is what bounds the class launches.
#ifdef A_TRACE
__device__ unsigned long long g_atrace[1 << 21][2];   // per light unit: begin, end (wall_clock64, 100 MHz)
__device__ unsigned int g_astamp[1 << 21][4];          // offsets from begin: unit read, first rater records in, walk done
struct ATraceEnd { long long u; long long t0;
    __device__ ~ATraceEnd() { if (threadIdx.x == 0 && u < (1 << 21)) { g_atrace[u][0] = (unsigned long long)t0; g_atrace[u][1] = wall_clock64(); } } };
#define A_STAMP(k) do { if (threadIdx.x == 0 && unit < (1 << 21)) g_astamp[unit][k] = (unsigned)(wall_clock64() - (unsigned long long)tr_.t0); } while (0)
#else
#define A_STAMP(k) do {} while (0)
#endif
// A field of the kernel's argument struct, read from the kernarg segment where it is used (a volatile scalar load: it stays
// at that place).  The finalisation needs seventeen pointers the walk never touches; as plain uses of A they are all loaded
// at the kernel's entry and kept -- 101 SGPRs, i.e. 7 waves per SIMD, or ~100 v_writelane / v_readlane spill moves per unit
// (a sixth of its vector instructions) when the kernel is held to 8.
// KARG reads at offsetof(TriArgs, field) from the kernarg base: correct only while the struct is the kernel's FIRST and ONLY
// parameter (k_pair_tri(TriArgs A), k_pair_heavy(TriArgs A)): keep it so.
static_assert(std::is_standard_layout<TriArgs>::value, "KARG() addresses TriArgs fields by offsetof");
#ifdef EXP_NOKARG
#define KARG(field) (A.field)
#else
#define KARG(field) (*(decltype(TriArgs::field) const volatile __attribute__((address_space(4))) *)( \
    (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TriArgs, field)))
#endif
template <int METHOD, int LOG_SLOTS, int NW, bool LS>
#ifdef EXP_NOMINW
#define PAIR_MINW 1
#else
#define PAIR_MINW ((LOG_SLOTS == 7 && !LS) ? 8 : 1)
#endif
__global__ __launch_bounds__(64 * NW, PAIR_MINW) void k_pair_tri(TriArgs A) {   // (128 slots: 8 waves per SIMD fit, keep the SGPRs below the limit for that)
    constexpr int SLOTS_ = 1 << LOG_SLOTS;
    constexpr bool ADJ = METHOD == XMAP_ADJUST_COSINE;
    using RT = typename std::conditional<LS, double, float>::type;      // rating type of the profile copy and the rater records
    // one slot = key 4 B (item; bit 31 = the slot's lock while several waves share the table) + counters + sum: 24 B (adjusted
    // cosine: (value, error) sum) for the rows below WIDE_MIN raters -- n_ij and the mutuality count in 16 bits each -- so six
    // 1024-slot tables fit one CU (round 2: 32 KB + 8 B each with 64-bit counters and a lock word: four)
    // (profiles hold an item once there: n_ij <= n_i < WIDE_MIN; the AlterEgo rows of LS may repeat items: 32-bit halves)
    using CM = typename std::conditional<NW == 16 || LS, unsigned long long, unsigned>::type;
    constexpr int MSH = (NW == 16 || LS) ? 32 : 16;
    constexpr CM NMASK = (CM)(((CM)1 << MSH) - 1);
    constexpr uint32_t LOCKBIT = 0x80000000u;
    __shared__ uint32_t key[SLOTS_];
    __shared__ CM cm[SLOTS_];                     // n_ij (low half) | mutuality (high half)
    __shared__ double dot[SLOTS_];
    __shared__ double dlo[ADJ ? SLOTS_ : 1];
    __shared__ unsigned short claim[ADJ && NW == 1 ? SLOTS_ : 1];   // (lane ids; NW > 1 locks the key word)
    __shared__ double s_ny[LS ? SLOTS_ : 1];              // LS: norm of the partner
    __shared__ unsigned long long s_ls[LS ? SLOTS_ : 1];  // LS: running maximum (ls_key)
    __shared__ int s_ovf;
    static_assert(!LS || ADJ, "the local-sensitivity pass keeps the similarity in dlo[]");

    const int lane = lane_id();
    const long long unit = A.unit_lo + blockIdx.x;
    if (unit >= A.unit_hi) return;
#ifdef A_TRACE
    ATraceEnd tr_{unit, (long long)wall_clock64()};
#endif
    // the unit's record in one round trip: item; (partition, first rater, end of raters, partitions of the row)
    const int i = uniform(A.uq_item[unit]);
    const int4 ud = ((const int4 *)A.uq_q)[unit];
    const int q = uniform(ud.x), p0 = uniform(ud.y), p1 = uniform(ud.z), Qi = uniform(ud.w);
    const int w = uniform((int)uniform((int)(threadIdx.x >> 6)));      // (a scalar: the block loop of the walk is a scalar loop)
    for (int s = threadIdx.x; s < SLOTS_; s += 64 * NW) {
        key[s] = T_EMPTY; cm[s] = (CM)0; dot[s] = 0.0;
        if (ADJ) dlo[s] = 0.0;
    }
    if (NW > 1) {
        if (threadIdx.x == 0) s_ovf = 0;
        __syncthreads();
    }
    const double nx = A.nrm[i];     // (for the finalisation: in flight during the walk)
    int ovf = 0;
    A_STAMP(0);

    // walk(body): every co-rating of this unit's raters (those of hash partition q); body(act, j, jw, rj, ri, a, gei)
    // runs once per lane and 64 co-ratings.  Wave w takes every NW-th block of RB raters, one rater record per lane
    // (RB = 64 when the wave is alone or the row is very long; the rows of the shared 1024 / 512-slot tables have 158 / 33
    // raters on average: blocks of 16 deal them out evenly -- a unit lives as long as its busiest wave, and holds its table
    // that long).  The prefixes of a block are walked as ONE flat list (round 2 gave each rater 8 lanes: half of the
    // lanes idle, one dependent load per 8 entries of the longest of eight prefixes, 10-18 us per unit of which the
    // table work was a fraction -- profiles/tools/trace_a.py): an inclusive scan of the prefix lengths over the lanes, then
    // lane l of round t takes co-rating 64 t + l, finds its rater by binary search over the scan (log2 RB permutes) and loads
    // its entry; all loads of WU rounds are in flight together and every lane of every round but the last is busy.
#ifndef EXP_RB       // (tuning builds: profiles/tools/a_variants.sh; shipped values below)
#define EXP_RB 16
#endif
#ifndef EXP_WU
#define EXP_WU 2
#endif
    constexpr int RB = (NW == 1 || NW == 16) ? 64 : EXP_RB;
    constexpr int WU = EXP_WU;
    // Loads of the walk: every one is unconditional (a clamped index for a lane that has nothing to load) and nothing is
    // done with a loaded value before the loads that can go out with it are out -- a load under `if (act)`, or a select on a
    // freshly prefetched record, is waited for on the spot, which had put a block's record prefetch, the user averages and the
    // two rounds' entries one round trip after the other.  The user average is read per co-rating next to the profile entry
    // (the lanes of one rater read one address), not per rater ahead of the rounds.
    struct RawRec { int e0, pw, usr; RT r; };
    auto rater = [&](int p) {
        const int pc = (lane < RB && p < p1) ? p : p0;
        RawRec o;
        if (LS) {       // fp64 ratings, user average 0 by construction
            const RaterRecWide rr = ((const RaterRecWide *)A.rc)[pc];
            o.e0 = rr.e0; o.pw = rr.pos_ge; o.r = (RT)rr.rating; o.usr = 0;
        } else {
            const RaterRec rr = A.rc[pc];
            o.e0 = rr.e0; o.pw = rr.pos_ge; o.r = (RT)rr.rating; o.usr = rr.user;
        }
        return o;
    };
    auto entry = [&](int e, int &jw_, RT &rj_) {        // one entry of a sorted profile
        if (LS) { const UbWide v = ((const UbWide *)A.ub)[e]; jw_ = v.item_ge; rj_ = (RT)v.rating; }
        else { const int2 v = A.ub[e]; jw_ = v.x; rj_ = (RT)__int_as_float(v.y); }
    };
    auto walk = [&](auto &&body) {
        int base = p0 + RB * w;
        if (base >= p1) return;
        RawRec cur = rater(base + lane);
        for (; base < p1; base += RB * NW) {
            RawRec nxt = cur;
            if (base + RB * NW < p1) nxt = rater(base + RB * NW + lane);      // this wave's next block (a scalar branch)
            const bool ok = lane < RB && base + lane < p1;
            const int e0 = cur.e0, pw = cur.pw, usr = cur.usr;
            const RT r = cur.r;
            const int len = ok ? (pw & 0x7fffffff) : 0;       // the rater's prefix: entries [e0, e0 + len) of its profile
            int end = len;
#pragma unroll
            for (int d = 1; d < RB; d <<= 1) { const int v = __shfl_up(end, d, 64); if (lane >= d) end += v; }
            const int total = rl32(end, RB - 1);
            const int start = end - len;
#ifdef A_TRACE
            if (base == p0 + RB * w) A_STAMP(1);
#endif
            for (int f0 = 0; f0 < total; f0 += 64 * WU) {
                int jw[WU]; RT rj[WU]; bool act[WU]; double ri[WU], a[WU]; unsigned gei[WU];
                int tt[WU], ee[WU], uu[WU];
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    const int f = f0 + u * 64 + lane;
                    act[u] = f < total;
                    int t = 0;                                // the rater of co-rating f: #{k : end[k] <= f}
#pragma unroll
                    for (int step = RB / 2; step >= 1; step >>= 1) { const int v = __shfl(end, t + step - 1, 64); if (v <= f) t += step; }
                    const int eb = __shfl(e0, t, 64), sb = __shfl(start, t, 64), ut = __shfl(usr, t, 64);
                    tt[u] = t;
                    ee[u] = act[u] ? eb + (f - sb) : 0;
                    uu[u] = act[u] ? ut : 0;
                }
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    entry(ee[u], jw[u], rj[u]);
                    a[u] = 0.0;
#ifndef EXP_NOUAVG
                    if (ADJ && !LS) a[u] = A.u_avg[uu[u]];
#endif
                }
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    const int pwt = __shfl(pw, tt[u], 64);
                    ri[u] = (double)__shfl(r, tt[u], 64);
                    gei[u] = ((unsigned)pwt) >> 31;
                }
#pragma unroll
                for (int u = 0; u < WU; u++) {
                    if (u && f0 + u * 64 >= total) continue;  // (uniform)
                    const int j = jw[u] & 0x7fffffff;
                    bool ac = act[u];
                    if (ac && Qi > 1) ac = (int)__umulhi(mix32((uint32_t)j), (uint32_t)Qi) == q;
                    body(ac, j, jw[u], rj[u], ri[u], a[u], gei[u]);
                }
            }
            cur = nxt;
        }
    };

    // pass 1: accumulate n_ij, mutuality and the dot product per partner
    walk([&](bool act, int j, int jw, RT rj, double ri, double a, unsigned gei) {
        uint32_t h = 0;
        if (act) {
            h = ((uint32_t)j * 0x9E3779B1u) >> (32 - LOG_SLOTS);
            int probes = 0;
            for (;;) {
                uint32_t prev = atomicCAS(&key[h], T_EMPTY, (uint32_t)j);
                if (prev == T_EMPTY || (prev & ~LOCKBIT) == (uint32_t)j) break;
                h = (h + 1) & (SLOTS_ - 1);
                if (++probes >= SLOTS_) { act = false; ovf = 1; break; }
            }
        }
        if (act) {
            const CM inc = (CM)1 | (((((unsigned)jw) >> 31) == gei) ? ((CM)1 << MSH) : (CM)0);
            atomicAdd(&cm[h], inc);
            if (METHOD == XMAP_COSINE) atomicAdd(&dot[h], (1.0 * ri) * (double)rj);   // integer-exact
        }
        if (ADJ) {
            const double term = (ri - a) * ((double)rj - a);
            // volatile: the sums are shared between lanes (and waves); the compiler must neither forward the
            // claim / lock store to the load nor hoist the sum loads out of the loop
            // (LDS-qualified: through a generic volatile pointer these become flat_load / flat_store, which also count on
            // vmcnt -- every turn then waited for the prefix loads in flight as well)
            typedef __attribute__((address_space(3))) volatile double lds_vf64;
            typedef __attribute__((address_space(3))) volatile unsigned short lds_vu16;
            lds_vf64 *vhi = (lds_vf64 *)dot, *vlo = (lds_vf64 *)dlo;
            bool pending = act;
            if (NW == 1) {
                lds_vu16 *vclaim = (lds_vu16 *)claim;
                while (__ballot(pending)) {       // lanes that share a slot take turns
                    if (pending) vclaim[h] = (unsigned short)lane;
                    if (pending && vclaim[h] == (unsigned short)lane) {
                        double hi = vhi[h], lo = vlo[h];
                        dd_add(hi, lo, term);
                        vhi[h] = hi; vlo[h] = lo;
                        pending = false;
                    }
                }
            } else {
                while (__ballot(pending)) {       // a lock per slot (bit 31 of its key): the holder releases in the same pass
                    if (pending && !(atomicOr(&key[h], LOCKBIT) & LOCKBIT)) {
                        double hi = vhi[h], lo = vlo[h];
                        dd_add(hi, lo, term);
                        vhi[h] = hi; vlo[h] = lo;
                        __threadfence_block();
                        atomicAnd(&key[h], ~LOCKBIT);
                        pending = false;
                    }
                }
            }
        }
    });
    A_STAMP(2);
    if (NW > 1) {
        if (ovf) s_ovf = 1;
        __syncthreads();          // all raters are in the table (and every lock bit is clear again)
        ovf = s_ovf;
    }
    if (__ballot(ovf)) {
        if (threadIdx.x == 0) atomicOr(&A.counters[2], 1ull);
        return;
    }
    if (LS) {
        // finalise the slots: an item paired with itself was met once per user holding it twice, the reference lists
        // both orders (recommenderSim.py:71-72): count and inner product double (exactly)
        const double nx = A.nrm[i];
        for (int s = threadIdx.x; s < SLOTS_; s += 64 * NW) {
            const uint32_t kj = key[s];
            if (kj == T_EMPTY) continue;
            int n = (int)(cm[s] & NMASK);
            double inner = dot[s];
            if ((int)kj == i) { n *= 2; inner *= 2.0; }
            const double ny = A.nrm[kj];
            const double np = nx * ny;
            cm[s] = (CM)(unsigned)n;
            dot[s] = inner;
            dlo[s] = weighted((np != 0.0) ? 1.0 * inner / np : 0.0, n, A.cap);   // NaN != 0: divides, like the reference
            s_ny[s] = ny;
            s_ls[s] = 0ull;
        }
        if (NW > 1) __syncthreads();
        // pass 2: leave-one-out variants (recommenderSim.py:98-116)
        walk([&](bool act, int j, int jw, RT rj, double ri, double a, unsigned gei) {
            if (!act) return;
            uint32_t h = ((uint32_t)j * 0x9E3779B1u) >> (32 - LOG_SLOTS);
            while (key[h] != (uint32_t)j) h = (h + 1) & (SLOTS_ - 1);
            const double inner = dot[h], sim = dlo[h], ny = s_ny[h];
            const int n = (int)cm[h];
            const double r0 = ri, r1 = (double)rj;
            const double rest = inner - r0 * r1;
            const double m1 = sqrt((nx * nx - r0 * r0) * (ny * ny));
            const double m2 = sqrt((nx * nx) * (ny * ny - r1 * r1));
            const double d1 = fabs(weighted((m1 != 0.0) ? 1.0 * rest / m1 : 0.0, n - 1, A.cap) - sim);
            const double d2 = fabs(weighted((m2 != 0.0) ? 1.0 * rest / m2 : 0.0, n - 1, A.cap) - sim);
            const unsigned long long k1 = ls_key(d1), k2 = ls_key(d2);
            atomicMax(&s_ls[h], k1 > k2 ? k1 : k2);
        });
        if (NW > 1) __syncthreads();
        append_pairs(A, i, w * (SLOTS_ / NW), (w + 1) * (SLOTS_ / NW),
            [&](int s, int &j, int &n, int &m, double &sv, bool &o) {
                const uint32_t kj = key[s];
                o = kj != T_EMPTY;
                if (!o) return false;
                j = (int)kj; n = (int)cm[s]; m = 0; sv = dlo[s];
                return true;
            },
            [&](int s, bool o, bool keep, double sv) {},
            [&](int s, int &j, int &n, int &m, double &sv) {
                const uint32_t kj = key[s];
                if (kj == T_EMPTY) return false;
                j = (int)kj; n = (int)cm[s]; m = 0; sv = dlo[s];
                return true;
            },
            [&](int s) { return __longlong_as_double((long long)s_ls[s]); });
        return;
    }
    // finalisation: this wave's share of the slots stays in registers (NIT rounds of 64); the norms of all partners are
    // gathered in one go (round 2: norm gather -> cursor atomic -> heavy-id gather, three dependent round trips and the slots
    // read twice from LDS), then one returning atomic on the shard cursor, then the stores.  The mirrored row counts are
    // not taken here (one device atomic per kept pair): xmap_sim2_pairs counts them from the COO afterwards (k_cbs_*)
    // (the argument fields of this part: see KARG)
    const auto k_coo_i = KARG(coo_i);
    const auto k_coo_j = KARG(coo_j);
    const auto k_coo_sim = KARG(coo_sim);
    const auto k_coo_mutu = KARG(coo_mutu);
    const auto k_coo_nij = KARG(coo_nij);
    const auto k_coo_aux = KARG(coo_aux);
    const auto k_nrm = KARG(nrm);
    const auto k_rowcnt = KARG(rowcnt);
    const auto k_shard_cur = KARG(shard_cur);
    const auto k_shard_occ = KARG(shard_occ);
    const auto k_shard_cap = KARG(shard_cap);
    const auto k_cap = KARG(cap);
    const auto k_raw = KARG(raw);
    const auto k_counters = KARG(counters);
    constexpr int NIT = SLOTS_ / NW / 64;
    const int sb0 = w * (SLOTS_ / NW);
    int fj[NIT], fn[NIT], fm[NIT];
    double fs[NIT], fy[NIT], fa[NIT];
    bool fo[NIT], fk[NIT];
#pragma unroll
    for (int t = 0; t < NIT; t++) {
        const int sl = sb0 + t * 64 + lane;
        const uint32_t kj = key[sl];
        fo[t] = kj != T_EMPTY;
        fj[t] = (int)kj;
        const CM c = cm[sl];
        fn[t] = (int)(c & NMASK); fm[t] = (int)(c >> MSH);
        fs[t] = dot[sl];
        fa[t] = (ADJ && k_raw) ? dlo[sl] : 0.0;
        fy[t] = 0.0;
        if (fo[t]) {
#ifdef EXP_NONRM
            fy[t] = nx;
#else
            if (!k_raw) fy[t] = k_nrm[kj];
#endif

        }
    }
    int kept = 0, occ = 0;
#pragma unroll
    for (int t = 0; t < NIT; t++) {
        bool keep = fo[t];
        if (keep && !k_raw) {          // cosine (:91-95), significance weighting (:84-89), zero filter (:198,:207): finish_pair
            const double np = nx * fy[t];
            const double cs = (np != 0.0) ? 1.0 * fs[t] / np : 0.0;
            const int mn = fn[t] < k_cap ? fn[t] : k_cap;
            fs[t] = 1.0 * cs * (double)mn / (double)k_cap;
            keep = (fs[t] != 0.0) && (fm[t] != 0);
        }
        fk[t] = keep;
        kept += __popcll(__ballot(keep));
        occ += __popcll(__ballot(fo[t]));
    }
    const int shard = (blockIdx.x * NW + w) & (COO_SHARDS - 1);
    if (lane == 0 && occ) atomicAdd(&k_shard_occ[shard], (unsigned long long)occ);
    if (!kept) return;
    unsigned long long cbase = 0;
    if (lane == 0) {
        cbase = atomicAdd(&k_shard_cur[shard], (unsigned long long)kept);
        atomicAdd(&k_rowcnt[i], kept);
    }
    cbase = ((unsigned long long)(unsigned)rl32((int)(cbase >> 32), 0) << 32) | (unsigned)rl32((int)(cbase & 0xffffffffull), 0);
    if ((long long)(cbase + kept) > k_shard_cap) {
        if (lane == 0) atomicOr(&k_counters[3], 1ull);
        return;
    }
    cbase += (unsigned long long)shard * (unsigned long long)k_shard_cap;
#pragma unroll
    for (int t = 0; t < NIT; t++) {
        const unsigned long long km = __ballot(fk[t]);
        if (fk[t]) {
            const long long pp = (long long)cbase + __popcll(km & lanemask_lt());
            const int j = fj[t];
            k_coo_i[pp] = i; k_coo_j[pp] = j;
#ifndef EXP_NOCOO
            k_coo_sim[pp] = fs[t]; k_coo_mutu[pp] = fm[t]; k_coo_nij[pp] = fn[t];
            if (k_coo_aux) k_coo_aux[pp] = fa[t];
#endif
        }
        cbase += __popcll(km);
    }
}

// rows of H: chunk c of the raters, dense table over H (partners of a heavy row are heavier, hence in H)
// HEAVY_WAVES waves share the unit's dense table (each takes every HEAVY_WAVES-th block of 64 raters): a unit is a chain
// of dependent gathers (rater record -> prefix entry -> heavy id) and one wave per 26 KB table left 6 waves on a CU.
constexpr int HEAVY_WAVES = 4;
template <int METHOD>
__global__ __launch_bounds__(64 * HEAVY_WAVES) void k_pair_heavy(TriArgs A) {
    __shared__ uint32_t cnt[HMAX];
    __shared__ uint32_t mut[HMAX];
    __shared__ double dot[HMAX];
    __shared__ double dlo[METHOD == XMAP_ADJUST_COSINE ? HMAX : 1];
    __shared__ unsigned lockw[METHOD == XMAP_ADJUST_COSINE ? HMAX : 1];
    const int lane = lane_id(), w = uniform((int)(threadIdx.x >> 6));
    const int unit = blockIdx.x;
    for (int s = threadIdx.x; s < HMAX; s += 64 * HEAVY_WAVES) {
        cnt[s] = 0; mut[s] = 0; dot[s] = 0.0;
        if (METHOD == XMAP_ADJUST_COSINE) { dlo[s] = 0.0; lockw[s] = 0u; }
    }
    __syncthreads();
    const int i = uniform(A.uc_item[unit]);
    if (A.heavy_mod > 1 && (i % A.heavy_mod) != A.heavy_rem) return;      // another rank's heavy row (by ITEM index: the dense
                                                                          // heavy ids are handed out by atomics and differ between ranks)
    const int c = uniform(A.uc_c[unit]);
    const int CH = uniform(*A.CH);
    const int base0 = uniform((int)A.iptr[i]);
    const int p0 = base0 + c * CH;
    int p1 = uniform((int)A.iptr[i + 1]);
    if (p0 + CH < p1) p1 = p0 + CH;
    for (int base = p0 + 64 * w; base < p1; base += 64 * HEAVY_WAVES) {
        const int p = base + lane;
        int e0 = 0, pw = 0;
        float r = 0.f;
        double au = 0.0;
        if (p < p1) {
            const RaterRec rr = A.rc[p];
            e0 = rr.e0; pw = rr.pos_ge; r = rr.rating;
            if (METHOD == XMAP_ADJUST_COSINE) au = A.u_avg[rr.user];
        }
        // one rater per lane: within H a rater's prefix is short (0.8 entries on average at BASELINE configs[1]), so
        // every lane walks its own; lanes that meet on a partner use LDS atomics / a lock word per slot
        const int b1 = (p < p1) ? e0 + (pw & 0x7fffffff) : e0;
        const unsigned gei = ((unsigned)pw) >> 31;
        const double ri = (double)r;
        for (int e = e0; __ballot(e < b1); e++) {
            const bool act = e < b1;
            int h = 0;
            double term = 0.0;
            if (act) {
                const int2 v = A.ub[e];
                const int jw = v.x;
                const float rj = __int_as_float(v.y);
                h = A.hid[jw & 0x7fffffff];
                atomicAdd(&cnt[h], 1u);
                if ((((unsigned)jw) >> 31) == gei) atomicAdd(&mut[h], 1u);
                if (METHOD == XMAP_COSINE) atomicAdd(&dot[h], (1.0 * ri) * (double)rj);   // integer-exact
                else term = (ri - au) * ((double)rj - au);
            }
            if (METHOD == XMAP_ADJUST_COSINE) {
                volatile double *vhi = dot, *vlo = dlo;
                bool pending = act;
                while (__ballot(pending)) {       // a lock per slot: the holder releases in the same pass
                    if (pending && atomicCAS(&lockw[h], 0u, 1u) == 0u) {
                        double hi = vhi[h], lo = vlo[h];
                        dd_add(hi, lo, term);
                        vhi[h] = hi; vlo[h] = lo;
                        __threadfence_block();
                        atomicExch(&lockw[h], 0u);
                        pending = false;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < HMAX; s += 64 * HEAVY_WAVES) {
        size_t o = (size_t)unit * HMAX + s;
        A.hp_cnt[o] = (int)cnt[s];
        A.hp_mut[o] = (int)mut[s];
        A.hp_hi[o] = dot[s];
        A.hp_lo[o] = (METHOD == XMAP_ADJUST_COSINE) ? dlo[s] : 0.0;
    }
}

template <int METHOD>
__global__ __launch_bounds__(256) void k_heavy_merge(TriArgs A, int n_heavy) {
    __shared__ uint32_t cnt[HMAX];
    __shared__ uint32_t mut[HMAX];
    __shared__ double dot[HMAX];
    const int h = blockIdx.x;
    if (h >= n_heavy) return;
    const int i = A.hlist[h];
    if (A.heavy_mod > 1 && (i % A.heavy_mod) != A.heavy_rem) return;
    const int nc = A.C[i];
    const long long u0 = A.uc_ptr[i];
    // four waves per row: the most popular item has ~80 chunks of partials to fold
    for (int s = threadIdx.x; s < HMAX; s += 256) {
        unsigned cn = 0, mu = 0;
        double hi = 0.0, lo = 0.0;
        for (int c = 0; c < nc; c++) {
            size_t o = (size_t)(u0 + c) * HMAX + s;
            cn += (unsigned)A.hp_cnt[o];
            mu += (unsigned)A.hp_mut[o];
            if (METHOD == XMAP_COSINE) {
                hi += A.hp_hi[o];
            } else {
                dd_add(hi, lo, A.hp_hi[o]);
                dd_add(hi, lo, A.hp_lo[o]);
            }
        }
        cnt[s] = cn; mut[s] = mu; dot[s] = hi;
    }
    if (nc == 0) return;
    __syncthreads();
    const int w = uniform((int)(threadIdx.x >> 6));
    append_pairs(A, i, w * (HMAX / 4), (w + 1) * (HMAX / 4),
        [&](int s, int &j, int &n, int &m, double &sv, bool &o) {
            o = cnt[s] != 0;
            if (!o) return false;
            j = A.hlist[s]; n = (int)cnt[s]; m = (int)mut[s];
            return finish_pair<METHOD>(A, i, j, n, m, dot[s], sv);
        },
        [&](int s, bool o, bool keep, double sv) {
            if (o) { if (keep) dot[s] = sv; else cnt[s] = 0; }
        },
        [&](int s, int &j, int &n, int &m, double &sv) {
            if (cnt[s] == 0) return false;
            j = A.hlist[s]; n = (int)cnt[s]; m = (int)mut[s]; sv = dot[s];
            return true;
        },
        [&](int s) { return 0.0; });
}

// Mirror the half COO into the CSR.  Records of one unit are contiguous and share the lighter item i, so the
// i-side cursor is bumped once per run of equal i inside a wave and those writes are coalesced; the j-side
// (heavier item) writes are scattered.
// SC_U groups of 64 COO slots per wave, their loads, cursor atomics and row-pointer gathers issued together (a slot is a
// chain of four dependent round trips: coo_i -> the rest of the record -> cursors / row pointers -> writes): 2.65 -> 2.53 ms.
// Neither the latency nor the cursor atomics bound the kernel (removing the j-side atomic altogether: 2.3 ms); what is
// left is its traffic, 7.2 GB for 1.7 GB of records (four partial-sector writes per mirrored entry).
constexpr int SC_U = 4;
__global__ __launch_bounds__(256) void k_scatter(long long n, const int *coo_i, const int *coo_j, const double *coo_sim,
                                                 const int *coo_mutu, const int *coo_nij, const double *coo_aux,
                                                 const long long *row_ptr, int *fill, int *col, double *sim, int *mutu,
                                                 int *nij, double *aux) {
    const int lane = lane_id();
    const long long r0 = ((long long)blockIdx.x * (blockDim.x >> 6) + uniform((int)(threadIdx.x >> 6))) * (64 * SC_U) + lane;
    int i[SC_U], j[SC_U], m[SC_U], nn[SC_U];
    double s[SC_U], x[SC_U];
    bool valid[SC_U];
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
        const long long r = r0 + 64 * u;
        i[u] = (r < n) ? coo_i[r] : -1;
    }
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
        const long long r = r0 + 64 * u;
        valid[u] = i[u] >= 0;
        j[u] = 0; m[u] = 0; nn[u] = 0; s[u] = 0.0; x[u] = 0.0;
        if (valid[u]) {
            j[u] = coo_j[r]; s[u] = coo_sim[r]; m[u] = coo_mutu[r]; nn[u] = coo_nij[r];
            if (coo_aux) x[u] = coo_aux[r];
        } else {
            i[u] = -1 - lane;   // inactive lanes: unique fake rows
        }
    }
    int base[SC_U], lead[SC_U], bj[SC_U];
    long long rpi[SC_U], rpj[SC_U];
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
        const int prev = __shfl_up(i[u], 1, 64);
        const bool leader = (lane == 0) || (prev != i[u]);
        const unsigned long long lm = __ballot(leader);
        const unsigned long long below = lm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        lead[u] = 63 - __clzll((long long)below);
        const unsigned long long above = (lane == 63) ? 0ull : (lm >> (lane + 1));
        // run length as seen by the leader: distance to the next leader
        const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
        base[u] = 0; bj[u] = 0; rpi[u] = 0; rpj[u] = 0;
        if (leader && valid[u]) base[u] = atomicAdd(&fill[i[u]], next - lane);
        if (valid[u]) {
            rpi[u] = row_ptr[i[u]];
            if (j[u] != i[u]) {   // (a row paired with itself -- RecommenderSim -- is one entry)
                bj[u] = atomicAdd(&fill[j[u]], 1);
                rpj[u] = row_ptr[j[u]];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
        const int bs = __shfl(base[u], lead[u], 64);
        if (valid[u]) {
            const long long a = rpi[u] + bs + (lane - lead[u]);
            col[a] = j[u]; sim[a] = s[u]; mutu[a] = m[u]; nij[a] = nn[u];
            if (aux) aux[a] = x[u];
            if (j[u] != i[u]) {
                const long long b = rpj[u] + bj[u];
                col[b] = i[u]; sim[b] = s[u]; mutu[b] = m[u]; nij[b] = nn[u];
                if (aux) aux[b] = x[u];
            }
        }
    }
}

// ---- user-sharded input ("each GPU produces partial (dot, n, mutu) for all pairs touched by its users") ------
// A partial record is 32 bytes: key = lower item index << 32 | higher index, the dot product as an exact (value, error)
// pair, n_ij | mutuality << 32.  A rank has at most one record per pair (every unordered pair belongs to one work unit).
__global__ __launch_bounds__(256) void k_pack_partials(long long n_coo, const int *coo_i, const int *coo_j, const double *coo_hi,
                                                       const double *coo_lo, const int *coo_mutu, const int *coo_nij,
                                                       unsigned long long *cursor, long long *rec) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = e < n_coo && coo_i[e] >= 0;
    const unsigned long long m = __ballot(v);
    if (!m) return;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)rl32((int)(base >> 32), 0) << 32) | (unsigned)rl32((int)(base & 0xffffffffull), 0);
    if (!v) return;
    const long long p = (long long)base + __popcll(m & lanemask_lt());
    const int i = coo_i[e], j = coo_j[e];
    const int a = i < j ? i : j, b = i < j ? j : i;
    rec[p * 4 + 0] = ((long long)a << 32) | (long long)(unsigned)b;
    rec[p * 4 + 1] = __double_as_longlong(coo_hi[e]);
    rec[p * 4 + 2] = __double_as_longlong(coo_lo[e]);
    rec[p * 4 + 3] = (long long)(unsigned)coo_nij[e] | ((long long)coo_mutu[e] << 32);
}

// sort key of a record: the pair key squeezed to 2 * bits_b bits (same order), or -- n_owners > 0, before the exchange --
// the rank that owns the lower item (items [I r / n_owners, I (r + 1) / n_owners) belong to rank r)
// the kept pairs of a rank for the stage-B exchange: valid COO entries -> 24-byte records (i | j << 32, sim bits,
// mutu | n_ij << 32), and back
__global__ __launch_bounds__(256) void k_pack_pairs(long long n_coo, const int *coo_i, const int *coo_j, const double *coo_sim,
                                                    const int *coo_mutu, const int *coo_nij, unsigned long long *cursor,
                                                    long long *rec) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = e < n_coo && coo_i[e] >= 0;
    const unsigned long long m = __ballot(v);
    if (!m) return;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)rl32((int)(base >> 32), 0) << 32) | (unsigned)rl32((int)(base & 0xffffffffull), 0);
    if (!v) return;
    const long long p = (long long)base + __popcll(m & lanemask_lt());
    rec[p * 3 + 0] = (long long)(unsigned)coo_i[e] | ((long long)coo_j[e] << 32);
    rec[p * 3 + 1] = __double_as_longlong(coo_sim[e]);
    rec[p * 3 + 2] = (long long)(unsigned)coo_mutu[e] | ((long long)coo_nij[e] << 32);
}

__global__ __launch_bounds__(256) void k_unpack_pairs(long long n, const long long *rec, int *coo_i, int *coo_j, double *coo_sim,
                                                      int *coo_mutu, int *coo_nij) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long a = rec[t * 3], c = rec[t * 3 + 2];
    coo_i[t] = (int)(a & 0xffffffffll); coo_j[t] = (int)(a >> 32);
    coo_sim[t] = __longlong_as_double(rec[t * 3 + 1]);
    coo_mutu[t] = (int)(c & 0xffffffffll); coo_nij[t] = (int)(c >> 32);
}

__global__ __launch_bounds__(256) void k_partial_keys(long long n, const long long *rec, int bits_b, int n_items, int n_owners,
                                                      unsigned long long *keys, int *vals) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned long long k = (unsigned long long)rec[t * 4];
    const long long a = (long long)(k >> 32), b = (long long)(k & 0xffffffffull);
    if (n_owners > 0) {
        long long r = a * n_owners / n_items;
        while (r + 1 < n_owners && (long long)n_items * (r + 1) / n_owners <= a) r++;
        while (r > 0 && (long long)n_items * r / n_owners > a) r--;
        keys[t] = (unsigned long long)r;
    } else {
        keys[t] = ((unsigned long long)a << bits_b) | (unsigned long long)b;
    }
    vals[t] = (int)t;
}

__global__ __launch_bounds__(256) void k_partial_gather(long long n, const long long *rec, const int *vals, long long *out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long s = vals[t];
    const longlong2 a = *(const longlong2 *)(rec + s * 4), b = *(const longlong2 *)(rec + s * 4 + 2);
    *(longlong2 *)(out + t * 4) = a;
    *(longlong2 *)(out + t * 4 + 2) = b;
}

// records sorted by key, equal keys in rank order: the thread at the head of a run adds the run up (the dot product
// exactly: the shares are exact (value, error) pairs), finishes the pair like finish_pair and appends it to the half COO
template <int METHOD>
__global__ __launch_bounds__(256) void k_merge_partials(long long n, const long long *rec, const double *nrm, int cap,
                                                        unsigned long long *counters /*[0] kept, [1] evaluated*/, int *coo_i,
                                                        int *coo_j, double *coo_sim, int *coo_mutu, int *coo_nij, int *rowcnt) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool head = false, keep = false;
    int i = 0, j = 0, nn = 0, mm = 0;
    double simv = 0.0;
    if (t < n) {
        const long long key = rec[t * 4];
        head = t == 0 || rec[(t - 1) * 4] != key;
        if (head) {
            double hi = 0.0, lo = 0.0;
            for (long long u = t; u < n && rec[u * 4] == key; u++) {
                const double ph = __longlong_as_double(rec[u * 4 + 1]), pl = __longlong_as_double(rec[u * 4 + 2]);
                if (METHOD == XMAP_COSINE) hi += ph;          // integer-exact
                else { dd_add(hi, lo, ph); dd_add(hi, lo, pl); }
                const unsigned long long c = (unsigned long long)rec[u * 4 + 3];
                nn += (int)(c & 0xffffffffull); mm += (int)(c >> 32);
            }
            i = (int)(key >> 32); j = (int)(key & 0xffffffffll);
            const double np = nrm[i] * nrm[j];                 // finish_pair
            const double cs = (np != 0.0) ? 1.0 * hi / np : 0.0;
            const int mn = nn < cap ? nn : cap;
            simv = 1.0 * cs * (double)mn / (double)cap;
            keep = (simv != 0.0) && (mm != 0);
        }
    }
    const unsigned long long hm = __ballot(head), km = __ballot(keep);
    if (!hm) return;
    unsigned long long base = 0;
    if (lane_id() == 0) {
        atomicAdd(&counters[1], (unsigned long long)__popcll(hm));
        if (km) base = atomicAdd(&counters[0], (unsigned long long)__popcll(km));
    }
    base = ((unsigned long long)(unsigned)rl32((int)(base >> 32), 0) << 32) | (unsigned)rl32((int)(base & 0xffffffffull), 0);
    if (!keep) return;
    const long long p = (long long)base + __popcll(km & lanemask_lt());
    coo_i[p] = i; coo_j[p] = j; coo_sim[p] = simv; coo_mutu[p] = mm; coo_nij[p] = nn;
    atomicAdd(&rowcnt[i], 1);
    atomicAdd(&rowcnt[j], 1);
}


// =================================================================================================================
// Round 3: ONE transposition per pass.  Round 2 built the CSC (count + scan + fill with returning cursor atomics), read
// it for the item statistics, and transposed a second time in k_rater_records; the mirror of the kept pairs was a third
// scatter with cursor atomics.  Now: item counts and rating sums in one pass over the CSR (k_count3) -> profiles sorted
// by weight, each entry leaving as a 16-byte sort record keyed by its item (k_sort_profiles3) -> the records moved to
// item order by the two-level tile sort (tilesort.h), where the rater records get their final form and W+ is summed ->
// item statistics from the rater records (k_item_stats3).  The CSC arrays are not built at all.
// =================================================================================================================
constexpr int CNT_SLOTS = 4096;
constexpr int CNT_CHUNK = 8192;
__device__ __forceinline__ int cnt_slot(int it) { return (int)(mix32((uint32_t)it) & (CNT_SLOTS - 1)); }

// raters per item.  Popular items (8e4 raters at BASELINE configs[1]) would serialise that many atomics on one word:
// every workgroup counts its entries in a direct-mapped LDS cache of (item, count) slots first and goes to memory once
// per occupied slot; entries whose slot is taken by another item use the global word directly.
__global__ __launch_bounds__(256) void k_count3(long long nnz, const int *uitem, int *cnt) {
    __shared__ int tag[CNT_SLOTS], loc[CNT_SLOTS];
    for (int t = threadIdx.x; t < CNT_SLOTS; t += 256) { tag[t] = -1; loc[t] = 0; }
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * CNT_CHUNK;
    for (int q = threadIdx.x; q < CNT_CHUNK; q += 256) {
        const long long e = e0 + q;
        if (e >= nnz) break;
        const int it = uitem[e];
        const int sl = cnt_slot(it);
        const int old = atomicCAS(&tag[sl], -1, it);
        if (old == -1 || old == it) atomicAdd(&loc[sl], 1); else atomicAdd(&cnt[it], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CNT_SLOTS; t += 256)
        if (loc[t]) atomicAdd(&cnt[tag[t]], loc[t]);
}

// The same counts for large inputs without a global atomic per rating (k_count3 issues ~0.9 per rating whatever its LDS cache
// catches: a chunk of 8192 ratings holds ~7000 different items; 9.7e6 device-scope atomics are 0.37 ms at BASELINE configs[1]).
// The item column is first partitioned into <= 1024 buckets of 2^sh consecutive items -- bucket histogram, scan, scatter with
// one reservation per (workgroup, bucket) -- and then counted slice by slice of the partitioned column in an LDS window of
// CB_WIN items anchored at the slice's first bucket: a slice of 8192 entries lies in one or two buckets, so its counts leave
// as one atomic per item it holds (~3e6 atomics in all, three coalesced passes over 39 MB).
constexpr int CB_MAX = 1024;          // buckets
constexpr int CB_CHUNK = 8192;        // entries per workgroup of the histogram / scatter / count passes
constexpr int CB_WIN = 8192;          // items of the count pass's LDS window
constexpr int CB_SPB = 4;             // segments (chunks) per workgroup
constexpr int CB_T = 1024;            // threads per workgroup of the three passes

// exclusive scan of the bucket counts (one workgroup of CB_MAX threads); clears the scatter cursors
__global__ __launch_bounds__(CB_MAX) void k_cb_scan(const unsigned *bcnt, long long *bptr, unsigned *bcur) {
    __shared__ long long ws[CB_MAX / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long c = bcnt[t];
    long long inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    long long base = 0;
    for (int x = 0; x < w; x++) base += ws[x];
    bptr[t] = base + inc - c;
    if (t == CB_MAX - 1) bptr[CB_MAX] = base + inc;
    bcur[t] = 0u;
}

__global__ __launch_bounds__(CB_T) void k_cb_count(long long nnz, const int *part, int sh, const long long *bptr, int n_items, int *cnt) {
    __shared__ unsigned win[CB_WIN];
    __shared__ int s_b;
    if (nnz < 0) nnz = bptr[CB_MAX];       // (the partitioned column's length is only known on the device)
    const long long p0 = (long long)blockIdx.x * (CB_CHUNK * CB_SPB);
    if (p0 >= nnz) return;
    if (threadIdx.x == 0) {       // the bucket that holds position p0: the last b with bptr[b] <= p0
        int lo = 0, hi = CB_MAX;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bptr[mid] <= p0) lo = mid; else hi = mid; }
        s_b = lo;
    }
    for (int t = threadIdx.x; t < CB_WIN; t += CB_T) win[t] = 0u;
    __syncthreads();
    const int item0 = s_b << sh;
#pragma unroll 4
    for (int q = threadIdx.x; q < CB_CHUNK * CB_SPB; q += CB_T) {
        const long long p = p0 + q;
        if (p >= nnz) break;
        const int it = part[p];
        const unsigned d = (unsigned)(it - item0);
        if (d < (unsigned)CB_WIN) atomicAdd(&win[d], 1u); else atomicAdd(&cnt[it], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CB_WIN; t += CB_T)
        if (win[t] && item0 + t < n_items) atomicAdd(&cnt[item0 + t], (int)win[t]);
}

// The same three passes over the partner column of the half COO (shard s = entries [s shard_cap, s shard_cap + cur[s])): the
// mirrored row counts.  Round 2 / 3 counted them with one device-scope atomic per kept pair inside the pair kernels
// (2.65e7 per pass at BASELINE configs[1], replicas for the heavy partners): taken out, the class launches are 31 % shorter
// (2.77 -> 1.91 ms summed; profiles/r03e_pair_mirsep.txt) -- the atomics, not the walk, were what the kernels waited for.
// SELF: an entry may pair a row with itself (RecommenderSim) and then has no mirrored entry.
// A workgroup of 1024 threads takes CB_SPB segments (a segment = chunk c of range r, listed chunk-major: with a sharded COO
// consecutive segments are the same chunk of consecutive shards, each a quarter full) into ONE LDS histogram: the global
// atomics of these passes are one per (workgroup, bucket), so the more entries a workgroup holds the fewer there are.
template <bool SELF>
__global__ __launch_bounds__(CB_T) void k_cbs_hist(long long range_cap, int n_ranges, const unsigned long long *cur, const int *coo_i,
                                                   const int *coo_j, int sh, unsigned *bcnt) {
    __shared__ unsigned h[CB_MAX];
    for (int t = threadIdx.x; t < CB_MAX; t += CB_T) h[t] = 0u;
    __syncthreads();
    const long long cpr = (range_cap + CB_CHUNK - 1) / CB_CHUNK;        // chunks per range
#pragma unroll
    for (int sg = 0; sg < CB_SPB; sg++) {
        const long long g = (long long)blockIdx.x * CB_SPB + sg;
        const long long c = g / n_ranges;
        const int r = (int)(g - c * n_ranges);
        if (c >= cpr) break;
        const long long n = (cur && (long long)cur[r] < range_cap) ? (long long)cur[r] : range_cap;       // (no cursors: all valid)
        const long long b = (long long)r * range_cap;
#pragma unroll
        for (int q = 0; q < CB_CHUNK / CB_T; q++) {
            const long long e = c * CB_CHUNK + q * CB_T + threadIdx.x;
            if (e < n) {
                const int j = coo_j[b + e];
                if (!SELF || j != coo_i[b + e]) atomicAdd(&h[j >> sh], 1u);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CB_MAX; t += CB_T)
        if (h[t]) atomicAdd(&bcnt[t], h[t]);
}

// The workgroup's entries are staged in LDS in bucket order and leave as runs (consecutive lanes write consecutive words
// of a bucket's range): written straight from the registers every lane hits another bucket, a 4-byte transaction each
// (0.25 ms for the 2.65e7 partners of a pass against 0.06 ms for the histogram pass over the same data).
template <bool SELF>
__global__ __launch_bounds__(CB_T) void k_cbs_scatter(long long range_cap, int n_ranges, const unsigned long long *cur, const int *coo_i,
                                                      const int *coo_j, int sh, const long long *bptr, unsigned *bcur, int *out) {
    static_assert(CB_T == CB_MAX, "one thread per bucket in the scan");
    __shared__ unsigned h[CB_MAX], off[CB_MAX + 1], wsum[CB_T / 64];
    __shared__ long long base[CB_MAX];
    __shared__ int stage[CB_CHUNK * CB_SPB];
    const int tid = threadIdx.x;
    h[tid] = 0u;
    __syncthreads();
    constexpr int EPT = CB_CHUNK / CB_T;
    const long long cpr = (range_cap + CB_CHUNK - 1) / CB_CHUNK;
    int it[CB_SPB][EPT];
    unsigned rk[CB_SPB][EPT];
#pragma unroll
    for (int sg = 0; sg < CB_SPB; sg++) {
        const long long g = (long long)blockIdx.x * CB_SPB + sg;
        const long long c = g / n_ranges;
        const int r = (int)(g - c * n_ranges);
        const long long n = (c < cpr) ? ((cur && (long long)cur[r] < range_cap) ? (long long)cur[r] : range_cap) : 0;
        const long long b = (long long)r * range_cap;
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const long long e = c * CB_CHUNK + q * CB_T + tid;
            it[sg][q] = -1;
            if (e < n) {
                const int j = coo_j[b + e];
                if (!SELF || j != coo_i[b + e]) it[sg][q] = j;
            }
        }
    }
#pragma unroll
    for (int sg = 0; sg < CB_SPB; sg++)
#pragma unroll
        for (int q = 0; q < EPT; q++) rk[sg][q] = it[sg][q] >= 0 ? atomicAdd(&h[it[sg][q] >> sh], 1u) : 0u;
    __syncthreads();
    {   // exclusive scan of the bucket counts (one bucket per thread), and the workgroup's place in every bucket's range
        const unsigned c = h[tid];
        const int lane = tid & 63, w = tid >> 6;
        unsigned inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        unsigned bs = 0;
        for (int x = 0; x < w; x++) bs += wsum[x];
        off[tid] = bs + inc - c;
        if (tid == CB_T - 1) off[CB_MAX] = bs + inc;
        if (c) base[tid] = bptr[tid] + (long long)atomicAdd(&bcur[tid], c);
    }
    __syncthreads();
#pragma unroll
    for (int sg = 0; sg < CB_SPB; sg++)
#pragma unroll
        for (int q = 0; q < EPT; q++)
            if (it[sg][q] >= 0) stage[off[it[sg][q] >> sh] + rk[sg][q]] = it[sg][q];
    __syncthreads();
    const int total = (int)off[CB_MAX];
    for (int x = tid; x < total; x += CB_T) {
        const int v = stage[x];
        const int bk = v >> sh;
        out[base[bk] + (x - (int)off[bk])] = v;
    }
}

// Sort records (tilesort.h: key = low 32 bits of word 0).  Narrow (float ratings): {item, pos | flag << 31, rating bits,
// user}.  Wide (fp64 ratings -- RecommenderSim over AlterEgo means, core/recommenderSim.py:64-133 takes np.float64): {item,
// pos | flag, rating (8 B), user, -}.

// The mutuality flag (rating >= item average) needs the item averages, which come out of the rater records this sort
// feeds: the profile copy and the sort records leave without it, k_item_stats3 sets it in the rater records and
// k_ub_flags in the profile copy.
template <bool WIDE>
__device__ __forceinline__ void sort_entry3(const int *uitem, const float *ur32, const double *ur64, const int *cnt,
                                            long long e, unsigned long long &key, int &px, long long &py) {
    const int it = uitem[e];
    key = wkey(cnt[it], it);
    px = it;
    py = WIDE ? __double_as_longlong(ur64[e]) : (long long)(unsigned)__float_as_int(ur32[e]);
}

template <bool WIDE>
__device__ __forceinline__ void bitonic_desc3(int width, int pos, unsigned long long &key, int &px, long long &py) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
        if (k2 <= width)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long ok = __shfl_xor(key, j, 64);
            const int ox = __shfl_xor(px, j, 64);
            long long oy;
            if (WIDE) oy = __shfl_xor(py, j, 64); else oy = (long long)(unsigned)__shfl_xor((int)py, j, 64);
            const bool desc = (pos & k2) == 0;
            const bool lower = (pos & j) == 0;
            const bool take_other = (lower == desc) ? (ok > key) : (ok < key);
            if (take_other) { key = ok; px = ox; py = oy; }
        }
    }
}

// one sorted entry at position `rank` of user u's profile [a, a + d): the profile copy the pair kernel walks and the sort
// record that becomes the item's rater record
template <bool WIDE>
__device__ __forceinline__ void emit_entry3(long long a, int d, int rank, long long u, int px, long long py, void *ub,
                                            unsigned long long *srec) {
    const long long e = a + rank;
    const unsigned pos_ge = (unsigned)(d >= 2 ? rank : 0) | ((unsigned)px & 0x80000000u);   // users with >= 2 ratings only (:184-185)
    const unsigned long long w0 = (unsigned long long)((unsigned)px & 0x7fffffffu) | ((unsigned long long)pos_ge << 32);
    if (WIDE) {
        UbWide v; v.item_ge = px; v.pad = 0; v.rating = __longlong_as_double(py);
        ((UbWide *)ub)[e] = v;
        srec[e * 3 + 0] = w0; srec[e * 3 + 1] = (unsigned long long)py; srec[e * 3 + 2] = (unsigned long long)(unsigned)u;
    } else {
        ((int2 *)ub)[e] = make_int2(px, (int)py);
        ulonglong2 w; w.x = w0; w.y = (unsigned long long)(unsigned)py | ((unsigned long long)(unsigned)u << 32);
        *(ulonglong2 *)(srec + e * 2) = w;
    }
}

// k_sort_profiles with the item's rater count as the only gather and the sort records as second output
template <bool WIDE>
__global__ __launch_bounds__(256) void k_sort_profiles3(long long U, const long long *uptr, const int *uitem, const float *ur32,
                                                        const double *ur64, const int *cnt, unsigned long long *ub_key,
                                                        void *ub, unsigned long long *srec) {
    __shared__ unsigned long long lkeys[4][SORT_LDS];
    const long long u0 = ((long long)blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6))) * 4;
    if (u0 >= U) return;
    const int lane = lane_id();
    const int g = lane >> 4, gl = lane & 15;
    {   // the short profiles, one per 16-lane group
        const long long u = u0 + g;
        long long a = 0;
        int d = 0;
        if (u < U) { a = uptr[u]; d = (int)(uptr[u + 1] - a); }
        const bool small = d <= 16;
        int wmax = small ? d : 0;
#pragma unroll
        for (int m = 32; m >= 16; m >>= 1) wmax = max(wmax, __shfl_xor(wmax, m, 64));
        wmax = rl32(wmax, 0);
        if (wmax > 0) {
            unsigned long long key = 0ull;   // pads sort last
            int px = 0;
            long long py = 0;
            if (small && gl < d) sort_entry3<WIDE>(uitem, ur32, ur64, cnt, a + gl, key, px, py);
            bitonic_desc3<WIDE>(pow2_at_least(wmax), gl, key, px, py);
            if (small && gl < d) emit_entry3<WIDE>(a, d, gl, u, px, py, ub, srec);
        }
    }
    for (int q = 0; q < 4; q++) {   // the longer ones on the whole wave
        const long long u = u0 + q;
        if (u >= U) break;
        const long long a = uptr[u];
        const int d = (int)(uptr[u + 1] - a);
        if (d <= 16) continue;
        if (d <= 64) {
            unsigned long long key = 0ull;
            int px = 0;
            long long py = 0;
            if (lane < d) sort_entry3<WIDE>(uitem, ur32, ur64, cnt, a + lane, key, px, py);
            bitonic_desc3<WIDE>(pow2_at_least(d), lane, key, px, py);
            if (lane < d) emit_entry3<WIDE>(a, d, lane, u, px, py, ub, srec);
            continue;
        }
        unsigned long long *keys = d <= SORT_LDS ? lkeys[uniform((int)(threadIdx.x >> 6))] : ub_key + a;
        for (int p = lane; p < d; p += 64) {
            unsigned long long key;
            int px;
            long long py;
            sort_entry3<WIDE>(uitem, ur32, ur64, cnt, a + p, key, px, py);
            keys[p] = key;
        }
        __threadfence_block();
        for (int p = lane; p < d; p += 64) {
            unsigned long long key;
            int px;
            long long py;
            sort_entry3<WIDE>(uitem, ur32, ur64, cnt, a + p, key, px, py);
            int rank = 0;   // equal keys (an item twice in one profile: AlterEgo rows) keep their order
            for (int o = 0; o < d; o++) rank += (keys[o] > key) || (keys[o] == key && o < p);
            emit_entry3<WIDE>(a, d, rank, u, px, py, ub, srec);
        }
    }
}

// final form of a rater record from its sort record: e0 = first entry of the user's profile
template <bool WIDE>
__device__ __forceinline__ ulonglong2 rater_record(const unsigned long long *w, const long long *uptr) {
    const unsigned user = WIDE ? (unsigned)w[2] : (unsigned)(w[1] >> 32);
    const unsigned e0 = (unsigned)uptr[user];
    ulonglong2 o;
    o.x = (unsigned long long)e0 | (w[0] & 0xffffffff00000000ull);     // {e0, pos | flag}
    o.y = w[1];                                                        // narrow: {rating bits, user}; wide: the fp64 rating
    return o;
}

// level C of the rater records: one workgroup per tile.  The small keys' records are ranked by LDS cursors, converted,
// laid out in final order in LDS and written as whole rows; W+ of every small key (sum of its raters' prefix lengths = the
// contributions of its row) is summed on the way.
template <bool WIDE>
__global__ __launch_bounds__(ts::CT) void k_rc_tiles(ts::Geo G, const unsigned long long *bufB, const long long *uptr,
                                                     ulonglong2 *rc, unsigned long long *Wp) {
    constexpr int RW = WIDE ? 3 : 2;
    __shared__ unsigned cur[ts::NK_MAX], kst[ts::NK_MAX];
    __shared__ unsigned long long wsum[ts::NK_MAX];
    __shared__ ulonglong2 lrec[ts::CAP];
    const ts::TileHead h = ts::tile_head(G, blockIdx.x);
    if (h.nk <= 0) return;
    for (int x = threadIdx.x; x < h.nk; x += ts::CT) {
        cur[x] = 0u; wsum[x] = 0ull;
        kst[x] = (unsigned)(G.ptr[h.k0 + x] - h.pos0);
    }
    __syncthreads();
    const bool in_lds = h.n <= ts::CAP;
    constexpr int UN = 4;
    for (int base = 0; base < h.n; base += ts::CT * UN) {
        unsigned long long w[UN][RW];
        bool on[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            const int idx = base + t * ts::CT + threadIdx.x;
            on[t] = idx < h.n;
            const size_t o = (size_t)(h.pos0 + (on[t] ? idx : 0)) * RW;
#pragma unroll
            for (int x = 0; x < RW; x++) w[t][x] = bufB[o + x];
        }
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if (!on[t]) continue;
            const int kk = (int)((unsigned)w[t][0]) - h.k0;
            const unsigned q = kst[kk] + atomicAdd(&cur[kk], 1u);
            atomicAdd(&wsum[kk], (unsigned long long)((unsigned)(w[t][0] >> 32) & 0x7fffffffu));
            const ulonglong2 o = rater_record<WIDE>(w[t], uptr);
            if (in_lds) lrec[q] = o; else rc[h.pos0 + q] = o;
        }
    }
    __syncthreads();
    if (in_lds)
        for (int q = threadIdx.x; q < h.n; q += ts::CT) rc[h.pos0 + q] = lrec[q];
    for (int x = threadIdx.x; x < h.nk; x += ts::CT) Wp[h.k0 + x] = wsum[x];
}

// the slices of the large keys: their records sit in their final range already (any order inside a key is a result)
template <bool WIDE>
__global__ __launch_bounds__(ts::LT) void k_rc_large(ts::Geo G, const unsigned long long *bufB, const long long *uptr,
                                                     ulonglong2 *rc, unsigned long long *Wp) {
    constexpr int RW = WIDE ? 3 : 2;
    if (blockIdx.x >= G.counters[1]) return;
    const int2 sl = G.slist[blockIdx.x];
    const long long lo = G.ptr[sl.x] + (long long)sl.y * ts::SL;
    const long long hi = min(G.ptr[sl.x + 1], lo + ts::SL);
    unsigned long long sum = 0ull;
    for (long long p = lo + threadIdx.x; p < hi; p += ts::LT) {
        unsigned long long w[RW];
#pragma unroll
        for (int x = 0; x < RW; x++) w[x] = bufB[(size_t)p * RW + x];
        sum += (unsigned long long)((unsigned)(w[0] >> 32) & 0x7fffffffu);
        rc[p] = rater_record<WIDE>(w, uptr);
    }
    sum = (unsigned long long)wave_sum_ll((long long)sum);
    if (lane_id() == 0 && sum) atomicAdd(&Wp[sl.x], sum);
}

// item statistics from the rater records (item_stats.h); the records' mutuality flags are set once the average is known
struct RcSrc {
    RaterRec *rc; const double *u_avg;
    static constexpr bool has_flags = true;
    __device__ __forceinline__ void load(long long p, double &r, int &u) const { const RaterRec x = rc[p]; r = (double)x.rating; u = x.user; }
    __device__ __forceinline__ double uavg(int u) const { return u_avg[u]; }
    __device__ __forceinline__ void set_flag(long long p, bool ge) const { if (ge) rc[p].pos_ge |= (int)0x80000000u; }
};
struct RcWideSrc {
    const RaterRecWide *rc;
    static constexpr bool has_flags = false;       // RecommenderSim has no mutuality
    __device__ __forceinline__ void load(long long p, double &r, int &u) const { r = rc[p].rating; u = 0; }
    __device__ __forceinline__ double uavg(int) const { return 0.0; }
    __device__ __forceinline__ void set_flag(long long, bool) const {}
};

// Items with more than STAT_BIG raters (up to 1e5 at BASELINE configs[1]: one wave walking them was the kernel's tail, and
// the unrolled walk they need cost every wave of the kernel its registers) are
// cut into chunks of STAT_CHK raters: k_item_stats3 lists them, k_item_chunks sums every chunk on a wave of its own,
// k_item_big adds an item's chunk sums up in chunk order (the adjusted norm exactly) and finishes it, k_item_big_flags sets
// the flags of its rater records.
constexpr int STAT_BIG = 512;
constexpr int STAT_CHK = 2048;
struct BigList {
    unsigned *counters;       // [0] chunks listed, [1] big items listed
    int2 *chunks;             // (item, chunk)
    int4 *items;              // (item, first chunk, chunks, -)
    double *part;             // [chunk][5]
    long long chunk_cap, item_cap;
};

template <typename Src>
__global__ __launch_bounds__(256) void k_item_stats3(int I, int lo, int hi, const long long *iptr, const Src src, double *info,
                                                     double *norms, BigList B) {
    const int i0 = lo + (blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6))) * 4;
    if (i0 >= hi) return;
    const int lane = lane_id();
    {
        const int i = i0 + (lane >> 4);
        const bool on = i < hi && iptr[i + 1] - iptr[i] <= 64;
        item_stats_group<16, Src, false>(on, i, lane & 15, I, iptr, src, info, norms, nullptr, nullptr);
    }
    for (int t = 0; t < 4; t++) {
        const int i = i0 + t;
        if (i >= hi) break;
        const long long n = iptr[i + 1] - iptr[i];
        if (n <= 64) continue;
        if (n > STAT_BIG) {
            if (lane == 0) {
                const int nch = (int)((n + STAT_CHK - 1) / STAT_CHK);
                const unsigned base = atomicAdd(&B.counters[0], (unsigned)nch);
                const unsigned slot = atomicAdd(&B.counters[1], 1u);
                if ((long long)slot < B.item_cap) B.items[slot] = make_int4(i, (int)base, nch, 0);
                for (int x = 0; x < nch; x++)
                    if ((long long)base + x < B.chunk_cap) B.chunks[base + x] = make_int2(i, x);
            }
            continue;
        }
        item_stats_group<64, Src, false>(true, i, lane, I, iptr, src, info, norms, nullptr, nullptr);
    }
}

// one wave per listed chunk: the item's partial sums over raters [c STAT_CHK, (c + 1) STAT_CHK)
template <typename Src>
__global__ __launch_bounds__(256) void k_item_chunks(const long long *iptr, const Src src, BigList B) {
    const unsigned c = blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6));
    if (c >= B.counters[0]) return;
    const int2 d = B.chunks[c];
    const int lane = lane_id();
    const long long p0 = iptr[d.x] + (long long)d.y * STAT_CHK;
    const long long p1 = min(iptr[d.x + 1], p0 + STAT_CHK);
    constexpr int UN = 8;
    double s = 0.0, q = 0.0, a2 = 0.0, a2lo = 0.0;
    for (long long p = p0 + lane; p < p1; p += 64 * UN) {
        double rr[UN], av[UN];
        int uu[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            rr[t] = 0.0; uu[t] = -1;
            if (p + 64 * t < p1) src.load(p + 64 * t, rr[t], uu[t]);
        }
#pragma unroll
        for (int t = 0; t < UN; t++) av[t] = uu[t] >= 0 ? src.uavg(uu[t]) : 0.0;
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if (uu[t] < 0) continue;
            const double dlt = rr[t] - av[t];
            s += rr[t];
            q += rr[t] * rr[t];
            dd_add(a2, a2lo, dlt * dlt);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); q += __shfl_xor(q, m, 64); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double oh = __shfl_down(a2, m, 64), ol = __shfl_down(a2lo, m, 64);
        dd_add(a2, a2lo, oh);
        dd_add(a2, a2lo, ol);
    }
    if (lane == 0) {
        double *o = B.part + (size_t)c * 5;
        o[0] = s; o[1] = q; o[2] = a2; o[3] = a2lo; o[4] = (double)(p1 - p0);
    }
}

__global__ __launch_bounds__(64) void k_item_big(int I, BigList B, double *info, double *norms) {
    const unsigned b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B.counters[1]) return;
    const int4 d = B.items[b];
    double s = 0.0, q = 0.0, a2 = 0.0, a2lo = 0.0, n = 0.0;
    for (int x = 0; x < d.z; x++) {
        const double *o = B.part + (size_t)(d.y + x) * 5;
        s += o[0]; q += o[1]; n += o[4];
        dd_add(a2, a2lo, o[2]);
        dd_add(a2, a2lo, o[3]);
    }
    const int i = d.x;
    info[(size_t)i * 4 + 0] = (n > 0.0) ? 1.0 * s / n : 0.0;
    info[(size_t)i * 4 + 1] = sqrt(q);
    info[(size_t)i * 4 + 2] = sqrt(a2);
    info[(size_t)i * 4 + 3] = 1.0 * n;
    norms[i] = sqrt(q);
    norms[(size_t)I + i] = sqrt(a2);
}

template <typename Src>
__global__ __launch_bounds__(256) void k_item_big_flags(const long long *iptr, const Src src, BigList B, const double *info) {
    const unsigned c = blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6));
    if (c >= B.counters[0]) return;
    const int2 d = B.chunks[c];
    const double avg = info[(size_t)d.x * 4];
    const long long p0 = iptr[d.x] + (long long)d.y * STAT_CHK;
    const long long p1 = min(iptr[d.x + 1], p0 + STAT_CHK);
    for (long long p = p0 + lane_id(); p < p1; p += 64) {
        double r; int u;
        src.load(p, r, u);
        src.set_flag(p, r >= avg);
    }
}

// the rater records' flags from the complete item info (sharded item statistics: a rank's k_item_stats3 flagged the records
// of ITS items only; after the all-gather of the item info every record is done here -- idempotent).  A record knows its
// profile entry (e0 + position), the entry knows its item.
__global__ __launch_bounds__(256) void k_rc_flags(long long nnz, RaterRec *rc, const int2 *ub, const double *info) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const RaterRec r = rc[p];
    const int it = ub[(long long)r.e0 + (r.pos_ge & 0x7fffffff)].x & 0x7fffffff;
    if ((double)r.rating >= info[(size_t)it * 4]) rc[p].pos_ge = (int)((unsigned)r.pos_ge | 0x80000000u);
}

// the profile copy's flags: rating >= average of the entry's item
__global__ __launch_bounds__(256) void k_ub_flags(long long nnz, int2 *ub, const double *info) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int2 v = ub[e];
    if ((double)__int_as_float(v.y) >= info[(size_t)v.x * 4]) ub[e].x = (int)((unsigned)v.x | 0x80000000u);
}

// ---- the mirror (round 3): own half written in runs, mirrored half through the tile sort --------------------------------------
// Row i of the CSR = [the pairs row i computed itself (own[i]) | the pairs computed in lighter rows (mir[i])].
__global__ __launch_bounds__(256) void k_row_totals(int I, const int *own, const int *mir, int *tot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < I) tot[i] = own[i] + mir[i];
}

// level-A loader of the mirror: the half COO (SoA, cut into shards that are filled from their start) as 24-byte records
// keyed by the heavier item; its chunks are listed from the shard cursors (k_coo_chunks), so every slot of a chunk is a
// record.  The same workgroup writes the OWN half of its chunk (extra): the records of one unit are contiguous in the
// COO and share the lighter item i -- one cursor bump per run, coalesced writes, the chunk still in the caches.
// AUX: a sixth COO column travels along (RecommenderSim: the pair's local sensitivity; 32-byte records), and a row may pair
// with itself -- such a record has an own entry and no mirrored one (k_ts_bin does not route it)
template <bool AUX, bool SHARE>      // SHARE: only the rows [row_lo, row_hi) are built
struct CooLoaderT {
    static constexpr int RW = AUX ? 4 : 3;
    const int *__restrict__ coo_i; const int *__restrict__ coo_j; const double *__restrict__ coo_sim;
    const int *__restrict__ coo_mutu; const int *__restrict__ coo_nij; const double *__restrict__ coo_aux;
    const longlong2 *chunks; const unsigned *n_chunks;
    const long long *row_ptr; int *fill; int *col; double *sim; int *mutu; int *nij; double *aux;
    int row_lo, row_hi;       // the rows this call builds (an item-sharded rank: its share; the counts outside it are zero)
    __device__ __forceinline__ bool chunk(long long, long long &i0, long long &i1) const {
        if (blockIdx.x >= *n_chunks) return false;
        const longlong2 c = chunks[blockIdx.x];
        i0 = c.x; i1 = c.x + c.y;
        return true;
    }
    __device__ __forceinline__ void load(long long idx, unsigned long long (&w)[RW]) const {
        w[0] = (unsigned long long)(unsigned)coo_j[idx] | ((unsigned long long)(unsigned)coo_i[idx] << 32);
        w[1] = (unsigned long long)__double_as_longlong(coo_sim[idx]);
        w[2] = (unsigned long long)(unsigned)coo_mutu[idx] | ((unsigned long long)(unsigned)coo_nij[idx] << 32);
        if (AUX) w[RW - 1] = (unsigned long long)__double_as_longlong(coo_aux[idx]);
    }
    __device__ __forceinline__ bool keep(const unsigned long long (&w)[RW]) const {
        const int j = (int)(unsigned)w[0];
        return (!AUX || j != (int)(w[0] >> 32)) && (!SHARE || (j >= row_lo && j < row_hi));
    }
    // own half of the chunk: for a fixed r the lanes of a wave hold consecutive COO slots
    __device__ __forceinline__ void extra(long long, const unsigned long long (&w)[ts::Chunk<RW>::EPT][RW],
                                          const bool (&on)[ts::Chunk<RW>::EPT]) const {
        constexpr int E = ts::Chunk<RW>::EPT;
        const int lane = lane_id();
        int lead[E], base[E];
        long long rp[E];
#pragma unroll
        for (int r = 0; r < E; r++) {
            const int iw = (int)(w[r][0] >> 32);
            const bool mine = on[r] && (!SHARE || (iw >= row_lo && iw < row_hi));
            const int iu = mine ? iw : -1 - lane;                           // inactive lanes: unique fake rows
            const int prev = __shfl_up(iu, 1, 64);
            const bool leader = (lane == 0) || (prev != iu);
            const unsigned long long lm = __ballot(leader);
            const unsigned long long below = lm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            lead[r] = 63 - __clzll((long long)below);
            const unsigned long long above = (lane == 63) ? 0ull : (lm >> (lane + 1));
            const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            base[r] = 0; rp[r] = -1;
            if (leader && mine) base[r] = atomicAdd(&fill[iu], next - lane);
            if (mine) rp[r] = row_ptr[iu];
        }
#pragma unroll
        for (int r = 0; r < E; r++) {
            const int bs = __shfl(base[r], lead[r], 64);
            if (rp[r] >= 0) {
                const long long a = rp[r] + bs + (lane - lead[r]);
                col[a] = (int)(unsigned)w[r][0]; sim[a] = __longlong_as_double((long long)w[r][1]);
                mutu[a] = (int)(unsigned)w[r][2]; nij[a] = (int)(w[r][2] >> 32);
                if (AUX) aux[a] = __longlong_as_double((long long)w[r][RW - 1]);
            }
        }
    }
};

// chunks of CH records of the COO's shards: shard s holds its records in slots [s shard_cap, s shard_cap + fill[s])
// (cur == NULL: one range of n_fill records)
__global__ __launch_bounds__(256) void k_coo_chunks(int n_shards, long long shard_cap, const unsigned long long *cur, long long n_fill,
                                                    longlong2 *chunks, unsigned *n_chunks, long long cap) {
    constexpr int CH = ts::Chunk<3>::CH;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_shards) return;
    long long f = cur ? (long long)cur[s] : n_fill;
    if (f > shard_cap) f = shard_cap;
    const int nch = (int)((f + CH - 1) / CH);
    if (nch == 0) return;
    const unsigned base = atomicAdd(n_chunks, (unsigned)nch);
    for (int x = 0; x < nch; x++)
        if ((long long)base + x < cap) {
            longlong2 c;
            c.x = (long long)s * shard_cap + (long long)x * CH;
            c.y = min((long long)CH, f - (long long)x * CH);
            chunks[base + x] = c;
        }
}

// level C of the mirror: the small keys of a tile, laid out as CSR columns in LDS, leave as whole row segments
template <bool AUX>
__global__ __launch_bounds__(ts::CT) void k_mir_tiles(ts::Geo G, const unsigned long long *bufB, const long long *row_ptr,
                                                      const int *own, int *col, double *sim, int *mutu, int *nij, double *aux) {
    constexpr int RW = AUX ? 4 : 3;
    constexpr int CAPX = AUX ? 2048 : ts::CAP;      // (with the sixth column: 68 KB of LDS, two workgroups per CU still)
    __shared__ unsigned cur[ts::NK_MAX], kst[ts::NK_MAX];
    __shared__ long long gsh[ts::NK_MAX];
    __shared__ double lsim[CAPX];
    __shared__ double laux[AUX ? CAPX : 1];
    __shared__ int lcol[CAPX], lmutu[CAPX], lnij[CAPX];
    __shared__ unsigned short lkk[CAPX];
    const ts::TileHead h = ts::tile_head(G, blockIdx.x);
    if (h.nk <= 0 || h.n <= 0) return;
    for (int x = threadIdx.x; x < h.nk; x += ts::CT) {
        const int k = h.k0 + x;
        cur[x] = 0u;
        kst[x] = (unsigned)(G.ptr[k] - h.pos0);
        gsh[x] = row_ptr[k] + own[k] - G.ptr[k];        // mirrored position -> CSR position of key k
    }
    __syncthreads();
    const bool in_lds = h.n <= CAPX;
    constexpr int UN = 4;
    for (int base = 0; base < h.n; base += ts::CT * UN) {
        unsigned long long w[UN][RW];
        bool on[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            const int idx = base + t * ts::CT + threadIdx.x;
            on[t] = idx < h.n;
            const size_t o = (size_t)(h.pos0 + (on[t] ? idx : 0)) * RW;
#pragma unroll
            for (int x = 0; x < RW; x++) w[t][x] = bufB[o + x];
        }
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if (!on[t]) continue;
            const int kk = (int)((unsigned)w[t][0]) - h.k0;
            const unsigned q = kst[kk] + atomicAdd(&cur[kk], 1u);
            const int ci = (int)(w[t][0] >> 32), cm = (int)(unsigned)w[t][2], cn = (int)(w[t][2] >> 32);
            const double cs = __longlong_as_double((long long)w[t][1]);
            const double ca = AUX ? __longlong_as_double((long long)w[t][RW - 1]) : 0.0;
            if (in_lds) {
                lcol[q] = ci; lsim[q] = cs; lmutu[q] = cm; lnij[q] = cn; lkk[q] = (unsigned short)kk;
                if (AUX) laux[q] = ca;
            } else {
                const long long P = h.pos0 + q + gsh[kk];
                col[P] = ci; sim[P] = cs; mutu[P] = cm; nij[P] = cn;
                if (AUX) aux[P] = ca;
            }
        }
    }
    if (!in_lds) return;
    __syncthreads();
    for (int q = threadIdx.x; q < h.n; q += ts::CT) {
        const long long P = h.pos0 + q + gsh[lkk[q]];
        col[P] = lcol[q]; sim[P] = lsim[q]; mutu[P] = lmutu[q]; nij[P] = lnij[q];
        if (AUX) aux[P] = laux[q];
    }
}

template <bool AUX>
__global__ __launch_bounds__(ts::LT) void k_mir_large(ts::Geo G, const unsigned long long *bufB, const long long *row_ptr,
                                                      const int *own, int *col, double *sim, int *mutu, int *nij, double *aux) {
    constexpr int RW = AUX ? 4 : 3;
    if (blockIdx.x >= G.counters[1]) return;
    const int2 sl = G.slist[blockIdx.x];
    const int k = sl.x;
    const long long lo = G.ptr[k] + (long long)sl.y * ts::SL;
    const long long hi = min(G.ptr[k + 1], lo + ts::SL);
    const long long sh = row_ptr[k] + own[k] - G.ptr[k];
    for (long long p = lo + threadIdx.x; p < hi; p += ts::LT) {
        const unsigned long long w0 = bufB[(size_t)p * RW], w1 = bufB[(size_t)p * RW + 1], w2 = bufB[(size_t)p * RW + 2];
        const long long P = p + sh;
        col[P] = (int)(w0 >> 32); sim[P] = __longlong_as_double((long long)w1);
        mutu[P] = (int)(unsigned)w2; nij[P] = (int)(w2 >> 32);
        if (AUX) aux[P] = __longlong_as_double((long long)bufB[(size_t)p * RW + 3]);
    }
}

}  // namespace xmap

using namespace xmap;

// forked streams for the per-class pair kernels (created once per process and device; never destroyed)
struct SideStreams { hipStream_t s[N_CLASSES + 1]; hipEvent_t fork, done[N_CLASSES + 1]; int dev; };   // + one for the heavy rows
static SideStreams *side_streams() {
    static thread_local SideStreams *cur[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("hipGetDevice failed"); return nullptr; }
    if (cur[dev]) return cur[dev];
    SideStreams *p = new SideStreams();
    p->dev = dev;
    bool ok = hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) == hipSuccess;
    for (int c = 0; c <= N_CLASSES && ok; c++)
        ok = hipStreamCreateWithFlags(&p->s[c], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&p->done[c], hipEventDisableTiming) == hipSuccess;
    if (!ok) { set_error("could not create the side streams"); delete p; return nullptr; }
    cur[dev] = p;
    return p;
}


// ---- tile sort: host side ---------------------------------------------------------------------------------------------
namespace {
// geometry for K keys and M records (tilesort.h): tile measure 2^ts_log, key weight KW, T = NA * NB tiles
void ts_geometry(int K, long long M, int ch, ts::Geo &G) {
    G.K = K; G.M = M; G.ch = ch;
    G.ts_log = 11;
    for (;;) {
        const long long TS = 1ll << G.ts_log;
        long long kw = M / (4 * (long long)(K > 0 ? K : 1));
        const long long kw_min = TS / (ts::NK_MAX - 2) + 1;          // keys of a tile <= TS / KW + 1 <= NK_MAX
        if (kw < kw_min) kw = kw_min;
        if (kw < 4) kw = 4;
        const long long tiles = ((M + (long long)K * kw) >> G.ts_log) + 1;
        if (tiles <= (long long)ts::NA_MAX * ts::NB || G.ts_log >= 30) {
            G.KW = (int)kw;
            G.NA = (int)((tiles + ts::NB - 1) / ts::NB);
            if (G.NA < 1) G.NA = 1;
            if (G.NA > ts::NA_MAX) G.NA = ts::NA_MAX;
            G.T = G.NA * ts::NB;
            return;
        }
        G.ts_log++;
    }
}

// tables of one sort (arena temporaries of the calling entry point) + plan and chunk kernels
int ts_prepare(hipStream_t st, ts::Geo &G, const long long *ptr) {
    G.ptr = ptr;
    G.clist_cap = G.M / G.ch + G.NA + 1;
    G.slist_cap = G.M / ts::SL + (G.M >> G.ts_log) + 2;
    XM_HIP(xm_malloc_async((void **)&G.tk, sizeof(unsigned) * (size_t)(G.K > 0 ? G.K : 1), st));
    XM_HIP(xm_malloc_async((void **)&G.tile_key0, sizeof(int) * ((size_t)G.T + 1), st));
    XM_HIP(xm_malloc_async((void **)&G.tile_pos0, sizeof(long long) * ((size_t)G.T + 1), st));
    XM_HIP(xm_malloc_async((void **)&G.tile_large, sizeof(int) * (size_t)G.T, st));
    // cursors and counters in one zeroed block: curA [NA], curB [2 T], counters [2 x 4 B]
    unsigned long long *z = nullptr;
    const size_t nz = (size_t)G.NA + 2 * (size_t)G.T + 1;
    XM_HIP(xm_malloc_async((void **)&z, sizeof(unsigned long long) * nz, st));
    G.curA = z; G.curB = z + G.NA; G.counters = (unsigned *)(z + G.NA + 2 * (size_t)G.T);
    XM_HIP(xm_malloc_async((void **)&G.clist, sizeof(int2) * (size_t)G.clist_cap, st));
    XM_HIP(xm_malloc_async((void **)&G.slist, sizeof(int2) * (size_t)G.slist_cap, st));
    XM_HIP(hipMemsetAsync(z, 0, sizeof(unsigned long long) * nz, st));
    XM_HIP(hipMemsetAsync(G.tile_large, 0xff, sizeof(int) * (size_t)G.T, st));
    ts::k_ts_plan<<<dim3((unsigned)((G.K + 255) / 256)), dim3(256), 0, st>>>(G);
    XM_LAUNCH_CHECK();
    ts::k_ts_chunks<<<dim3((unsigned)((G.NA + 255) / 256)), dim3(256), 0, st>>>(G);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}  // namespace

namespace {
template <bool AUX, bool SHARE>
int mirror_levels(hipStream_t st, const ts::Geo &G, int64_t coo_cap, const int32_t *coo_i, const int32_t *coo_j, const double *coo_sim,
                  const int32_t *coo_mutu, const int32_t *coo_nij, const double *coo_aux, const longlong2 *chunks, const unsigned *n_chunks,
                  long long chunk_cap, int64_t n_pairs, const int32_t *own, const int64_t *row_ptr, int32_t *fill, void *bufA, void *bufB,
                  int32_t *col, double *sim, int32_t *mutu, int32_t *nij, double *aux, int row_lo, int row_hi) {
    constexpr int RW = AUX ? 4 : 3;
    CooLoaderT<AUX, SHARE> LA{coo_i, coo_j, coo_sim, coo_mutu, coo_nij, coo_aux, chunks, n_chunks, (const long long *)row_ptr, fill,
                              col, sim, mutu, nij, aux, row_lo, row_hi};
    ts::RecLoader<RW> LB{(const unsigned long long *)bufA};
    ts::k_ts_bin<RW, false, CooLoaderT<AUX, SHARE>><<<dim3((unsigned)chunk_cap), dim3(ts::BT), 0, st>>>(G, LA, coo_cap, (unsigned long long *)bufA);
    XM_LAUNCH_CHECK();
    ts::k_ts_bin<RW, true, ts::RecLoader<RW>><<<dim3((unsigned)G.clist_cap), dim3(ts::BT), 0, st>>>(G, LB, n_pairs, (unsigned long long *)bufB);
    XM_LAUNCH_CHECK();
    k_mir_tiles<AUX><<<dim3((unsigned)G.T), dim3(ts::CT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)row_ptr, own, col,
                                                                   sim, mutu, nij, aux);
    XM_LAUNCH_CHECK();
    k_mir_large<AUX><<<dim3((unsigned)G.slist_cap), dim3(ts::LT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)row_ptr, own,
                                                                            col, sim, mutu, nij, aux);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}  // namespace

namespace {
// counts[j] += entries of the half COO whose partner (second index) is j -- the mirrored entries row j will get; self pairs
// (skip_self) have none.  n_ranges ranges of range_cap slots, the first cur[r] of range r valid (cur == NULL: all of them).
// part: range_cap * n_ranges ints of scratch.  Three passes (k_cbs_hist, k_cbs_scatter, k_cb_count), no atomic per entry.
int mirror_counts(hipStream_t st, int n_items, long long range_cap, int n_ranges, const unsigned long long *cur, const int *coo_i,
                  const int *coo_j, bool skip_self, int *part, int *counts) {
    if (n_items <= 0 || range_cap <= 0 || n_ranges <= 0) return XMAP_OK;
    int sh = 0;
    while (((long long)(n_items - 1) >> sh) >= CB_MAX) sh++;
    unsigned *bcnt = nullptr, *bcur = nullptr;
    long long *bptr = nullptr;
    XM_HIP(xm_malloc_async((void **)&bcnt, sizeof(unsigned) * CB_MAX, st));
    XM_HIP(xm_malloc_async((void **)&bcur, sizeof(unsigned) * CB_MAX, st));
    XM_HIP(xm_malloc_async((void **)&bptr, sizeof(long long) * (CB_MAX + 1), st));
    XM_HIP(hipMemsetAsync(bcnt, 0, sizeof(unsigned) * CB_MAX, st));
    const long long segs = (long long)n_ranges * ((range_cap + CB_CHUNK - 1) / CB_CHUNK);
    const dim3 g((unsigned)((segs + CB_SPB - 1) / CB_SPB));
    if (skip_self) k_cbs_hist<true><<<g, dim3(CB_T), 0, st>>>(range_cap, n_ranges, cur, coo_i, coo_j, sh, bcnt);
    else k_cbs_hist<false><<<g, dim3(CB_T), 0, st>>>(range_cap, n_ranges, cur, coo_i, coo_j, sh, bcnt);
    XM_LAUNCH_CHECK();
    k_cb_scan<<<dim3(1), dim3(CB_MAX), 0, st>>>(bcnt, bptr, bcur);
    XM_LAUNCH_CHECK();
    if (skip_self) k_cbs_scatter<true><<<g, dim3(CB_T), 0, st>>>(range_cap, n_ranges, cur, coo_i, coo_j, sh, bptr, bcur, part);
    else k_cbs_scatter<false><<<g, dim3(CB_T), 0, st>>>(range_cap, n_ranges, cur, coo_i, coo_j, sh, bptr, bcur, part);
    XM_LAUNCH_CHECK();
    const long long cap = range_cap * n_ranges;
    k_cb_count<<<dim3((unsigned)((cap + (long long)CB_CHUNK * CB_SPB - 1) / ((long long)CB_CHUNK * CB_SPB))), dim3(CB_T), 0, st>>>(
        -1, part, sh, bptr, n_items, counts);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}  // namespace

extern "C" {
#ifdef A_TRACE
int xmap_debug_astamp(unsigned int *host, long long n_units) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xmap::g_astamp), (size_t)n_units * 16);
}
int xmap_debug_atrace(unsigned long long *host, long long n_units) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xmap::g_atrace), (size_t)n_units * 16);
}
#endif

int xmap_sim2_layout(void *stream, const xmap_ratings *R, const double *info, int32_t ch_min, int32_t *hist /*[U+2]*/,
                     int64_t *pre /*[U+3]*/, int32_t *ctl /*[4]: CH, n_heavy*/, int32_t *hid, int32_t *hlist /*[1024]*/,
                     uint64_t *ub_key /*[nnz] scratch*/, void *ub /*[nnz] 8 B*/, void *rc /*[nnz] 16 B*/,
                     uint64_t *Wp /*[I] out*/, int32_t dups, int32_t *h_ctl /*[2]*/) {
    XM_ARG(R && info && hist && pre && ctl && hid && hlist && ub_key && ub && rc && Wp && ch_min >= 64);
    XM_ARG(R->nnz < 0x7fffffffLL && R->n_users < 0x7ffffff0LL);
    hipStream_t st = (hipStream_t)stream;
    const int I = R->n_items;
    const int HB = (int)R->n_users + 2;
    XM_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)HB, st));
    XM_HIP(hipMemsetAsync(ctl, 0x7f, sizeof(int32_t), st));          // CH = 0x7f7f7f7f: "no heavy rows"
    XM_HIP(hipMemsetAsync(ctl + 1, 0, 3 * sizeof(int32_t), st));
    if (I > 0) {
        k_hist<<<dim3((unsigned)((I + 256 * HIST_PER - 1) / (256 * HIST_PER))), dim3(256), 0, st>>>(
            I, (const long long *)R->item_ptr, HB, hist);
        XM_LAUNCH_CHECK();
    }
    int rcode = xmap_exclusive_scan_i32_to_i64(stream, hist, pre, HB, nullptr);
    if (rcode) return rcode;
    if (HB - 1 > ch_min) {
        k_threshold<<<dim3((unsigned)((HB + 255) / 256)), dim3(256), 0, st>>>(I, HB, (const long long *)pre, ch_min, ctl);
        XM_LAUNCH_CHECK();
    }
    if (I > 0) {
        k_mark_heavy<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, (const long long *)R->item_ptr, ctl, hid,
                                                                                hlist, ctl + 1);
        XM_LAUNCH_CHECK();
    }
    if (R->n_users > 0) {
        k_sort_profiles<<<dim3((unsigned)((R->n_users + 15) / 16)), dim3(256), 0, st>>>(
            R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating, (const long long *)R->item_ptr, info,
            (unsigned long long *)ub_key, (int2 *)ub);
        XM_LAUNCH_CHECK();
    }
    XM_HIP(hipMemsetAsync(Wp, 0, sizeof(uint64_t) * (size_t)(I > 0 ? I : 1), st));
    if (R->nnz > 0) {
        // profiles that may hold an item twice: the sort is done with ub_key, which then serves as the (zeroed) copy
        // counters of k_rater_records
        if (dups) XM_HIP(hipMemsetAsync(ub_key, 0, sizeof(int32_t) * (size_t)R->nnz, st));
        k_rater_records<<<dim3((unsigned)((R->nnz + 255) / 256)), dim3(256), 0, st>>>(
            I, R->nnz, (const long long *)R->item_ptr, R->item_user, (const long long *)R->user_ptr, (const int2 *)ub,
            (RaterRec *)rc, dups ? (int *)ub_key : nullptr, (unsigned long long *)Wp);
        XM_LAUNCH_CHECK();
    }
    if (h_ctl) {
        XM_HIP(hipMemcpyAsync(h_ctl, ctl, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (h_ctl[1] > HMAX) {
            set_error("heavy set larger than %d", HMAX);
            return XMAP_ERR_OVERFLOW;
        }
    }
    return XMAP_OK;
}

int xmap_sim2_plan(void *stream, const xmap_ratings *R, int32_t slot_target, const void *rc, const int64_t *pre,
                   const int32_t *hid, const int32_t *ctl, int32_t *Q, int32_t *C, uint8_t *small, uint64_t *Wp /*[I] out*/,
                   int32_t *Qcat /*[4 I]*/, int64_t *uq_ptr /*[4 I + 1]*/, int64_t *uc_ptr, int32_t dups,
                   int64_t *h_counts /*[8]: light units, heavy units, first unit of table class rank 0..4, light units*/) {
    XM_ARG(R && rc && Wp && pre && hid && ctl && Q && C && small && Qcat && uq_ptr && uc_ptr && h_counts);
    XM_ARG(slot_target > 0 && slot_target <= T_SLOTS);
    hipStream_t st = (hipStream_t)stream;
    const int I = R->n_items;
    XM_HIP(hipMemsetAsync(Qcat, 0, sizeof(int32_t) * (size_t)N_CLASSES * (size_t)(I > 0 ? I : 1), st));
    if (I > 0) {
        k_plan2<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(
            I, (const long long *)R->item_ptr, (const RaterRec *)rc, (const long long *)pre, (int)R->n_users + 2, hid, ctl,
            slot_target, dups, Q, C, small, (unsigned long long *)Wp, Qcat);
        XM_LAUNCH_CHECK();
    }
    int rcode = xmap_exclusive_scan_i32_to_i64(stream, Qcat, uq_ptr, (int64_t)N_CLASSES * I, &h_counts[0]);
    if (rcode) return rcode;
    rcode = xmap_exclusive_scan_i32_to_i64(stream, C, uc_ptr, I, &h_counts[1]);
    if (rcode) return rcode;
    // class boundaries uq_ptr[c I], c = 0..4: one strided copy (the scan above has synchronised the stream)
    if (I > 0)
        XM_HIP(hipMemcpy2DAsync(&h_counts[2], sizeof(int64_t), uq_ptr, sizeof(int64_t) * (size_t)I, sizeof(int64_t),
                                N_CLASSES, hipMemcpyDeviceToHost, st));
    else
        for (int c = 0; c < N_CLASSES; c++) h_counts[2 + c] = 0;
    XM_HIP(hipStreamSynchronize(st));
    h_counts[2 + N_CLASSES] = h_counts[0];
    return XMAP_OK;
}

int xmap_sim2_units(void *stream, int32_t n_items, const int64_t *item_ptr, const int32_t *Qcat, const int64_t *uq_ptr,
                    int32_t *uq_item, int32_t *uq_q, const int32_t *C, const int64_t *uc_ptr, int32_t *uc_item, int32_t *uc_c) {
    XM_ARG(item_ptr && Qcat && uq_ptr && uq_item && uq_q && C && uc_ptr && uc_item && uc_c);
    if (n_items == 0) return XMAP_OK;
    k_fill_units2<<<dim3((unsigned)(((long long)N_CLASSES * n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, (const long long *)item_ptr, Qcat, (const long long *)uq_ptr, uq_item, uq_q, C, (const long long *)uc_ptr, uc_item, uc_c, 0x7fffffffffffffffLL,
        0x7fffffffffffffffLL);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

/* xmap_sim2_plan + xmap_sim2_units with ONE synchronisation: the unit arrays are sized by the caller from bounds it knows
 * without asking the device (light units <= contributions / slot_target + n_items, heavy units <= nnz / ch_min + 1024), the
 * scans leave their totals on the device, and the counts the launches need come back in one copy together with the heavy
 * set's {CH, |H|}.  h_out [10] = {light units, heavy units, first unit of table class rank 0..4, light units, CH, |H|}. */
int xmap_sim3_plan(void *stream, const xmap_ratings *R, int32_t slot_target, const int64_t *pre, const int32_t *hid,
                   const int32_t *ctl, int32_t *Q, int32_t *C, uint8_t *small, uint64_t *Wp, int32_t *Qcat, int64_t *uq_ptr,
                   int64_t *uc_ptr, int32_t dups, int32_t *uq_item, int32_t *uq_q, int32_t *uc_item, int32_t *uc_c,
                   int64_t cap_light, int64_t cap_heavy, int64_t *h_out) {
    XM_SCOPE(stream);
    XM_ARG(R && Wp && pre && hid && ctl && Q && C && small && Qcat && uq_ptr && uc_ptr && h_out);
    XM_ARG(uq_item && uq_q && uc_item && uc_c && cap_light >= 0 && cap_heavy >= 0);
    XM_ARG(slot_target > 0 && slot_target <= T_SLOTS);
    hipStream_t st = (hipStream_t)stream;
    const int I = R->n_items;
    for (int c = 0; c < 10; c++) h_out[c] = 0;
    XM_HIP(hipMemsetAsync(Qcat, 0, sizeof(int32_t) * (size_t)N_CLASSES * (size_t)(I > 0 ? I : 1), st));
    if (I > 0) {
        k_plan2<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(
            I, (const long long *)R->item_ptr, nullptr, (const long long *)pre, (int)R->n_users + 2, hid, ctl, slot_target, dups, Q, C,
            small, (unsigned long long *)Wp, Qcat);
        XM_LAUNCH_CHECK();
    }
    int rcode = xmap_exclusive_scan_i32_to_i64(stream, Qcat, uq_ptr, (int64_t)N_CLASSES * I, nullptr);
    if (rcode) return rcode;
    rcode = xmap_exclusive_scan_i32_to_i64(stream, C, uc_ptr, I, nullptr);
    if (rcode) return rcode;
    if (I > 0) {
        k_fill_units2<<<dim3((unsigned)(((long long)N_CLASSES * I + 255) / 256)), dim3(256), 0, st>>>(
            I, (const long long *)R->item_ptr, Qcat, (const long long *)uq_ptr, uq_item, uq_q, C, (const long long *)uc_ptr, uc_item, uc_c,
            cap_light, cap_heavy);
        XM_LAUNCH_CHECK();
        // class boundaries uq_ptr[c I], c = 0..5 (the last one is the total), the heavy units' total, {CH, |H|}
        XM_HIP(hipMemcpy2DAsync(&h_out[2], sizeof(int64_t), uq_ptr, sizeof(int64_t) * (size_t)I, sizeof(int64_t), N_CLASSES + 1,
                                hipMemcpyDeviceToHost, st));
        XM_HIP(hipMemcpyAsync(&h_out[1], uc_ptr + I, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    int32_t h_ctl[2] = {0, 0};
    XM_HIP(hipMemcpyAsync(h_ctl, ctl, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    h_out[0] = h_out[2 + N_CLASSES];
    h_out[8] = h_ctl[0]; h_out[9] = h_ctl[1];
    if (h_ctl[1] > HMAX) {
        set_error("heavy set larger than %d", HMAX);
        return XMAP_ERR_OVERFLOW;
    }
    if (h_out[0] > cap_light || h_out[1] > cap_heavy) {
        set_error("unit arrays too small: %lld light / %lld heavy units, room for %lld / %lld", (long long)h_out[0], (long long)h_out[1],
                  (long long)cap_light, (long long)cap_heavy);
        return XMAP_ERR_CAPACITY;
    }
    return XMAP_OK;
}

int xmap_sim2_pairs(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg, const double *norms,
                    const void *rc, const void *ub, const int32_t *Q,
                    const uint8_t *small, const int32_t *uq_item, const int32_t *uq_q, const int64_t *cls_ptr /*host [6]*/,
                    int64_t unit_lo, int64_t unit_hi, const int32_t *hid,
                    const int32_t *hlist, const int32_t *ctl, const int32_t *C, const int64_t *uc_ptr,
                    const int32_t *uc_item, const int32_t *uc_c, int32_t n_heavy_units, int32_t n_heavy, int phases,
                    double *hp_hi, double *hp_lo, int32_t *hp_cnt, int32_t *hp_mut, int64_t coo_cap, int32_t *coo_i,
                    int32_t *coo_j, double *coo_sim, int32_t *coo_mutu, int32_t *coo_nij, double *coo_ls /*or NULL*/,
                    int32_t *rowcnt, int32_t *rowcnt_h /*[64][1024]*/, int64_t *d_shards /*[2][4096]*/,
                    int64_t *d_counters /*[4]*/, int32_t *mircnt /*[I] or NULL*/) {
    XM_SCOPE(stream);
    XM_ARG(R && u_avg && norms && rc && ub);
    XM_ARG(Q && small && uq_item && uq_q && cls_ptr && hid && hlist && ctl && C && uc_ptr && uc_item && uc_c);
    XM_ARG(coo_i && coo_j && coo_sim && coo_mutu && coo_nij && rowcnt && rowcnt_h && d_shards && d_counters);
    XM_ARG(coo_cap >= COO_SHARDS);
    XM_ARG(method == XMAP_COSINE || method == XMAP_ADJUST_COSINE);
    XM_ARG(n_heavy_units == 0 || !(phases & 5) || (hp_hi && hp_lo && hp_cnt && hp_mut));
    // coo_ls selects the RecommenderSim variant: exact (double-double) sums, no filter, local sensitivity; its layout
    // has no heavy rows
    const bool raw = (phases & 32) != 0;      // partial sums of a user share: coo_ls is then the error column of the dot
    XM_ARG(!coo_ls || ((raw || method == XMAP_ADJUST_COSINE) && n_heavy_units == 0));
    XM_ARG(!raw || (coo_ls && n_heavy_units == 0 && n_heavy == 0));
    hipStream_t st = (hipStream_t)stream;
    if (phases & 8) {   // reset the COO cursor / counters / row counts
        XM_HIP(hipMemsetAsync(d_counters, 0, 4 * sizeof(int64_t), st));
        XM_HIP(hipMemsetAsync(d_shards, 0, 2 * COO_SHARDS * sizeof(int64_t), st));
        if (!(phases & 128)) XM_HIP(hipMemsetAsync(coo_i, 0xff, sizeof(int32_t) * (size_t)coo_cap, st));   // -1 = unused entry (bit 128: the
                                                                    // caller reads the COO through the shard cursors only: 0.4 GB less to write)
        XM_HIP(hipMemsetAsync(rowcnt, 0, sizeof(int32_t) * (size_t)(R->n_items > 0 ? R->n_items : 1), st));
        if (mircnt) XM_HIP(hipMemsetAsync(mircnt, 0, sizeof(int32_t) * (size_t)(R->n_items > 0 ? R->n_items : 1), st));
    }
    TriArgs A;
    memset(&A, 0, sizeof(A));
    A.iptr = (const long long *)R->item_ptr; A.rc = (const RaterRec *)rc; A.ub = (const int2 *)ub;
    A.u_avg = u_avg; A.nrm = norms + (method == XMAP_COSINE ? 0 : (size_t)R->n_items); A.cap = cap;
    A.Q = Q; A.small = small; A.uq_item = uq_item; A.uq_q = uq_q; A.unit_lo = unit_lo; A.unit_hi = unit_hi;
    A.hid = hid; A.hlist = hlist; A.CH = ctl; A.uc_item = uc_item; A.uc_c = uc_c;
    A.uc_ptr = (const long long *)uc_ptr; A.C = C;
    A.hp_hi = hp_hi; A.hp_lo = hp_lo; A.hp_cnt = hp_cnt; A.hp_mut = hp_mut;
    A.shard_cap = coo_cap / COO_SHARDS; A.shard_cur = (unsigned long long *)d_shards;
    A.shard_occ = (unsigned long long *)d_shards + COO_SHARDS; A.coo_i = coo_i; A.coo_j = coo_j; A.coo_sim = coo_sim; A.coo_mutu = coo_mutu; A.coo_nij = coo_nij;
    A.coo_aux = coo_ls;
    A.raw = raw ? 1 : 0;
    A.heavy_mod = (phases >> 16) & 0xff; A.heavy_rem = (phases >> 8) & 0xff;
    if (A.heavy_mod < 1) A.heavy_mod = 1;
    A.rowcnt = rowcnt; A.mircnt = mircnt; A.rowcnt_h = rowcnt_h; A.counters = (unsigned long long *)d_counters;
    // phases 1 | 2 | 4 in one call: the heavy rows (chunk partials, then their merge) run on a side stream of their own,
    // next to the class launches of the light rows -- they share nothing but the atomic COO cursors and counters
    const bool heavy_aside = (phases & 7) == 7 && n_heavy_units > 0 && unit_hi > unit_lo;
    hipStream_t hs = st;
    SideStreams *side = nullptr;
    if (((phases & 2) && unit_hi > unit_lo) || heavy_aside) {
        side = side_streams();
        if (!side) return XMAP_ERR_HIP;
        XM_HIP(hipEventRecord(side->fork, st));
    }
    if (heavy_aside) {
        hs = side->s[N_CLASSES];
        XM_HIP(hipStreamWaitEvent(hs, side->fork, 0));
    }
    if ((phases & 1) && n_heavy_units > 0) {   // heavy rows: chunk partials
        if (method == XMAP_COSINE) k_pair_heavy<XMAP_COSINE><<<dim3((unsigned)n_heavy_units), dim3(64 * HEAVY_WAVES), 0, hs>>>(A);
        else k_pair_heavy<XMAP_ADJUST_COSINE><<<dim3((unsigned)n_heavy_units), dim3(64 * HEAVY_WAVES), 0, hs>>>(A);
        XM_LAUNCH_CHECK();
    }
    if (heavy_aside && n_heavy > 0) {
        if (method == XMAP_COSINE) k_heavy_merge<XMAP_COSINE><<<dim3((unsigned)n_heavy), dim3(256), 0, hs>>>(A, n_heavy);
        else k_heavy_merge<XMAP_ADJUST_COSINE><<<dim3((unsigned)n_heavy), dim3(256), 0, hs>>>(A, n_heavy);
        XM_LAUNCH_CHECK();
    }
    if (heavy_aside) {
        XM_HIP(hipEventRecord(side->done[N_CLASSES], hs));
        XM_HIP(hipStreamWaitEvent(st, side->done[N_CLASSES], 0));
    }
    if ((phases & 2) && unit_hi > unit_lo) {
        // one launch per table class: the class's units within [unit_lo, unit_hi).  The classes are independent (they
        // only share the COO cursors, which are atomic) and each ends in a tail of long rows at low occupancy: they
        // run side by side on forked streams and the caller's stream joins them.
        for (int c = 0; c < N_CLASSES; c++) {
            const long long lo = unit_lo > cls_ptr[c] ? unit_lo : cls_ptr[c];
            const long long hi = unit_hi < cls_ptr[c + 1] ? unit_hi : cls_ptr[c + 1];
            if (hi <= lo) continue;
            A.unit_lo = lo; A.unit_hi = hi;
            const dim3 grid((unsigned)(hi - lo));
            hipStream_t cs = side->s[c];
            XM_HIP(hipStreamWaitEvent(cs, side->fork, 0));
            if (coo_ls && !raw) {
                if (c == 0) k_pair_tri<XMAP_ADJUST_COSINE, 10, 16, true><<<grid, dim3(1024), 0, cs>>>(A);
                else if (c == 1) k_pair_tri<XMAP_ADJUST_COSINE, 10, 4, true><<<grid, dim3(256), 0, cs>>>(A);
                else if (c == 2) k_pair_tri<XMAP_ADJUST_COSINE, 9, 2, true><<<grid, dim3(128), 0, cs>>>(A);
                else if (c == 3) k_pair_tri<XMAP_ADJUST_COSINE, 8, 1, true><<<grid, dim3(64), 0, cs>>>(A);
                else k_pair_tri<XMAP_ADJUST_COSINE, 7, 1, true><<<grid, dim3(64), 0, cs>>>(A);
            } else if (method == XMAP_COSINE) {
                if (c == 0) k_pair_tri<XMAP_COSINE, 10, 16, false><<<grid, dim3(1024), 0, cs>>>(A);
                else if (c == 1) k_pair_tri<XMAP_COSINE, 10, 4, false><<<grid, dim3(256), 0, cs>>>(A);
                else if (c == 2) k_pair_tri<XMAP_COSINE, 9, 2, false><<<grid, dim3(128), 0, cs>>>(A);
                else if (c == 3) k_pair_tri<XMAP_COSINE, 8, 1, false><<<grid, dim3(64), 0, cs>>>(A);
                else k_pair_tri<XMAP_COSINE, 7, 1, false><<<grid, dim3(64), 0, cs>>>(A);
            } else {
                if (c == 0) k_pair_tri<XMAP_ADJUST_COSINE, 10, 16, false><<<grid, dim3(1024), 0, cs>>>(A);
                else if (c == 1) k_pair_tri<XMAP_ADJUST_COSINE, 10, 4, false><<<grid, dim3(256), 0, cs>>>(A);
                else if (c == 2) k_pair_tri<XMAP_ADJUST_COSINE, 9, 2, false><<<grid, dim3(128), 0, cs>>>(A);
                else if (c == 3) k_pair_tri<XMAP_ADJUST_COSINE, 8, 1, false><<<grid, dim3(64), 0, cs>>>(A);
                else k_pair_tri<XMAP_ADJUST_COSINE, 7, 1, false><<<grid, dim3(64), 0, cs>>>(A);
            }
            XM_LAUNCH_CHECK();
            XM_HIP(hipEventRecord(side->done[c], cs));
            XM_HIP(hipStreamWaitEvent(st, side->done[c], 0));
        }
        A.unit_lo = unit_lo; A.unit_hi = unit_hi;
    }
    if ((phases & 4) && !heavy_aside && n_heavy_units > 0 && n_heavy > 0) {
        if (method == XMAP_COSINE) k_heavy_merge<XMAP_COSINE><<<dim3((unsigned)n_heavy), dim3(256), 0, st>>>(A, n_heavy);
        else k_heavy_merge<XMAP_ADJUST_COSINE><<<dim3((unsigned)n_heavy), dim3(256), 0, st>>>(A, n_heavy);
        XM_LAUNCH_CHECK();
    }
    if (phases & 64) {      // d_counters is [6]: [4] = kept pairs, [5] = unordered pairs evaluated (sums over the shards)
        XM_HIP(hipMemsetAsync(d_counters + 4, 0, 2 * sizeof(int64_t), st));
        k_shard_sums<<<dim3(1), dim3(256), 0, st>>>((const unsigned long long *)d_shards, (unsigned long long *)d_counters);
        XM_LAUNCH_CHECK();
    }
    if ((phases & 16) && !raw && !mircnt && R->n_items > 0) {
        // the round-2 sequence / cross-checks (one combined count per row): the mirrored counts on top of the own ones, from
        // the partner column of the COO.  The round-3 caller (mircnt given) counts in xmap_sim3_mircount, where it has scratch.
        int *part = nullptr;
        XM_HIP(xm_malloc_async((void **)&part, sizeof(int) * (size_t)coo_cap, st));
        int rc16 = mirror_counts(st, R->n_items, coo_cap / COO_SHARDS, COO_SHARDS, (const unsigned long long *)d_shards, coo_i, coo_j,
                                 coo_ls != nullptr, part, rowcnt);
        if (rc16) return rc16;
    }
    return XMAP_OK;
}

int xmap_sim2_pack_partials(void *stream, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j, const double *coo_hi,
                            const double *coo_lo, const int32_t *coo_mutu, const int32_t *coo_nij, int64_t *rec /*[n_coo][4]*/,
                            int64_t *h_count) {
    XM_SCOPE(stream);
    XM_ARG(coo_i && coo_j && coo_hi && coo_lo && coo_mutu && coo_nij && rec && h_count && n_coo >= 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *cur = nullptr;
    XM_HIP(xm_malloc_async((void **)&cur, sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(cur, 0, sizeof(unsigned long long), st));
    if (n_coo > 0) {
        k_pack_partials<<<dim3((unsigned)((n_coo + 255) / 256)), dim3(256), 0, st>>>(n_coo, coo_i, coo_j, coo_hi, coo_lo, coo_mutu,
                                                                                     coo_nij, cur, (long long *)rec);
        XM_LAUNCH_CHECK();
    }
    XM_HIP(hipMemcpyAsync(h_count, cur, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    XM_HIP(xm_free_async(cur, st));
    return XMAP_OK;
}

int xmap_sim2_pack_pairs(void *stream, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j, const double *coo_sim,
                         const int32_t *coo_mutu, const int32_t *coo_nij, int64_t *rec /*[n_coo][3]*/, int64_t *h_count) {
    XM_SCOPE(stream);
    XM_ARG(coo_i && coo_j && coo_sim && coo_mutu && coo_nij && rec && h_count && n_coo >= 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *cur = nullptr;
    XM_HIP(xm_malloc_async((void **)&cur, sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(cur, 0, sizeof(unsigned long long), st));
    if (n_coo > 0) {
        k_pack_pairs<<<dim3((unsigned)((n_coo + 255) / 256)), dim3(256), 0, st>>>(n_coo, coo_i, coo_j, coo_sim, coo_mutu, coo_nij, cur,
                                                                                  (long long *)rec);
        XM_LAUNCH_CHECK();
    }
    XM_HIP(hipMemcpyAsync(h_count, cur, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    XM_HIP(xm_free_async(cur, st));
    return XMAP_OK;
}

int xmap_sim2_unpack_pairs(void *stream, int64_t n, const int64_t *rec /*[n][3]*/, int32_t *coo_i, int32_t *coo_j, double *coo_sim,
                           int32_t *coo_mutu, int32_t *coo_nij) {
    XM_ARG(rec && coo_i && coo_j && coo_sim && coo_mutu && coo_nij && n >= 0);
    if (n > 0) {
        k_unpack_pairs<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(n, (const long long *)rec, coo_i, coo_j,
                                                                                                 coo_sim, coo_mutu, coo_nij);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

int xmap_sim2_sort_partials(void *stream, int64_t n, const int64_t *rec, int64_t *rec_sorted, int32_t n_items, int32_t n_owners) {
    XM_SCOPE(stream);
    XM_ARG(rec && rec_sorted && n >= 0 && n < 0x7fffffffLL && n_items > 0 && n_owners >= 0 && n_owners <= 65536);
    if (n == 0) return XMAP_OK;
    int bits_b = 1;
    while ((1ll << bits_b) < (long long)n_items) bits_b++;
    int bits = 2 * bits_b;
    if (n_owners > 0) { bits = 1; while ((1 << bits) < n_owners) bits++; }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = nullptr;
    int *vals = nullptr;
    XM_HIP(xm_malloc_async((void **)&keys, sizeof(unsigned long long) * 2 * (size_t)n, st));
    XM_HIP(xm_malloc_async((void **)&vals, sizeof(int) * 2 * (size_t)n, st));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    k_partial_keys<<<grid, block, 0, st>>>(n, (const long long *)rec, bits_b, n_items, n_owners, keys, vals);
    XM_LAUNCH_CHECK();
    int rc = radix_sort_pairs(st, keys, vals, keys + n, vals + n, n, bits);
    if (rc) return rc;
    k_partial_gather<<<grid, block, 0, st>>>(n, (const long long *)rec, vals, (long long *)rec_sorted);
    XM_LAUNCH_CHECK();
    XM_HIP(xm_free_async(vals, st));
    XM_HIP(xm_free_async(keys, st));
    return XMAP_OK;
}

int xmap_sim2_merge_partials(void *stream, int method, int cap, int32_t n_items, int64_t n, const int64_t *rec_sorted,
                             const double *norms, int32_t *coo_i, int32_t *coo_j, double *coo_sim, int32_t *coo_mutu,
                             int32_t *coo_nij, int32_t *rowcnt, int64_t *h_counts /*[2]: kept, evaluated (unordered pairs)*/) {
    XM_SCOPE(stream);
    XM_ARG(rec_sorted && norms && coo_i && coo_j && coo_sim && coo_mutu && coo_nij && rowcnt && h_counts && n >= 0 && cap > 0);
    XM_ARG(method == XMAP_COSINE || method == XMAP_ADJUST_COSINE);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *cnt = nullptr;
    XM_HIP(xm_malloc_async((void **)&cnt, 2 * sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(rowcnt, 0, sizeof(int32_t) * (size_t)(n_items > 0 ? n_items : 1), st));
    if (n > 0) {
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        const double *nrm = norms + (method == XMAP_COSINE ? 0 : (size_t)n_items);
        if (method == XMAP_COSINE)
            k_merge_partials<XMAP_COSINE><<<grid, block, 0, st>>>(n, (const long long *)rec_sorted, nrm, cap, cnt, coo_i, coo_j, coo_sim,
                                                                  coo_mutu, coo_nij, rowcnt);
        else
            k_merge_partials<XMAP_ADJUST_COSINE><<<grid, block, 0, st>>>(n, (const long long *)rec_sorted, nrm, cap, cnt, coo_i, coo_j,
                                                                         coo_sim, coo_mutu, coo_nij, rowcnt);
        XM_LAUNCH_CHECK();
    }
    XM_HIP(hipMemcpyAsync(h_counts, cnt, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    XM_HIP(xm_free_async(cnt, st));
    return XMAP_OK;
}

int xmap_sim2_scatter(void *stream, int32_t n_items, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j,
                      const double *coo_sim, const int32_t *coo_mutu, const int32_t *coo_nij, const double *coo_ls /*or NULL*/,
                      const int64_t *row_ptr, int32_t *fill /*[I] scratch*/, const int32_t *hid, const int32_t *hlist,
                      int32_t *col, double *sim, int32_t *mutu, int32_t *nij, double *ls /*or NULL*/) {
    XM_ARG(coo_i && coo_j && coo_sim && coo_mutu && coo_nij && row_ptr && fill && hid && hlist && col && sim && mutu && nij);
    XM_ARG((coo_ls != nullptr) == (ls != nullptr));
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(hipMemsetAsync(fill, 0, sizeof(int32_t) * (size_t)(n_items > 0 ? n_items : 1), st));
    if (n_coo > 0) {
        // the kernel is bound by its partial-sector writes (PMC: 7.2 GB moved for 1.7 GB), not by the cursor atomics:
        // an LDS histogram bumping the heavy items' cursors once per workgroup was slower (2.9 vs 2.4 ms), 64
        // replicated cursors per heavy item changed nothing
        k_scatter<<<dim3((unsigned)((n_coo + 256 * SC_U - 1) / (256 * SC_U))), dim3(256), 0, st>>>(
            n_coo, coo_i, coo_j, coo_sim, coo_mutu, coo_nij, coo_ls, (const long long *)row_ptr, fill, col, sim, mutu, nij, ls);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

/* Round-3 layout of stage A (one transposition per pass): see the declarations in include/xmap_hip.h. */
int xmap_sim3_layout(void *stream, const xmap_ratings *R, int64_t *item_ptr, const double *rating64, int32_t ch_min, int32_t phases,
                     int32_t stats_lo, int32_t stats_hi, int32_t *cnt, double *u_avg, double *u_norm, int32_t *hist, int64_t *pre, int32_t *ctl, int32_t *hid, int32_t *hlist,
                     uint64_t *ub_key, void *ub, void *srec, void *bufA, void *bufB, void *rc, uint64_t *Wp, double *info,
                     double *norms, int32_t *h_ctl) {
    XM_SCOPE(stream);
    XM_ARG(R && item_ptr && (const int64_t *)item_ptr == R->item_ptr && cnt && u_avg && hist && pre && ctl && hid && hlist);
    XM_ARG(ub_key && ub && srec && bufA && bufB && rc && Wp && info && norms && ch_min >= 64);
    XM_ARG(R->nnz < 0x7fffffffLL && R->n_users < 0x7ffffff0LL && R->n_items >= 0);
    XM_ARG(rating64 || u_norm);
    XM_ARG(stats_lo >= 0 && stats_lo <= stats_hi && stats_hi <= R->n_items && (phases & ~15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int I = R->n_items;
    const long long nnz = R->nnz;
    const bool wide = rating64 != nullptr;
    const size_t In = (size_t)(I > 0 ? I : 1);
    int rcode = XMAP_OK;
    if (phases & 1) {
    // raters per item -> item_ptr
    XM_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * In, st));
    const char *cb_env = getenv("XMAP_COUNT_PART_MIN");          // (tests force the partitioned count on small inputs)
    const long long cb_min = cb_env ? atoll(cb_env) : 2000000ll;
    if (nnz >= cb_min && I > 0) {         // partitioned count (k_cbs_*, k_cb_count): the item column as one range, through bufA
        rcode = mirror_counts(st, I, nnz, 1, nullptr, nullptr, R->user_item, false, (int *)bufA, cnt);        // (free until the tile sort)
        if (rcode) return rcode;
    } else if (nnz > 0) {
        k_count3<<<dim3((unsigned)((nnz + CNT_CHUNK - 1) / CNT_CHUNK)), dim3(256), 0, st>>>(nnz, R->user_item, cnt);
        XM_LAUNCH_CHECK();
    }
    rcode = xmap_exclusive_scan_i32_to_i64(stream, cnt, item_ptr, I, nullptr);
    if (rcode) return rcode;
    if (!wide) {
        rcode = xmap_user_stats(stream, R, u_avg, u_norm);
        if (rcode) return rcode;
    }
    // rater-count histogram -> partner bounds, heavy set (as xmap_sim2_layout)
    const int HB = (int)R->n_users + 2;
    XM_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)HB, st));
    XM_HIP(hipMemsetAsync(ctl, 0x7f, sizeof(int32_t), st));          // CH = 0x7f7f7f7f: "no heavy rows"
    XM_HIP(hipMemsetAsync(ctl + 1, 0, 3 * sizeof(int32_t), st));
    if (I > 0) {
        k_hist<<<dim3((unsigned)((I + 256 * HIST_PER - 1) / (256 * HIST_PER))), dim3(256), 0, st>>>(I, (const long long *)item_ptr, HB, hist);
        XM_LAUNCH_CHECK();
    }
    rcode = xmap_exclusive_scan_i32_to_i64(stream, hist, pre, HB, nullptr);
    if (rcode) return rcode;
    if (HB - 1 > ch_min) {
        k_threshold<<<dim3((unsigned)((HB + 255) / 256)), dim3(256), 0, st>>>(I, HB, (const long long *)pre, ch_min, ctl);
        XM_LAUNCH_CHECK();
    }
    if (I > 0) {
        k_mark_heavy<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, (const long long *)item_ptr, ctl, hid, hlist, ctl + 1);
        XM_LAUNCH_CHECK();
    }
    // sorted profiles + sort records (no mutuality flags yet)
    if (R->n_users > 0 && nnz > 0) {
        const dim3 grid((unsigned)((R->n_users + 15) / 16));
        if (wide)
            k_sort_profiles3<true><<<grid, dim3(256), 0, st>>>(R->n_users, (const long long *)R->user_ptr, R->user_item, nullptr, rating64,
                                                               cnt, (unsigned long long *)ub_key, ub, (unsigned long long *)srec);
        else
            k_sort_profiles3<false><<<grid, dim3(256), 0, st>>>(R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating,
                                                                nullptr, cnt, (unsigned long long *)ub_key, ub, (unsigned long long *)srec);
        XM_LAUNCH_CHECK();
    }
    // sort records -> rater records in item order, W+
    XM_HIP(hipMemsetAsync(Wp, 0, sizeof(uint64_t) * In, st));
    if (nnz > 0 && I > 0) {
        ts::Geo G;
        ts_geometry(I, nnz, wide ? ts::Chunk<3>::CH : ts::Chunk<2>::CH, G);
        rcode = ts_prepare(st, G, (const long long *)item_ptr);
        if (rcode) return rcode;
        const dim3 gridA((unsigned)((nnz + G.ch - 1) / G.ch)), gridB((unsigned)G.clist_cap), gridL((unsigned)G.slist_cap);
        if (wide) {
            ts::RecLoader<3> LA{(const unsigned long long *)srec}, LB{(const unsigned long long *)bufA};
            ts::k_ts_bin<3, false, ts::RecLoader<3>><<<gridA, dim3(ts::BT), 0, st>>>(G, LA, nnz, (unsigned long long *)bufA);
            XM_LAUNCH_CHECK();
            ts::k_ts_bin<3, true, ts::RecLoader<3>><<<gridB, dim3(ts::BT), 0, st>>>(G, LB, nnz, (unsigned long long *)bufB);
            XM_LAUNCH_CHECK();
            k_rc_tiles<true><<<dim3((unsigned)G.T), dim3(ts::CT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)R->user_ptr,
                                                                          (ulonglong2 *)rc, (unsigned long long *)Wp);
            XM_LAUNCH_CHECK();
            k_rc_large<true><<<gridL, dim3(ts::LT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)R->user_ptr,
                                                             (ulonglong2 *)rc, (unsigned long long *)Wp);
            XM_LAUNCH_CHECK();
        } else {
            ts::RecLoader<2> LA{(const unsigned long long *)srec}, LB{(const unsigned long long *)bufA};
            ts::k_ts_bin<2, false, ts::RecLoader<2>><<<gridA, dim3(ts::BT), 0, st>>>(G, LA, nnz, (unsigned long long *)bufA);
            XM_LAUNCH_CHECK();
            ts::k_ts_bin<2, true, ts::RecLoader<2>><<<gridB, dim3(ts::BT), 0, st>>>(G, LB, nnz, (unsigned long long *)bufB);
            XM_LAUNCH_CHECK();
            k_rc_tiles<false><<<dim3((unsigned)G.T), dim3(ts::CT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)R->user_ptr,
                                                                           (ulonglong2 *)rc, (unsigned long long *)Wp);
            XM_LAUNCH_CHECK();
            k_rc_large<false><<<gridL, dim3(ts::LT), 0, st>>>(G, (const unsigned long long *)bufB, (const long long *)R->user_ptr,
                                                              (ulonglong2 *)rc, (unsigned long long *)Wp);
            XM_LAUNCH_CHECK();
        }
    }
    }   // phases & 1
    // item statistics of [stats_lo, stats_hi) from the rater records (+ those records' mutuality flags), big items in chunks
    if ((phases & 2) && stats_hi > stats_lo) {
        BigList B;
        B.chunk_cap = nnz / STAT_CHK + nnz / STAT_BIG + 2;
        B.item_cap = nnz / STAT_BIG + 2;
        XM_HIP(xm_malloc_async((void **)&B.counters, 2 * sizeof(unsigned), st));
        XM_HIP(xm_malloc_async((void **)&B.chunks, sizeof(int2) * (size_t)B.chunk_cap, st));
        XM_HIP(xm_malloc_async((void **)&B.items, sizeof(int4) * (size_t)B.item_cap, st));
        XM_HIP(xm_malloc_async((void **)&B.part, sizeof(double) * 5 * (size_t)B.chunk_cap, st));
        XM_HIP(hipMemsetAsync(B.counters, 0, 2 * sizeof(unsigned), st));
        const dim3 grid((unsigned)((stats_hi - stats_lo + 15) / 16)), gridC((unsigned)((B.chunk_cap + 3) / 4)), gridI((unsigned)((B.item_cap + 63) / 64));
        if (wide) {
            const RcWideSrc src{(const RaterRecWide *)rc};
            k_item_stats3<RcWideSrc><<<grid, dim3(256), 0, st>>>(I, stats_lo, stats_hi, (const long long *)item_ptr, src, info, norms, B);
            XM_LAUNCH_CHECK();
            k_item_chunks<RcWideSrc><<<gridC, dim3(256), 0, st>>>((const long long *)item_ptr, src, B);
            XM_LAUNCH_CHECK();
            k_item_big<<<gridI, dim3(64), 0, st>>>(I, B, info, norms);
            XM_LAUNCH_CHECK();
        } else {
            const RcSrc src{(RaterRec *)rc, u_avg};
            k_item_stats3<RcSrc><<<grid, dim3(256), 0, st>>>(I, stats_lo, stats_hi, (const long long *)item_ptr, src, info, norms, B);
            XM_LAUNCH_CHECK();
            k_item_chunks<RcSrc><<<gridC, dim3(256), 0, st>>>((const long long *)item_ptr, src, B);
            XM_LAUNCH_CHECK();
            k_item_big<<<gridI, dim3(64), 0, st>>>(I, B, info, norms);
            XM_LAUNCH_CHECK();
            k_item_big_flags<RcSrc><<<gridC, dim3(256), 0, st>>>((const long long *)item_ptr, src, B, info);
            XM_LAUNCH_CHECK();
        }
    }
    // mutuality flags from the COMPLETE item info: the profile copy's (4), the rater records' of the items another rank's
    // statistics covered (8)
    if (!wide && nnz > 0 && I > 0) {
        if (phases & 8) {
            k_rc_flags<<<dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st>>>(nnz, (RaterRec *)rc, (const int2 *)ub, info);
            XM_LAUNCH_CHECK();
        }
        if (phases & 4) {
            k_ub_flags<<<dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st>>>(nnz, (int2 *)ub, info);
            XM_LAUNCH_CHECK();
        }
    }
    if (h_ctl) {
        XM_HIP(hipMemcpyAsync(h_ctl, ctl, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (h_ctl[1] > HMAX) {
            set_error("heavy set larger than %d", HMAX);
            return XMAP_ERR_OVERFLOW;
        }
    }
    return XMAP_OK;
}

int xmap_sim3_mircount(void *stream, int32_t n_items, int64_t coo_cap, const int32_t *coo_i, const int32_t *coo_j,
                       const int64_t *d_shards, int64_t n_pairs, int32_t skip_self, void *scratch, int32_t *mir) {
    XM_SCOPE(stream);
    XM_ARG(coo_i && coo_j && scratch && mir && n_items >= 0 && coo_cap >= 0 && n_pairs >= 0);
    XM_ARG(d_shards ? (coo_cap >= COO_SHARDS) : (n_pairs <= coo_cap));
    hipStream_t st = (hipStream_t)stream;
    if (n_items == 0) return XMAP_OK;
    XM_HIP(hipMemsetAsync(mir, 0, sizeof(int32_t) * (size_t)n_items, st));
    if (n_pairs == 0 || coo_cap == 0) return XMAP_OK;
    if (d_shards)
        return mirror_counts(st, n_items, coo_cap / COO_SHARDS, COO_SHARDS, (const unsigned long long *)d_shards, coo_i, coo_j,
                             skip_self != 0, (int *)scratch, mir);
    return mirror_counts(st, n_items, n_pairs, 1, nullptr, coo_i, coo_j, skip_self != 0, (int *)scratch, mir);
}

int xmap_sim3_mirror(void *stream, int32_t n_items, int64_t coo_cap, const int32_t *coo_i, const int32_t *coo_j,
                     const double *coo_sim, const int32_t *coo_mutu, const int32_t *coo_nij, const int64_t *d_shards, int64_t n_pairs,
                     const int32_t *own, const int32_t *mir, int32_t *tot, int64_t *row_ptr, int64_t *mptr, int32_t *fill,
                     void *bufA, void *bufB, int32_t *col, double *sim, int32_t *mutu, int32_t *nij, const double *coo_aux, double *aux,
                     int32_t row_lo, int32_t row_hi) {
    XM_SCOPE(stream);
    XM_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= n_items);
    XM_ARG(coo_i && coo_j && coo_sim && coo_mutu && coo_nij && own && mir && tot && row_ptr && mptr && fill && bufA && bufB);
    XM_ARG(col && sim && mutu && nij && n_items >= 0 && coo_cap >= 0 && n_pairs >= 0 && n_pairs < 0x7fffffffLL);
    XM_ARG(d_shards ? (coo_cap >= COO_SHARDS) : (n_pairs <= coo_cap));
    XM_ARG((coo_aux != nullptr) == (aux != nullptr));
    hipStream_t st = (hipStream_t)stream;
    const int I = n_items;
    if (I > 0) {
        k_row_totals<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, own, mir, tot);
        XM_LAUNCH_CHECK();
    }
    int rcode = xmap_exclusive_scan_i32_to_i64(stream, tot, row_ptr, I, nullptr);
    if (rcode) return rcode;
    rcode = xmap_exclusive_scan_i32_to_i64(stream, mir, mptr, I, nullptr);
    if (rcode) return rcode;
    if (n_pairs == 0 || I == 0 || coo_cap == 0) return XMAP_OK;
    XM_HIP(hipMemsetAsync(fill, 0, sizeof(int32_t) * (size_t)I, st));
    ts::Geo G;
    ts_geometry(I, n_pairs, coo_aux ? ts::Chunk<4>::CH : ts::Chunk<3>::CH, G);      // (n_pairs bounds the mirrored records: a self pair has none)
    rcode = ts_prepare(st, G, (const long long *)mptr);
    if (rcode) return rcode;
    // the COO's chunks: from the shard cursors of the pair kernels, or one range of n_pairs records
    static_assert(ts::Chunk<3>::CH <= ts::Chunk<4>::CH && ts::Chunk<4>::CH % ts::Chunk<3>::CH == 0, "chunk lists by the narrow record width");
    const int n_shards = d_shards ? COO_SHARDS : 1;
    const long long shard_cap = d_shards ? coo_cap / COO_SHARDS : coo_cap;
    const long long chunk_cap = n_pairs / ts::Chunk<3>::CH + n_shards + 1;
    longlong2 *chunks = nullptr;
    unsigned *n_chunks = nullptr;
    XM_HIP(xm_malloc_async((void **)&chunks, sizeof(longlong2) * (size_t)chunk_cap, st));
    XM_HIP(xm_malloc_async((void **)&n_chunks, sizeof(unsigned), st));
    XM_HIP(hipMemsetAsync(n_chunks, 0, sizeof(unsigned), st));
    k_coo_chunks<<<dim3((unsigned)((n_shards + 255) / 256)), dim3(256), 0, st>>>(n_shards, shard_cap, (const unsigned long long *)d_shards,
                                                                                 n_pairs, chunks, n_chunks, chunk_cap);
    XM_LAUNCH_CHECK();
    const bool share = row_lo > 0 || row_hi < I;
#define XM_MIRROR(AUX_, SHARE_, CA_, A_) mirror_levels<AUX_, SHARE_>(st, G, coo_cap, coo_i, coo_j, coo_sim, coo_mutu, coo_nij, CA_, chunks, n_chunks, \
        chunk_cap, n_pairs, own, row_ptr, fill, bufA, bufB, col, sim, mutu, nij, A_, row_lo, row_hi)
    if (coo_aux) return share ? XM_MIRROR(true, true, coo_aux, aux) : XM_MIRROR(true, false, coo_aux, aux);
    return share ? XM_MIRROR(false, true, nullptr, nullptr) : XM_MIRROR(false, false, nullptr, nullptr);
#undef XM_MIRROR
}
}
