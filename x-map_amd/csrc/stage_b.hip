// stage_b.hip -- cross-domain top-k similarity extension (extender_pipeline, reference
// utils/assist.py:80-133; core/extender.py).
//
// Kernels (DESIGN.md section "Stage B"):
//   k_bridge_flags  : bb[i] = any kept pair of row i whose 2-char prefixes differ        (HBM-bound, one pass over D')
//   k_knn_classify  : per row, chunked bitonic sort in LDS by (|sim| desc, col asc) and the two
//                     filtered top-k lists of find_knn_items                              (HBM-bound, one pass over D')
//   k_reverse       : reverse adjacencies (attach / src / rnn) built in row order with an O(1)
//                     membership test against the k-th entry of the neighbour's list; count pass (attach + rnn together)
//                     leaves a byte per entry, the fill passes read it                    (bound by the CU's gather rate)
//   k_joint_list, k_att_columns, k_mid_rows<count|place>
//                   : middle lists -- per non-bridge x' the (t, s, x) records of its joint paths, grouped by column x
//                     (tile directory + 64-byte records); one block per x', tile counters in LDS, flat walk
//   k_col_home, k_col_ends, k_paths4 (heads_Q), finalize_*, k_merge_groups, k_merge
//                   : the path enumeration (default): start-major, heads merged by column, one row update per column,
//                     rows indexed by end rank, exact (value, error) sums, fused top-10     (random HBM row updates)
//   k_w_*           : exact per-start path counts (scheduling weights)
//   k_paths         : per-path enumeration, one wave per start (fallback beyond the middle-list budget; cross-check)
//   under XMAP_CROSSCHECK (libxmap_hip_xcheck.so only): k_mid_build / k_mid_dir (dense-table middle lists), k_paths2
#include "common.h"
#include <stdlib.h>

namespace xmap {

// =============================================================================================
__global__ __launch_bounds__(256) void k_bridge_flags(int I, const long long *row_ptr, const int *col,
                                                      const int *prefix_cls, uint8_t *bb) {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    int lane = lane_id();
    long long lo = row_ptr[i], hi = row_ptr[i + 1];
    int pi = prefix_cls[i];
    int found = 0;
    for (long long b = lo; b < hi && !found; b += 64) {
        long long p = b + lane;
        int f = (p < hi) && (prefix_cls[col[p]] != pi);
        found = __ballot(f) != 0ull;
    }
    if (lane == 0) bb[i] = (uint8_t)found;
}

// =============================================================================================
constexpr int K_THREADS = 256;
constexpr int K_CH = 2048;  // entries sorted per chunk (32 KB of LDS)
constexpr int K_CH_SMALL = 512;   // rows up to this length: the 8 KB instance of k_knn_classify
constexpr int K_WIN = 1024;  // entries streamed against the thresholds per step (rows longer than one chunk)

__device__ __forceinline__ bool before(unsigned long long ka, int ca, unsigned long long kb, int cb) {
    return (ka > kb) || (ka == kb && ca < cb);
}

// exclusive scan of one long long per thread across the block (256 threads)
__device__ __forceinline__ long long block_scan_ll(long long v, long long *total, long long *smem) {
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
#pragma unroll
    for (int k = 0; k < K_THREADS / 64; k++) {
        long long s = smem[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

struct KnnArgs {
    int I, k, row_lo;
    const long long *row_ptr;
    const int *col;
    const double *sim;
    const int *mutu;
    const int *nij;
    const double *info;
    const double *frac;
    const uint8_t *bb;
    const int *suffix_cls;
    const uint32_t *contains_mask;
    uint8_t *cls;
    int *kcnt;
    int *kcol;
    double *kval;
};

// CH = entries sorted per chunk = the block's LDS (16 B each).  Two instances share the rows: rows of at most CH_SMALL
// entries (nearly all of them) run with 8 KB of LDS -- eight blocks per CU instead of five: a block is a chain of dependent
// gathers (row, class predicates, list values), and more blocks in flight is what hides them --, the long rows with 32 KB.
template <int CH, int MODE>      // MODE 0: every row; 1: rows of at most CH entries; 2: rows of more than K_CH_SMALL
__global__ __launch_bounds__(K_THREADS) void k_knn_classify(KnnArgs A) {
    __shared__ unsigned long long skey[CH];
    __shared__ int scol[CH];
    __shared__ int spos[CH];
    __shared__ long long sscan[4];
    __shared__ unsigned long long s_thrk[2];
    __shared__ int s_thrc[2], s_has[2], s_fill;

    const int i = blockIdx.x + A.row_lo;
    const int tid = threadIdx.x;
    const long long lo = A.row_ptr[i];
    const int n = (int)(A.row_ptr[i + 1] - lo);
    if ((MODE == 1 && n > CH) || (MODE == 2 && n <= K_CH_SMALL)) return;      // (the other instance's row)
    const int k = A.k;
    // the unused tail of a list is zero (the tables come uninitialised: a fill of the ~1 GB they take at k = 50 cost
    // more than the lists of the few short rows)
    auto zero_tail = [&](int nA, int nB) {
        for (int e = tid; e < 2 * k; e += K_THREADS) {
            const int l = e >= k, r = e - l * k;
            if (r < (l ? nB : nA)) continue;
            const size_t o = ((size_t)i * 2 + l) * k + r;
            A.kcol[o] = 0; A.kval[o * 3] = 0.0; A.kval[o * 3 + 1] = 0.0; A.kval[o * 3 + 2] = 0.0;
        }
    };
    if (n == 0) {
        if (tid == 0) {
            A.cls[i] = 0;
            A.kcnt[(size_t)i * 2] = 0;
            A.kcnt[(size_t)i * 2 + 1] = 0;
        }
        zero_tail(0, 0);
        return;
    }
    const bool isbb = A.bb[i] != 0;
    const int sc = A.suffix_cls[i];
    if (tid < 2) s_has[tid] = 0;
    int nc = 0, consumed = 0;
    for (;;) {
        int total;
        if (consumed == 0 || CH - nc < K_WIN) {
            const int take = (CH - nc) < (n - consumed) ? (CH - nc) : (n - consumed);
            for (int t = tid; t < take; t += K_THREADS) {
                int p = consumed + t;
                double s = A.sim[lo + p];
                skey[nc + t] = (unsigned long long)__double_as_longlong(fabs(s));
                scol[nc + t] = A.col[lo + p];
                spos[nc + t] = p;
            }
            total = nc + take;
            consumed += take;
        } else {
            // Rows longer than one chunk (the popular items: 1e5 entries and more): after the first sort the k-th best of
            // each list is known, and an entry that does not sort before it can never enter that list -- the rest of the
            // row is streamed against the two thresholds, K_WIN entries per step, and only the survivors are buffered
            // (a few hundred for 1e5 entries in random order) instead of sorting every 2048 of them.
            if (tid == 0) s_fill = nc;
            __syncthreads();
            for (;;) {
                const int f = s_fill;      // the same value for every thread: nobody is past the barrier below yet
                __syncthreads();
                if (consumed >= n || CH - f < K_WIN) break;
#pragma unroll
                for (int u = 0; u < K_WIN / K_THREADS; u++) {
                    const int p = consumed + tid + K_THREADS * u;
                    if (p < n) {
                        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(A.sim[lo + p]));
                        const int c = A.col[lo + p];
                        bool pa, pb;
                        if (isbb) {
                            bool has = (A.contains_mask[c] >> sc) & 1u;
                            pa = !has; pb = has;
                        } else {
                            pa = A.bb[c] != 0; pb = true;
                        }
                        const bool keep = (pa && (!s_has[0] || before(key, c, s_thrk[0], s_thrc[0]))) ||
                                          (pb && (!s_has[1] || before(key, c, s_thrk[1], s_thrc[1])));
                        if (keep) {
                            const int o = atomicAdd(&s_fill, 1);
                            skey[o] = key; scol[o] = c; spos[o] = p;
                        }
                    }
                }
                consumed = (consumed + K_WIN) < n ? (consumed + K_WIN) : n;
                __syncthreads();
            }
            total = s_fill;
        }
        int N = 2;
        while (N < total) N <<= 1;
        for (int t = total + tid; t < N; t += K_THREADS) {
            skey[t] = 0ull;
            scol[t] = 0x7fffffff;
            spos[t] = -1;
        }
        __syncthreads();
        // bitonic sort by (|sim| desc, col asc); pads (|sim| = 0) end up last.
        // Up to 128 entries (two thirds of the rows): ONE wave runs the whole network -- 64 compare-exchanges per step, the
        // LDS operations of a wave execute in order, so the 28 steps need no block barrier (a barrier per step, 36 of them
        // for 256 entries, was most of a short row's time: 400 000 blocks x ~17 us).
        if (N <= 128) {
            if (tid < 64) {
                for (int k2 = 2; k2 <= N; k2 <<= 1) {
                    for (int j = k2 >> 1; j > 0; j >>= 1) {
                        const int t = tid;
                        if (t < (N >> 1)) {
                            int a = 2 * t - (t & (j - 1));
                            int b = a + j;
                            bool up = (a & k2) == 0;
                            unsigned long long ka = skey[a], kb = skey[b];
                            int ca = scol[a], cb = scol[b];
                            bool sw = up ? before(kb, cb, ka, ca) : before(ka, ca, kb, cb);
                            if (sw) {
                                skey[a] = kb; skey[b] = ka;
                                scol[a] = cb; scol[b] = ca;
                                int pa = spos[a], pb = spos[b];
                                spos[a] = pb; spos[b] = pa;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
            }
            __syncthreads();
        } else
        for (int k2 = 2; k2 <= N; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (N >> 1); t += K_THREADS) {
                    int a = 2 * t - (t & (j - 1));
                    int b = a + j;
                    bool up = (a & k2) == 0;
                    unsigned long long ka = skey[a], kb = skey[b];
                    int ca = scol[a], cb = scol[b];
                    bool sw = up ? before(kb, cb, ka, ca) : before(ka, ca, kb, cb);
                    if (sw) {
                        skey[a] = kb; skey[b] = ka;
                        scol[a] = cb; scol[b] = ca;
                        int pa = spos[a], pb = spos[b];
                        spos[a] = pb; spos[b] = pa;
                    }
                }
                __syncthreads();
            }
        }
        // class predicates, ranks in sorted order
        const int per = (N + K_THREADS - 1) / K_THREADS;
        const int s0 = tid * per;
        int cA = 0, cB = 0;
        for (int t = s0; t < s0 + per && t < total; t++) {
            int c = scol[t];
            bool pa, pb;
            if (isbb) {
                bool has = (A.contains_mask[c] >> sc) & 1u;  // domain_label in pair[0]
                pa = !has; pb = has;
            } else {
                pa = A.bb[c] != 0; pb = true;                 // NB_NN keeps every neighbour
            }
            cA += pa; cB += pb;
        }
        long long tot;
        long long ex = block_scan_ll(((long long)cB << 32) | (unsigned)cA, &tot, sscan);
        int rA = (int)(ex & 0xffffffffll), rB = (int)(ex >> 32);
        const int totA = (int)(tot & 0xffffffffll), totB = (int)(tot >> 32);
        const bool last = consumed >= n;
        if (last) {
            for (int t = s0; t < s0 + per && t < total; t++) {
                int c = scol[t];
                bool pa, pb;
                if (isbb) {
                    bool has = (A.contains_mask[c] >> sc) & 1u;
                    pa = !has; pb = has;
                } else {
                    pa = A.bb[c] != 0; pb = true;
                }
                long long p = lo + spos[t];
                if ((pa && rA < k) || (pb && rB < k)) {
                    double sv = A.sim[p];
                    double mu = (double)A.mutu[p];
                    double fr = A.frac ? A.frac[p]
                                       : 1.0 * mu / (A.info[(size_t)i * 4 + 3] + A.info[(size_t)c * 4 + 3] - (double)A.nij[p]);
                    if (pa && rA < k) {
                        size_t o = ((size_t)i * 2 + 0) * k + rA;
                        A.kcol[o] = c; A.kval[o * 3] = sv; A.kval[o * 3 + 1] = mu; A.kval[o * 3 + 2] = fr;
                    }
                    if (pb && rB < k) {
                        size_t o = ((size_t)i * 2 + 1) * k + rB;
                        A.kcol[o] = c; A.kval[o * 3] = sv; A.kval[o * 3 + 1] = mu; A.kval[o * 3 + 2] = fr;
                    }
                }
                rA += pa; rB += pb;
            }
            if (tid == 0) {
                int nA = totA < k ? totA : k, nB = totB < k ? totB : k;
                uint8_t c = isbb ? 1 : (nA > 0 ? 2 : 0);  // no bridge neighbour -> dropped (extender.py:39)
                A.cls[i] = c;
                A.kcnt[(size_t)i * 2] = c ? nA : 0;
                A.kcnt[(size_t)i * 2 + 1] = c ? nB : 0;
            }
            zero_tail(totA < k ? totA : k, totB < k ? totB : k);
            return;
        }
        // carry the selected <= 2k entries to the front (sorted order kept), then take the next chunk
        unsigned long long rk[CH / K_THREADS];
        int rc[CH / K_THREADS], rp[CH / K_THREADS];
        int nk = 0;
        for (int t = s0; t < s0 + per && t < total; t++) {
            int c = scol[t];
            bool pa, pb;
            if (isbb) {
                bool has = (A.contains_mask[c] >> sc) & 1u;
                pa = !has; pb = has;
            } else {
                pa = A.bb[c] != 0; pb = true;
            }
            if ((pa && rA < k) || (pb && rB < k)) { rk[nk] = skey[t]; rc[nk] = c; rp[nk] = spos[t]; nk++; }
            if (pa && rA == k - 1) { s_thrk[0] = skey[t]; s_thrc[0] = c; s_has[0] = 1; }     // the k-th best of a list
            if (pb && rB == k - 1) { s_thrk[1] = skey[t]; s_thrc[1] = c; s_has[1] = 1; }
            rA += pa; rB += pb;
        }
        long long tk;
        long long ek = block_scan_ll((long long)nk, &tk, sscan);
        for (int q = 0; q < nk; q++) {
            skey[ek + q] = rk[q]; scol[ek + q] = rc[q]; spos[ek + q] = rp[q];
        }
        nc = (int)tk;
        __syncthreads();
    }
}

// =============================================================================================
// membership of item a in list l of neighbour b, given |sim(a,b)| (bit-symmetric by construction):
// a is in the list iff it passes the list's class predicate and sorts at or before the list's
// last entry in the order (|sim| desc, col asc) -- or the list is not full.
// KnnThr: that last entry of every list as one 16-byte record (12.8 MB for 4e5 items: resident in the Infinity Cache,
// where the lists themselves, 1.1 GB, are not)
struct KnnThr { double la; int col; int cnt; };
__global__ __launch_bounds__(256) void k_knn_thresholds(int I, int k, const int *kcnt, const int *kcol, const double *kval, KnnThr *thr) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2ll * I) return;
    KnnThr th;
    th.cnt = kcnt[t]; th.col = 0; th.la = 0.0;
    if (th.cnt > 0) {
        const size_t o = (size_t)t * k + (th.cnt - 1);
        th.la = fabs(kval[o * 3]); th.col = kcol[o];
    }
    thr[t] = th;
}

struct RevArgs {
    int I, k, mode;
    int row_lo, row_hi;          // the rows (= targets of the reverse lists) of this call: a rank's share, or [0, I)
    const KnnThr *thr;
    const int *long_rows;        // [0] = count, then the rows with more than rev_long entries (or NULL)
    uint8_t *eflag;              // per entry of the rows [row_lo, row_hi) (index p - row_ptr[row_lo]), or NULL: the count pass leaves
                                 // bit 0 = "b lists a", bit 1 = joint here and the fill pass reads it instead of testing again
    int rev_long;
    const long long *row_ptr;
    const int *col;
    const double *sim;
    const int *mutu;
    const int *nij;
    const double *info;
    const double *frac;
    const uint8_t *bb;
    const uint8_t *cls;
    const int *kcnt;
    const int *kcol;
    const double *kval;
    const int *suffix_cls;
    const uint32_t *contains_mask;
    const uint8_t *flags;
    const long long *attach_ptr;
    int *rcnt;
    int *rcnt2;                  // mode 3 (attach and rnn lists counted in ONE pass over the rows): the rnn counts
    const long long *rptr;
    int *ridx;
    double *rval;
    uint8_t *rflag;
};

__device__ __forceinline__ bool in_list(const RevArgs &A, int b, int l, int a, double abs_sim) {
    KnnThr th;
    th.cnt = 0; th.col = 0; th.la = 0.0;
    if (A.thr) th = A.thr[(size_t)b * 2 + l];        // one 16-byte gather instead of count, last value, last column
    int c = A.thr ? th.cnt : A.kcnt[(size_t)b * 2 + l];
    if (c == 0) return false;
    bool pred;
    if (A.cls[b] == 1) {
        bool has = (A.contains_mask[a] >> A.suffix_cls[b]) & 1u;
        pred = (l == 0) ? !has : has;
    } else {
        pred = (l == 0) ? (A.bb[a] != 0) : true;
    }
    if (!pred) return false;
    if (c < A.k) return true;
    if (A.thr) return (abs_sim > th.la) || (abs_sim == th.la && a <= th.col);
    size_t o = ((size_t)b * 2 + l) * A.k + (c - 1);
    double la = fabs(A.kval[o * 3]);
    return (abs_sim > la) || (abs_sim == la && a <= A.kcol[o]);
}

// one entry p of row a: does b = col[p] list a?  (mode 0 attach, 1 src, 2 rnn; fl: the (t,s) is joint)
__device__ __forceinline__ bool rev_entry(const RevArgs &A, int a, long long p, long long hi, int &b, double &sv, uint8_t &fl) {
    bool ok = false;
    b = 0; sv = 0.0; fl = 0;
    if (p < hi) {
        b = A.col[p];
        sv = A.sim[p];
        double ab = fabs(sv);
        int cb = A.cls[b];     // (a 1-byte gather from a 400 KB table; the 16-byte threshold record only for the entries that pass it --
                               //  packing the class into that record made EVERY entry gather it: 6.7 -> 8.3 ms, round 4)
        if (A.mode == 0) {           // attach(a): x = b non-bridge record with a in NB_BB(x)
            ok = (cb == 2) && in_list(A, b, 0, a, ab);
        } else if (A.mode == 1) {    // src(t = a): s = b
            // (the three per-item tests packed into one byte table, one gather instead of up to three: 5.2 ms either way, round 4)
            ok = (cb == 1) && (A.flags[b] & 1) && (A.attach_ptr[b + 1] > A.attach_ptr[b]) &&
                 (in_list(A, b, 0, a, ab) || in_list(A, b, 1, a, ab));
            if (ok) {
                bool joint = (A.cls[a] == 1) && (A.attach_ptr[a + 1] > A.attach_ptr[a]) &&
                             (in_list(A, a, 0, b, ab) || in_list(A, a, 1, b, ab));
                fl = joint ? 1 : 0;
            }
        } else if (A.mode == 3) {    // attach and rnn together (count pass only): ok = attach, fl bit 1 = rnn (eflag bit 2)
            if (cb == 2) { ok = in_list(A, b, 0, a, ab); if (in_list(A, b, 1, a, ab)) fl = 2; }
        } else {                     // rnn(y = a): x = b non-bridge record with a in NB_NN(x)
            ok = (cb == 2) && in_list(A, b, 1, a, ab);
        }
    }
    return ok;
}
__device__ __forceinline__ void rev_write(const RevArgs &A, int a, long long p, long long o, int b, double sv, uint8_t fl) {
    double mu = (double)A.mutu[p];
    A.ridx[o] = b;
    A.rval[o * 3] = sv;
    A.rval[o * 3 + 1] = mu;
    A.rval[o * 3 + 2] = A.frac ? A.frac[p] : 1.0 * mu / (A.info[(size_t)a * 4 + 3] + A.info[(size_t)b * 4 + 3] - (double)A.nij[p]);
    if (A.rflag) A.rflag[o] = fl;
}

// The test of one entry, once: the count pass evaluates it (a 1-byte class gather per entry, a 16-byte threshold gather for
// those that pass: the passes run at the CU's rate of random gathers, not at the matrix's bandwidth) and, given A.eflag, leaves
// the outcome as a byte per entry; the fill pass then streams the bytes and touches only the entries it writes.
template <bool FILL>
__device__ __forceinline__ bool rev_test(const RevArgs &A, int a, long long p, long long hi, long long p0, int &b, double &sv,
                                         uint8_t &fl) {
    if (!A.eflag) return rev_entry(A, a, p, hi, b, sv, fl);
    if (!FILL) {
        const bool ok = rev_entry(A, a, p, hi, b, sv, fl);
        // (bit 0 = attach / src, bit 1 = joint (src), bit 2 = rnn: a fused count serves the fill passes of both of its lists)
        if (p < hi) A.eflag[p - p0] = (uint8_t)((ok ? (A.mode == 2 ? 4 : 1) : 0) | (fl << 1));
        return ok;
    }
    b = 0; sv = 0.0; fl = 0;
    bool ok = false;
    if (p < hi) {
        const uint8_t e = A.eflag[p - p0];
        ok = (e & (A.mode == 2 ? 4 : 1)) != 0;
        fl = (uint8_t)((e >> 1) & 1);
        if (ok) { b = A.col[p]; sv = A.sim[p]; }
    }
    return ok;
}

// Rows up to REV_LONG entries: one wave per row.  The rows of the popular items have 10^5 entries and more; walked by
// one wave each they were the whole duration of the pass (5 ms per pass for 0.2 ms of streaming): those rows are listed
// (k_rev_long_rows) and walked by blocks of 16 waves, 1024 entries per step, in the same (row) order.
constexpr int REV_LONG = 4096;      // default of RevArgs::rev_long (XMAP_REV_LONG overrides it: tests walk every row both ways)
constexpr int REV_WAVES = 16;
template <bool FILL>
__global__ __launch_bounds__(256) void k_reverse(RevArgs A) {
    int a = A.row_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= A.row_hi) return;
    int lane = lane_id();
    long long lo = A.row_ptr[a], hi = A.row_ptr[a + 1];
    if (A.long_rows && hi - lo > A.rev_long) return;
    bool row_ok = true;
    if (A.mode == 1) row_ok = (A.flags[a] & 2) != 0;  // "T:" in t
    long long out = FILL ? A.rptr[a] : 0;
    int total = 0, total2 = 0;
    const long long p0 = A.eflag ? A.row_ptr[A.row_lo] : 0;
    if (row_ok)
        for (long long base = lo; base < hi; base += 64) {
            long long p = base + lane;
            int b; double sv; uint8_t fl;
            const bool ok = rev_test<FILL>(A, a, p, hi, p0, b, sv, fl);
            unsigned long long m = __ballot(ok);
            if (FILL && ok) rev_write(A, a, p, out + __popcll(m & lanemask_lt()), b, sv, fl);
            int c = __popcll(m);
            out += c;
            total += c;
            if (!FILL && A.mode == 3) total2 += __popcll(__ballot((fl & 2) != 0));
        }
    if (!FILL && lane == 0) { A.rcnt[a] = total; if (A.mode == 3) A.rcnt2[a] = total2; }
}

__global__ __launch_bounds__(256) void k_rev_long_rows(int row_lo, int row_hi, const long long *row_ptr, int rev_long,
                                                       int *long_rows /*[0] = count*/) {
    const int a = row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (a < row_hi && row_ptr[a + 1] - row_ptr[a] > rev_long) long_rows[1 + atomicAdd(&long_rows[0], 1)] = a;
}

template <bool FILL>
__global__ __launch_bounds__(64 * REV_WAVES) void k_reverse_long(RevArgs A) {
    __shared__ int s_cnt[REV_WAVES];
    __shared__ int s_tot2;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int n_long = A.long_rows[0];
    const long long p0 = A.eflag ? A.row_ptr[A.row_lo] : 0;
    for (int r = blockIdx.x; r < n_long; r += gridDim.x) {
        const int a = A.long_rows[1 + r];
        const long long lo = A.row_ptr[a], hi = A.row_ptr[a + 1];
        const bool row_ok = (A.mode != 1) || ((A.flags[a] & 2) != 0);
        long long out = FILL ? A.rptr[a] : 0;
        int total = 0, total2 = 0;
        if (!FILL && A.mode == 3) { if (threadIdx.x == 0) s_tot2 = 0; __syncthreads(); }
        if (row_ok)
            for (long long base = lo; base < hi; base += 64 * REV_WAVES) {
                const long long p = base + threadIdx.x;
                int b; double sv; uint8_t fl;
                const bool ok = rev_test<FILL>(A, a, p, hi, p0, b, sv, fl);
                const unsigned long long m = __ballot(ok);
                if (!FILL && A.mode == 3) total2 += __popcll(__ballot((fl & 2) != 0));      // (this wave's rnn entries)
                if (lane == 0) s_cnt[w] = __popcll(m);
                __syncthreads();
                int before = 0, all = 0;
                for (int o = 0; o < REV_WAVES; o++) { const int c = s_cnt[o]; if (o < w) before += c; all += c; }
                if (FILL && ok) rev_write(A, a, p, out + before + __popcll(m & lanemask_lt()), b, sv, fl);
                out += all;
                total += all;
                __syncthreads();
            }
        if (!FILL && A.mode == 3) {
            if (lane == 0) atomicAdd(&s_tot2, total2);
            __syncthreads();
            if (threadIdx.x == 0) A.rcnt2[a] = s_tot2;
            __syncthreads();
        }
        if (!FILL && threadIdx.x == 0) A.rcnt[a] = total;
    }
}

// =============================================================================================
struct PathArgs {
    int I, k;
    const uint8_t *cls;
    const int *kcnt;
    const int *kcol;
    const double *kval;
    const uint8_t *flags;
    const long long *att_ptr; const int *att_idx; const double *att_val;
    const long long *src_ptr; const int *src_idx; const double *src_val; const uint8_t *src_flag;
    const long long *rnn_ptr; const int *rnn_idx; const double *rnn_val;
    // work units: (start, chunk c of G).  G == 1: the unit owns the start, accumulates in the wave's slot
    // row and finalises it.  G > 1: the start's (head, t) entries are dealt round-robin to G units, each
    // with a dedicated row (unit_row); k_merge adds the rows up and finalises.
    int n_units;
    const int *unit_start; const int *unit_c; const int *unit_G; const int *unit_row; int *unit_nt;
    int n_slots;
    double *acc; int *touched;     // slot rows   [n_slots][I][4] / [n_slots][I]
    double *hacc; int *htouched;   // heavy rows  [n_rows][I][4]  / [n_rows][I]
    int *n_cand; int *top_end; double *top_val;
    long long xs_cap; long long *xs_off; int *xs_end; double *xs_val;
    unsigned long long *counters;  // [0] total candidates, [1] paths, [2] work cursor, [3] xs cursor
    // rows of k_paths4 are indexed by the rank of an item among the items that can end a path (U of them) instead of by
    // the item: uitem[rank] = item, urank[item] = rank.  The older kernels leave these NULL / U = I.
    int U; const int *urank; const int *uitem;
    long long row_stride;          // entries per accumulator row (k_paths4: U unless an ablation build asks for more)
};

struct Carry { double sm, mu, c; };  // sum sim*mutu, sum mutu, prod frac_mutu along the path so far

__device__ __forceinline__ Carry first_edge(double sim, double mutu, double frac) {
    Carry r; r.sm = sim * mutu; r.mu = mutu; r.c = frac; return r;   // python sum(): 0 + x == x
}
__device__ __forceinline__ Carry add_edge(Carry a, double sim, double mutu, double frac) {
    Carry r; r.sm = a.sm + sim * mutu; r.mu = a.mu + mutu; r.c = a.c * frac; return r;
}

// Error-free accumulation (Knuth two-sum, double-double running sums): the per-(start,end) sums become
// independent of the order in which paths are enumerated (to ~2^-104), so items with identical
// path multisets tie exactly and the tie-break (ascending end index) is well defined.
struct WaveAcc {
    double *acc; int *touched; int nt; unsigned long long paths;
    __device__ __forceinline__ void add(bool active, int end, Carry p) {
        bool first = false;
        if (active) {
            double sp = (p.mu != 0.0) ? 1.0 * p.sm / p.mu : 0.0;   // calculate_path_confidence (extender.py:83-89)
            double *a = acc + (size_t)end * 4;
            double s_hi = a[0], s_lo = a[1], c_hi = a[2], c_lo = a[3];
            first = (c_hi == 0.0);
            dd_add(s_hi, s_lo, sp * p.c);
            dd_add(c_hi, c_lo, p.c);
            a[0] = s_hi; a[1] = s_lo; a[2] = c_hi; a[3] = c_lo;
        }
        unsigned long long m = __ballot(first);
        if (first) touched[nt + __popcll(m & lanemask_lt())] = end;
        nt += __popcll(m);
        paths += __popcll(__ballot(active));
    }
};

// tails of one (t,s) after edge (t,s): end s is accumulated by the caller (vector step over s);
// here: for x in attach(s): end x, then end y for y in NN(x)         (extender.py:134-138 / :154-158)
template <class ACC>
__device__ __forceinline__ void tails(const PathArgs &A, ACC &W, int s, Carry c_ts) {
    const int lane = lane_id();
    const int k = A.k;
    long long a0 = A.att_ptr[s], a1 = A.att_ptr[s + 1];
    for (long long ap = a0; ap < a1; ap++) {
        const int x = A.att_idx[ap];
        const Carry c_sx = add_edge(c_ts, A.att_val[ap * 3], A.att_val[ap * 3 + 1], A.att_val[ap * 3 + 2]);
        const int nn = A.kcnt[(size_t)x * 2 + 1];
        for (int b = 0; b < nn + 1; b += 64) {
            int idx = b + lane;
            bool act = idx < nn + 1;
            int end = x;
            Carry c = c_sx;
            if (act && idx > 0) {
                size_t o = ((size_t)x * 2 + 1) * k + (idx - 1);
                end = A.kcol[o];
                c = add_edge(c_sx, A.kval[o * 3], A.kval[o * 3 + 1], A.kval[o * 3 + 2]);
            }
            W.add(act, end, c);
        }
    }
}

// all (t,s) of src(t) behind a given head carry (head_len = number of edges in front of (t,s))
template <class ACC>
__device__ __forceinline__ void through_t(const PathArgs &A, ACC &W, int t, bool has_head, Carry head) {
    const int lane = lane_id();
    long long s0 = A.src_ptr[t], s1 = A.src_ptr[t + 1];
    for (long long base = s0; base < s1; base += 64) {
        long long p = base + lane;
        bool act = p < s1;
        int s = 0;
        Carry c; c.sm = 0; c.mu = 0; c.c = 0;
        if (act) {
            if (has_head && !(A.src_flag[p] & 1)) act = false;  // joint paths need (t,s) in TGT as well
        }
        if (act) {
            s = A.src_idx[p];
            double sv = A.src_val[p * 3], mu = A.src_val[p * 3 + 1], fr = A.src_val[p * 3 + 2];
            c = has_head ? add_edge(head, sv, mu, fr) : first_edge(sv, mu, fr);
        }
        W.add(act, s, c);  // path ... -> t -> s
        unsigned long long m = __ballot(act);
        while (m) {
            int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            int sb = rl32(s, l);
            Carry cb;
            cb.sm = rld(c.sm, l); cb.mu = rld(c.mu, l); cb.c = rld(c.c, l);
            tails(A, W, sb, cb);
        }
    }
}

// wave-wide selection of the XMAP_TOPC best of nt candidates in the order (|xsim| desc, end asc);
// get(b, end, val) returns candidate b.  Lane 0 writes the result.
template <typename Get>
__device__ __forceinline__ void select_topc(int nt, Get get, int *top_end, double *top_val) {
    const int lane = lane_id();
    unsigned long long pk = 0;
    int pe = -1;
    int nsel = nt < XMAP_TOPC ? nt : XMAP_TOPC;
    for (int r = 0; r < nsel; r++) {
        unsigned long long bk = 0;
        int be = 0x7fffffff;
        double bv = 0.0;
        bool have = false;
        for (int b = lane; b < nt; b += 64) {
            int e; double v;
            get(b, e, v);
            unsigned long long key = (unsigned long long)__double_as_longlong(fabs(v));
            bool after_prev = (r == 0) || (key < pk) || (key == pk && e > pe);
            if (after_prev && (!have || key > bk || (key == bk && e < be))) { bk = key; be = e; bv = v; have = true; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            unsigned long long ok = __shfl_xor(bk, m, 64);
            int oe = __shfl_xor(be, m, 64);
            double ov = __shfl_xor(bv, m, 64);
            int oh = __shfl_xor((int)have, m, 64);
            if (oh && (!have || ok > bk || (ok == bk && oe < be))) { bk = ok; be = oe; bv = ov; have = true; }
        }
        pk = bk; pe = be;
        if (lane == 0) { top_end[r] = be; top_val[r] = bv; }
    }
}

__global__ __launch_bounds__(256) void k_topc_lists(int I, const long long *xs_ptr, const int *xs_end, const double *xs_val,
                                                    int *n_cand, int *top_end, double *top_val) {
    int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= I) return;
    long long lo = xs_ptr[s];
    int nt = (int)(xs_ptr[s + 1] - lo);
    select_topc(nt, [&](int b, int &e, double &v) { e = xs_end[lo + b]; v = xs_val[lo + b]; },
                top_end + (size_t)s * XMAP_TOPC, top_val + (size_t)s * XMAP_TOPC);
    if (lane_id() == 0) n_cand[s] = nt;
}

// xsim = sum(s_p c_p) / sum(c_p) (get_sim, extender.py:198-201), fused top-XMAP_TOPC by (|xsim| desc,
// end asc) -- all a Generator reads (generator.py:85,109) --, optional full lists, row reset.
// ONE pass over the start's row: every touched entry is read once, divided, (full mode: written to the start's list,)
// zeroed, and offered to a running selection.  The row entries are random 32-byte accesses to HBM (a row is larger than
// an XCD's L2), so the earlier form -- a division pass, XMAP_TOPC selection passes over the row and a reset pass -- cost
// 12 random accesses per candidate against ~6 for accumulating it.  Running selection: candidates whose key is >= the
// key of the XMAP_TOPC-th best so far (ties included: the order among equal keys is by end index) are appended to a
// per-wave LDS buffer; when it passes FIN_CAP entries it is cut back to its exact XMAP_TOPC best, which raises the
// threshold.  An entry is only ever dropped when XMAP_TOPC entries with a strictly larger key exist, so the result is
// the exact top of the whole list; a stream in random order appends ~XMAP_TOPC ln(nt / XMAP_TOPC) entries.
constexpr int FIN_CAP = 128;
struct FinBuf { double v[FIN_CAP + 64]; int e[FIN_CAP + 64]; double ov[XMAP_TOPC]; int oe[XMAP_TOPC]; };

__device__ __forceinline__ unsigned long long xsim_key(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }

// the full-list cursor of one start (lane 0 draws it); returns whether the list fits
__device__ __forceinline__ bool fin_list_offset(const PathArgs &A, int nt, int start, unsigned long long &off) {
    off = 0;
    if (!(A.xs_cap > 0 && nt > 0)) return false;   // full candidate lists (extender_pipeline's RDD) via a cursor
    if (lane_id() == 0) off = atomicAdd(&A.counters[3], (unsigned long long)nt);
    off = ((unsigned long long)(unsigned)rl32((int)(off >> 32), 0) << 32) | (unsigned)rl32((int)(off & 0xffffffffull), 0);
    const bool full = (long long)(off + nt) <= A.xs_cap;
    if (lane_id() == 0) A.xs_off[start] = full ? (long long)off : -1;
    return full;
}

// exact XMAP_TOPC best of the nbuf buffered candidates; lane 0 writes them in order
__device__ __forceinline__ int fin_cut(FinBuf &F, int nbuf, int *out_e, double *out_v) {
    volatile double *bv = F.v;
    volatile int *be = F.e;
    select_topc(nbuf, [&](int b, int &e, double &v) { e = be[b]; v = bv[b]; }, out_e, out_v);
    return nbuf < XMAP_TOPC ? nbuf : XMAP_TOPC;
}

// One wave's share of the pass: candidates b = 64 (w + j NW) + lane.  Leaves the best ns of them, in order, in
// F.oe / F.ov and returns ns.
__device__ __forceinline__ int finalize_slice(const PathArgs &A, FinBuf &F, double *acc, const int *touched, int nt,
                                              unsigned long long off, bool full, int w, int NW, int gs = 1, int mem = 0) {
    const int lane = lane_id();
    volatile double *bv = F.v;
    volatile int *be = F.e;
    int nbuf = 0;
    unsigned long long thr = 0;   // key of the XMAP_TOPC-th best so far (0 while fewer have been seen)
    for (int b0 = 64 * w; b0 < nt; b0 += 64 * NW) {
        const int b = b0 + lane;
        const bool act = b < nt;
        int e = 0;
        double v = 0.0;
        unsigned long long key = 0;
        if (act) {
            e = touched[b];
            double *a = acc + ((size_t)e * gs + mem) * 4;
            v = 1.0 * (a[0] + a[1]) / (a[2] + a[3]);     // pairs of k_paths4 are not renormalised; a renormalised pair is its own sum
#ifndef Q_FIN3
            a[0] = 0.0; a[1] = 0.0; a[2] = 0.0; a[3] = 0.0;
#endif
            key = xsim_key(v);
            if (full) { A.xs_end[off + b] = A.uitem ? A.uitem[e] : e; A.xs_val[off + b] = v; }
        }
#ifdef Q_FIN3
        {   // (-DQ_FIN3, not the default: inside the box-to-box noise, 478-501 ms either way) the entries are zeroed by lane PAIRS:
            // both lanes of a pair write one half each of the even lane's entry, then of the odd lane's -- two store instructions
            // over 32 lines each instead of two over 64
            const unsigned long long ab = act ? (unsigned long long)(acc + ((size_t)e * gs + mem) * 4) : 0ull;
#define XM_PAIR(CTRL) { const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)ab, CTRL, 0xf, 0xf, true);            \
                        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(ab >> 32), CTRL, 0xf, 0xf, true);      \
                        typedef __attribute__((address_space(1))) double gf64;      /* (a GLOBAL pointer: the integer round trip loses the address space) */ \
                        gf64 *z = (gf64 *)(((unsigned long long)hi_ << 32) | lo_);                                                   \
                        if (z) { z[(lane & 1) * 2] = 0.0; z[(lane & 1) * 2 + 1] = 0.0; } }
            XM_PAIR(0xA0)       // quad_perm [0,0,2,2]: the even lane's entry
            XM_PAIR(0xF5)       // quad_perm [1,1,3,3]: the odd lane's entry
#undef XM_PAIR
        }
#endif
        // (rows of k_paths4 are indexed by end RANK; the item behind a rank -- a random 4-byte gather, a third of the pass's
        //  memory requests -- is looked up only for the candidates that pass the running threshold: ~10 ln(n / 10) per start)
        const bool q = act && key >= thr;
        const unsigned long long m = __ballot(q);
        if (q) { const int p = nbuf + __popcll(m & lanemask_lt()); bv[p] = v; be[p] = A.uitem ? A.uitem[e] : e; }
        nbuf += __popcll(m);
        if (nbuf > FIN_CAP) {
            const int ns = fin_cut(F, nbuf, F.oe, F.ov);
            int te = 0;
            double tv = 0.0;
            if (lane < ns) { te = ((volatile int *)F.oe)[lane]; tv = ((volatile double *)F.ov)[lane]; }
            if (lane < ns) { be[lane] = te; bv[lane] = tv; }
            nbuf = ns;
            thr = (ns == XMAP_TOPC) ? xsim_key(rld(tv, XMAP_TOPC - 1)) : 0ull;
        }
    }
    return fin_cut(F, nbuf, F.oe, F.ov);
}

__device__ __forceinline__ int finalize_start(const PathArgs &A, FinBuf &F, double *acc, const int *touched, int nt, int start,
                                              int gs = 1, int mem = 0) {
    const int lane = lane_id();
    if (lane == 0) A.n_cand[start] = nt;
    unsigned long long off;
    const bool full = fin_list_offset(A, nt, start, off);
    const int ns = finalize_slice(A, F, acc, touched, nt, off, full, 0, 1, gs, mem);
    if (lane < ns) {
        A.top_end[(size_t)start * XMAP_TOPC + lane] = ((volatile int *)F.oe)[lane];
        A.top_val[(size_t)start * XMAP_TOPC + lane] = ((volatile double *)F.ov)[lane];
    }
    return nt;
}

__global__ __launch_bounds__(256) void k_paths(PathArgs A) {
    __shared__ FinBuf fin[4];
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.n_slots) return;
    const int lane = lane_id();
    const int k = A.k;
    WaveAcc W;
    W.paths = 0;
    unsigned long long cand_total = 0;
    for (;;) {
        int u_ = 0;
        if (lane == 0) u_ = (int)atomicAdd(&A.counters[2], 1ull);
        const int unit = uniform(u_);
        if (unit >= A.n_units) break;  // every wave reaches this exit: the cursor only grows
        const int start = uniform(A.unit_start[unit]);
        const int c = uniform(A.unit_c[unit]);
        const int G = uniform(A.unit_G[unit]);
        const int row = uniform(A.unit_row[unit]);
        if (row < 0) {
            W.acc = A.acc + (size_t)slot * A.I * 4;
            W.touched = A.touched + (size_t)slot * A.I;
        } else {
            W.acc = A.hacc + (size_t)row * A.I * 4;
            W.touched = A.htouched + (size_t)row * A.I;
        }
        W.nt = 0;
        int ent = 0;  // running index of the start's (head, t) entries; unit c takes ent % G == c
        // role T: start = t (final_nonjoint_extend on every SRC record, extender.py:124-140,:180)
        if (A.flags[start] & 2) {
            if (G == 1 || ent % G == c) {
                Carry none; none.sm = 0; none.mu = 0; none.c = 0;
                through_t(A, W, start, false, none);
            }
            ent++;
        }
        // role X': start = x' in attach(t) (target_path, extender.py:160-163)
        if (A.cls[start] == 2) {
            int nb = A.kcnt[(size_t)start * 2];
            for (int q = 0; q < nb; q++) {
                size_t o = ((size_t)start * 2) * k + q;
                int t = A.kcol[o];
                if (!(A.flags[t] & 2)) continue;  // BB_other_intra_target keeps "T:" bridges only (:175)
                if (G == 1 || ent % G == c) {
                    Carry h = first_edge(A.kval[o * 3], A.kval[o * 3 + 1], A.kval[o * 3 + 2]);
                    through_t(A, W, t, true, h);
                }
                ent++;
            }
        }
        // role Y': start = y' in NN(x'), x' in attach(t) (longest_path, extender.py:164-167)
        {
            long long r0 = A.rnn_ptr[start], r1 = A.rnn_ptr[start + 1];
            for (long long rp = r0; rp < r1; rp++) {
                int xp = A.rnn_idx[rp];
                Carry h0 = first_edge(A.rnn_val[rp * 3], A.rnn_val[rp * 3 + 1], A.rnn_val[rp * 3 + 2]);
                int nb = A.kcnt[(size_t)xp * 2];
                for (int q = 0; q < nb; q++) {
                    size_t o = ((size_t)xp * 2) * k + q;
                    int t = A.kcol[o];
                    if (!(A.flags[t] & 2)) continue;
                    if (G == 1 || ent % G == c) {
                        Carry h = add_edge(h0, A.kval[o * 3], A.kval[o * 3 + 1], A.kval[o * 3 + 2]);
                        through_t(A, W, t, true, h);
                    }
                    ent++;
                }
            }
        }
        if (row < 0) cand_total += finalize_start(A, fin[threadIdx.x >> 6], W.acc, W.touched, W.nt, start);
        else if (lane == 0) A.unit_nt[unit] = W.nt;
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], cand_total);
        atomicAdd(&A.counters[1], W.paths);
    }
}

// =============================================================================================
// Second formulation of the enumeration ("middle lists").  Every joint path has the shape
//   [y'] - x' - t - s - [x - [y]]      with x', x non-bridge items, t in NB_BB(x'), (t,s) joint.
// For each non-bridge x' the middles (t,s,x) are materialised ONCE, grouped by x (one "tile" per (x', x); a dense
// n_nb x n_nb count table gives the tile offsets, so the build is a tally pass + a placement pass over
// (x', t) work items -- no per-x' serial section).  A head (start, x') then streams the tiles of x': all
// records of one tile hit the same ends {x} U NN(x), so each lane keeps its end's double-double sums in
// REGISTERS across the tile and the start's row in HBM is touched once per (head, tile) instead of once per
// path.  The edge products sim*mutu and the fractions are stored per edge, so a path's (sum sim*mutu, sum
// mutu, prod frac) is rebuilt in the reference's left-to-right order, bit for bit.
struct MidX { double sm2, sm3, sm4, f2, f3, f4, mu; int xid; int pad; };   // 64 B; xid = index of x in nb_list
struct MidDir { int x; int ne; int cnt; int pad; long long off; };          // one tile of x': item x, 1+|NN(x)| ends, records [off, off+cnt); pad = index of x in nb_list

struct MidArgs {
    int I, k;
    const uint8_t *cls; const int *kcnt; const int *kcol; const double *kval; const uint8_t *flags;
    const long long *att_ptr; const int *att_idx; const double *att_val;
    const long long *src_ptr; const int *src_idx; const double *src_val; const uint8_t *src_flag;
    int n_nb; const int *nb_list; const int *nb_id;
    const long long *jptr; const int *joff;     // joint (t, s) of every t, compacted: offsets into src(t) (k_joint_list; k_mid_rows)
    const int *axid;                            // column of every attach entry (k_att_columns)
    const long long *xoff; int *xl;             // rows wider than the LDS span: the columns of a row's records in walk order, [xoff[x'], xoff[x'+1])
    int *tile_cnt;                 // [n_nb * n_nb] tally, then placement cursor
    const long long *tile_off;     // [n_nb * n_nb + 1]
    MidX *midX;
};

#ifdef XMAP_CROSSCHECK      // the dense-table form of the middle lists (XMAP_MID_TABLE=1): a test formulation, libxmap_hip_xcheck.so only
// one wave per (x', position q in NB_BB(x')): lanes over the joint (t,s), each walks attach(s)
template <bool PLACE>
__global__ __launch_bounds__(256) void k_mid_build(MidArgs A) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long long)A.n_nb * A.k) return;
    const int xpid = (int)(w / A.k), q = (int)(w % A.k);
    const int xp = A.nb_list[xpid];
    if (q >= A.kcnt[(size_t)xp * 2]) return;
    const size_t o = ((size_t)xp * 2) * A.k + q;
    const int t = A.kcol[o];
    if (!(A.flags[t] & 2)) return;
    const int lane = lane_id();
    const double v2 = A.kval[o * 3], m2 = A.kval[o * 3 + 1], f2 = A.kval[o * 3 + 2];              // edge (x', t)
    for (long long p = A.src_ptr[t] + lane; p < A.src_ptr[t + 1]; p += 64) {
        if (!(A.src_flag[p] & 1)) continue;
        const int s = A.src_idx[p];
        const double v3 = A.src_val[p * 3], m3 = A.src_val[p * 3 + 1], f3 = A.src_val[p * 3 + 2];  // edge (t, s)
        for (long long ap = A.att_ptr[s]; ap < A.att_ptr[s + 1]; ap++) {
            const int xid = A.nb_id[A.att_idx[ap]];
            const size_t tile = (size_t)xpid * A.n_nb + xid;
            if (!PLACE) {
                atomicAdd(&A.tile_cnt[tile], 1);
            } else {
                const long long pos = A.tile_off[tile] + atomicAdd(&A.tile_cnt[tile], 1);
                const double v4 = A.att_val[ap * 3], m4 = A.att_val[ap * 3 + 1], f4 = A.att_val[ap * 3 + 2];  // edge (s, x)
                MidX r;
                r.sm2 = v2 * m2; r.sm3 = v3 * m3; r.sm4 = v4 * m4; r.f2 = f2; r.f3 = f3; r.f4 = f4;
                r.mu = (m2 + m3) + m4; r.xid = xid; r.pad = 0;
                A.midX[pos] = r;
            }
        }
    }
}

// Row-wise construction of the middle lists (default): ONE block per x', the tile sizes of its row in LDS (the row of
// the dense table without the table).  PHASE 0 counts the row's records and non-empty tiles; PHASE 1 repeats the tally,
// turns it into offsets (block scan), writes the row's tile directory in x order and places the records with LDS cursors.
// No global atomics (the table form spends 1.6e8 of them per pass, twice, on a 3 GB table) and no n_nb^2 memory.
#endif  // XMAP_CROSSCHECK
// The LDS holds the counters of `span` columns (<= XMAP_MID_ROWS_SPAN): a row with more non-bridge items than that is
// built in column ranges [x0, x0 + span), one after the other -- every range walks the row's (t, s, x) again and keeps the
// x of its range, the directory and the records of the ranges follow each other (x order is kept).  Rounds 1-2 fell back
// to the dense n_nb x n_nb table beyond 40 000 non-bridge items (120 GB at 1e5) and the coarse ABI refused.
// the value of lane SRC of every quad, in all four lanes of the quad (DPP quad_perm: no LDS traffic)
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int CTRL = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// joint (t, s) of every source list, compacted in list order: joff[jptr[t] .. jptr[t+1]) = the offsets inside src(t) of the
// entries with the joint flag (2.3 % of them at BASELINE configs[1]: the walk of k_mid_rows reads these instead of scanning the
// lists of a row's neighbours once per row).  joff == NULL: the counts (jcnt) only.
__global__ __launch_bounds__(256) void k_joint_list(int I, const long long *src_ptr, const uint8_t *src_flag, int *jcnt,
                                                    const long long *jptr, int *joff) {
    // a wave takes 64 items: their ranges one per lane (most items have no source list), then the non-empty lists one by one
    const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (t0 >= I) return;
    const int lane = lane_id();
    const int tl = t0 + lane;
    long long s0l = 0, s1l = 0, outl = 0;
    if (tl < I) { s0l = src_ptr[tl]; s1l = src_ptr[tl + 1]; if (joff) outl = jptr[tl]; }
    int totl = 0;
    unsigned long long todo = __ballot(s1l > s0l);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long long s0 = rl64(s0l, l), s1 = rl64(s1l, l);
        long long out = rl64(outl, l);
        int total = 0;
        for (long long base = s0; base < s1; base += 64) {
            const long long p = base + lane;
            const bool ok = p < s1 && (src_flag[p] & 1);
            const unsigned long long m = __ballot(ok);
            if (joff && ok) joff[out + __popcll(m & lanemask_lt())] = (int)(p - s0);
            out += __popcll(m);
            total += __popcll(m);
        }
        if (lane == l) totl = total;
    }
    if (!joff && tl < I) jcnt[tl] = totl;
}

// the column (index among the non-bridge items) of every attach entry: nb_id[att_idx[ap]] gathered ONCE per call -- the walks of
// k_mid_rows read it 1.6e8 times per walk, and a gather of 64 random lines costs the CU ~320 cycles per instruction
// (profiles/ta_rate.hip): 1.3 ms per walk, three walks per call
__global__ __launch_bounds__(256) void k_att_columns(long long bound, int I, const long long *att_ptr, const int *att_idx, const int *nb_id,
                                                     int *axid) {
    const long long ap = (long long)blockIdx.x * 256 + threadIdx.x;
    if (ap < bound && ap < att_ptr[I]) axid[ap] = nb_id[att_idx[ap]];
}

// records behind every t: sum of the attach-list lengths over its joint (t, s) -- and per row x' over its neighbours t: the
// exact record count of a row BEFORE it is built (rows wider than the LDS span keep their records' columns in a scratch list)
__global__ __launch_bounds__(256) void k_joint_records(int I, const long long *jptr, const int *joff, const long long *src_ptr,
                                                       const int *src_idx, const long long *att_ptr, long long *jrec) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= I) return;
    const int lane = lane_id();
    const long long j0 = jptr[t], j1 = jptr[t + 1], s0 = src_ptr[t];
    long long sum = 0;
    for (long long j = j0 + lane; j < j1; j += 64) {
        const int sx = src_idx[s0 + joff[j]];
        sum += att_ptr[sx + 1] - att_ptr[sx];
    }
    sum = wave_sum_ll(sum);
    if (lane == 0) jrec[t] = sum;
}
__global__ __launch_bounds__(256) void k_row_records(int n_nb, int k, const int *nb_list, const int *kcnt, const int *kcol,
                                                     const uint8_t *flags, const long long *jrec, long long *rowrec) {
    const int xpid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xpid >= n_nb) return;
    const int lane = lane_id();
    const int xp = nb_list[xpid];
    const int nq = kcnt[(size_t)xp * 2];
    long long sum = 0;
    for (int q = lane; q < nq; q += 64) {
        const int t = kcol[((size_t)xp * 2) * k + q];
        if (flags[t] & 2) sum += jrec[t];
    }
    sum = wave_sum_ll(sum);
    if (lane == 0) rowrec[xpid] = sum;
}

constexpr int MIDROW_WAVES = 16;
template <int PHASE, bool ONE_RANGE>
__global__ __launch_bounds__(64 * MIDROW_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mid_rows(MidArgs A, int span, int *ng, long long *nrec, const long long *dir_ptr,
                                                                const long long *rec_ptr, MidDir *dir) {
    extern __shared__ int bins[];                      // [span]
    __shared__ unsigned long long s_wave[MIDROW_WAVES];
    const int xpid = blockIdx.x;
    const int xp = A.nb_list[xpid];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int n_nb = A.n_nb;
    const int nq = A.kcnt[(size_t)xp * 2];
#ifdef EXP_MID_WALK1     // (rounds 2-4a: a serial loop over the row's neighbours t, one joint (t, s) per wave step -- kept for the ablation)
    // the joint (t,s) of the row are dealt round-robin to the waves (every wave scans the flags, 64 at a time); the
    // lanes of a wave walk attach(s) together (coalesced, distinct x: no two lanes meet on a counter)
    auto walk = [&](int x0, int x1, auto &&body) {
        int ctr = 0;
        for (int q = 0; q < nq; q++) {
            const size_t o = ((size_t)xp * 2) * A.k + q;
            const int t = A.kcol[o];
            if (!(A.flags[t] & 2)) continue;
            const double v2 = A.kval[o * 3], m2 = A.kval[o * 3 + 1], f2 = A.kval[o * 3 + 2];              // edge (x', t)
            const long long s0 = A.src_ptr[t], s1 = A.src_ptr[t + 1];
            for (long long base = s0; base < s1; base += 64) {
                const long long pl = base + lane;
                unsigned long long m = __ballot(pl < s1 && (A.src_flag[pl < s1 ? pl : s0] & 1));
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if ((ctr++ % MIDROW_WAVES) != w) continue;
                    const long long p = base + l;
                    const int s = A.src_idx[p];
                    for (long long ap = A.att_ptr[s] + lane; ap < A.att_ptr[s + 1]; ap += 64) {
                        const int xid = A.nb_id[A.att_idx[ap]];
                        if (ONE_RANGE || (xid >= x0 && xid < x1)) body(true, xid, v2, m2, f2, p, ap, 0ll);
                    }
                }
            }
        }
    };
#else
    // The walk, round 4.  A row of configs[1] has 10 000 records on average (median 12, p90 33 000, maximum 93 000:
    // profiles/r04m_mid_walk.txt) behind <= 50 neighbours t and their joint (t, s), and ONE block builds it: what the block
    // takes is the chain of dependent memory trips of its slowest wave.  The first form ran the neighbours as a serial loop
    // (three trips per t), scanned their source lists for the joint flag (2.3 % of the entries have it) and gave a wave one
    // joint per step: ~300 trips per wave and walk in the big rows.  Now:
    //  * the joint entries of every source list are compacted once per call (k_joint_list: jptr / joff), and the row's
    //    neighbours are loaded one per LANE into a block-shared table with the prefix sums of their joint counts: the row's
    //    joints are ONE flat list, taken 64 at a time by every wave alike (three trips: offset -> s -> attach range);
    //  * the records of a chunk of 64 joints are a flat list too (scan of the attach-list lengths over the lanes; a record's
    //    joint by a six-step search over the starts in LDS), dealt to the waves in rounds of 64 and walked MID_UNROLL rounds
    //    at a time, so that the two trips of a round (attach entry -> its column) overlap with those of its neighbours.
    // ~100 trips per wave and walk in the biggest row, and every lane of every round but a chunk's last is busy.
    constexpr int MID_UNROLL = PHASE == 0 ? 4 : 2;      // (the placement keeps a record's values live: two rounds fit 64 VGPRs)
    __shared__ long long q_s0[64], q_jlo[64];
    __shared__ double q_v2[64], q_m2[64], q_f2[64];
    __shared__ int q_joff[65];
    __shared__ int s_off[MIDROW_WAVES][64];          // (what else a record needs of its joint comes from the joint's LANE by
    volatile int *w_off = s_off[w];                  //  ds_bpermute: with 64.5 KB of counters at configs[1], two blocks per CU need the rest small)
    auto walk = [&](int x0, int x1, auto &&body) {
        long long cbase = 0;                            // records of the chunks in front: a record's index in the row's walk order
        for (int qb = 0; qb < nq; qb += 64) {
            __syncthreads();                            // (nobody reads the previous table any more)
            if (w == 0) {
                const int ql = qb + lane;
                long long s0 = 0, jlo = 0;
                int jn = 0;
                double v2 = 0.0, m2 = 0.0, f2 = 0.0;
                if (ql < nq) {
                    const size_t o = ((size_t)xp * 2) * A.k + ql;
                    const int t = A.kcol[o];
                    if (A.flags[t] & 2) {
                        s0 = A.src_ptr[t]; jlo = A.jptr[t]; jn = (int)(A.jptr[t + 1] - jlo);
                        v2 = A.kval[o * 3]; m2 = A.kval[o * 3 + 1]; f2 = A.kval[o * 3 + 2];          // edge (x', t)
                    }
                }
                int incl = jn;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
                q_s0[lane] = s0; q_jlo[lane] = jlo; q_v2[lane] = v2; q_m2[lane] = m2; q_f2[lane] = f2;
                q_joff[lane] = incl - jn;               // joints of the neighbours in front (a neighbour without joints: its successor's)
                if (lane == 63) q_joff[64] = incl;
            }
            __syncthreads();
            const int J = q_joff[64];
            for (int g0 = 0; g0 < J; g0 += 64) {        // 64 joints of the row; every wave takes every chunk, the ROUNDS are dealt
                const int g = g0 + lane;
                int q = 0;                              // the last neighbour whose joints start at or before g
#pragma unroll
                for (int st = 32; st >= 1; st >>= 1) if (q_joff[q + st] <= g) q += st;
                int jo = 0, len = 0;
                long long a0 = 0;
                if (g < J) {
                    jo = A.joff[q_jlo[q] + (g - q_joff[q])];
                    const int s = A.src_idx[q_s0[q] + jo];
                    a0 = A.att_ptr[s];
                    len = (int)(A.att_ptr[s + 1] - a0);
                }
                int incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
                const int T = rl32(incl, 63);
                if (T == 0) continue;
                const long long cb = cbase;
                cbase += T;
                __builtin_amdgcn_wave_barrier();
                w_off[lane] = incl - len;                // first record of the lane's joint (lanes without records: their successor's)
                __builtin_amdgcn_wave_barrier();
                const int a0_lo = (int)(a0 & 0xffffffffll), a0_hi = (int)(a0 >> 32);
                for (int r0 = 64 * w; r0 < T; r0 += 64 * MIDROW_WAVES * MID_UNROLL) {
                    int jj[MID_UNROLL], xi[MID_UNROLL], rr[MID_UNROLL];
                    long long ap[MID_UNROLL];
#pragma unroll
                    for (int u = 0; u < MID_UNROLL; u++) {
                        const int r = r0 + u * 64 * MIDROW_WAVES + lane;
                        rr[u] = r;
                        int j = 0;                       // the last lane whose joint starts at or before record r
#pragma unroll
                        for (int st = 32; st >= 1; st >>= 1) if (w_off[j + st] <= r) j += st;
                        jj[u] = r < T ? j : -1;
                        const long long ja0 = ((long long)__shfl(a0_hi, j, 64) << 32) | (unsigned int)__shfl(a0_lo, j, 64);
                        ap[u] = r < T ? ja0 + (r - w_off[j]) : 0;
                    }
#pragma unroll
                    for (int u = 0; u < MID_UNROLL; u++) xi[u] = jj[u] >= 0 ? A.axid[ap[u]] : -1;
#pragma unroll
                    for (int u = 0; u < MID_UNROLL; u++) {
                        const int jl = jj[u] >= 0 ? jj[u] : 0;
                        const int qq = __shfl(q, jl, 64), jjo = __shfl(jo, jl, 64);      // (all lanes take part in the exchange)
                        // (every lane calls: the placement moves its records between the lanes of a quad)
                        body(jj[u] >= 0 && (ONE_RANGE || (xi[u] >= x0 && xi[u] < x1)), xi[u], q_v2[qq], q_m2[qq], q_f2[qq], q_s0[qq] + jjo, ap[u], cb + rr[u]);
                    }
                }
            }
        }
    };
#endif
    unsigned long long done = 0;                        // (non-empty tiles << 40 | records) of the ranges before this one
    const long long rbase = PHASE ? rec_ptr[xpid] : 0, dbase = PHASE ? dir_ptr[xpid] : 0;
#ifndef EXP_MID_WALK1
    // A row wider than the LDS span is built range by range, and every range needs the tally of ITS columns.  The first form
    // walked the row again for every tally (27 walks per row at the S1 shape: nine ranges, count + tally + placement); now ONE
    // walk leaves the column of every record in a scratch list (the walk order is deterministic and the rows' record counts
    // are known beforehand: k_joint_records / k_row_records), and the ranges' tallies stream it.
    int *stash = nullptr;
    long long n_stash = 0;
    if (!ONE_RANGE) {
        stash = A.xl + A.xoff[xpid];
        n_stash = A.xoff[xpid + 1] - A.xoff[xpid];
        walk(0, n_nb, [&](bool valid, int xid, double, double, double, long long, long long, long long ridx) { if (valid && ridx < n_stash) stash[ridx] = xid; });
        __syncthreads();
    }
#endif
    for (int x0 = 0; x0 < n_nb; x0 += ONE_RANGE ? n_nb : span) {
        const int x1 = (ONE_RANGE || (x0 + span) >= n_nb) ? n_nb : (x0 + span), nx = x1 - x0;
        if (!ONE_RANGE) __syncthreads();                // (the previous range's placement is over)
        for (int i = threadIdx.x; i < nx; i += 64 * MIDROW_WAVES) bins[i] = 0;
        __syncthreads();
#ifndef EXP_MID_WALK1
        if (!ONE_RANGE) {
            for (long long i = threadIdx.x; i < n_stash; i += 64 * MIDROW_WAVES) {
                const int xid = stash[i];
                if (xid >= x0 && xid < x1) atomicAdd(&bins[xid - x0], 1);
            }
        } else
#endif
        walk(x0, x1, [&](bool valid, int xid, double, double, double, long long, long long, long long) { if (valid) atomicAdd(&bins[xid - x0], 1); });
        __syncthreads();
        // per thread a run of consecutive bins: (non-empty tiles << 40 | records), block-wide exclusive scan
        const int per = (nx + 64 * MIDROW_WAVES - 1) / (64 * MIDROW_WAVES);
        const int b0 = threadIdx.x * per, b1 = (b0 + per) < nx ? (b0 + per) : nx;
        unsigned long long mine = 0;
        for (int i = b0; i < b1; i++) { const int c = bins[i]; mine += (unsigned long long)c + (c ? (1ull << 40) : 0ull); }
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        unsigned long long before = 0, total = 0;
        for (int o = 0; o < MIDROW_WAVES; o++) { const unsigned long long v = s_wave[o]; if (o < w) before += v; total += v; }
        if (PHASE == 1) {
            const unsigned long long ex = done + before + incl - mine;
            int rank = (int)(ex >> 40);
            long long off = (long long)(ex & ((1ull << 40) - 1));
            for (int i = b0; i < b1; i++) {
                const int c = bins[i];
                bins[i] = (int)off;                         // placement cursor of the tile (records of a row fit 31 bits)
                if (c) {
                    MidDir d;
                    d.x = A.nb_list[x0 + i]; d.ne = 1 + A.kcnt[(size_t)d.x * 2 + 1]; d.cnt = c; d.pad = x0 + i; d.off = rbase + off;
                    dir[dbase + rank] = d;
                    rank++;
                    off += c;
                }
            }
            __syncthreads();
            walk(x0, x1, [&](bool valid, int xid, double v2, double m2, double f2, long long p, long long ap, long long) {
                long long pos = 0;
                MidX r;
                r.sm2 = 0.0; r.sm3 = 0.0; r.sm4 = 0.0; r.f2 = 0.0; r.f3 = 0.0; r.f4 = 0.0; r.mu = 0.0; r.xid = 0; r.pad = 0;
                if (valid) {
                    pos = rbase + atomicAdd(&bins[xid - x0], 1);
#ifdef EXP_MID_NOVAL      // (ablation: no gathers of the edge values -- wrong records, timing only)
                    const double v3 = 1.0, m3 = 1.0, f3 = 1.0, v4 = 1.0, m4 = (double)(p + ap), f4 = 1.0;
#else
                    const double v3 = A.src_val[p * 3], m3 = A.src_val[p * 3 + 1], f3 = A.src_val[p * 3 + 2];      // edge (t, s)
                    const double v4 = A.att_val[ap * 3], m4 = A.att_val[ap * 3 + 1], f4 = A.att_val[ap * 3 + 2];  // edge (s, x)
#endif
                    r.sm2 = v2 * m2; r.sm3 = v3 * m3; r.sm4 = v4 * m4; r.f2 = f2; r.f3 = f3; r.f4 = f4;
                    r.mu = (m2 + m3) + m4; r.xid = xid; r.pad = 0;
                }
#if defined(EXP_MID_NOSTORE)    // (ablation: the records are not written -- timing only)
                if (valid && r.mu == -1.5) A.midX[pos] = r;
#elif defined(EXP_MID_WALK1) || defined(EXP_MID_ST1)      // (a record per lane: four 16-byte stores over 64 lines each)
                if (valid) A.midX[pos] = r;
#else
                // The four records of a QUAD of lanes leave as four store instructions of 16 lines each instead of four of 64
                // (the CU's memory path charges a store by the lines it touches, profiles/ta_rate.hip): the 4 x 4 pieces of 16 bytes
                // are transposed inside the quad (two DPP butterfly steps), lane q then holds piece q of each of the quad's records
                uint4 P[4];
                {
                    const long long b0 = __double_as_longlong(r.sm2), b1 = __double_as_longlong(r.sm3), b2 = __double_as_longlong(r.sm4);
                    const long long b3 = __double_as_longlong(r.f2), b4 = __double_as_longlong(r.f3), b5 = __double_as_longlong(r.f4);
                    const long long b6 = __double_as_longlong(r.mu);
                    P[0] = make_uint4((unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32));
                    P[1] = make_uint4((unsigned)b2, (unsigned)(b2 >> 32), (unsigned)b3, (unsigned)(b3 >> 32));
                    P[2] = make_uint4((unsigned)b4, (unsigned)(b4 >> 32), (unsigned)b5, (unsigned)(b5 >> 32));
                    P[3] = make_uint4((unsigned)b6, (unsigned)(b6 >> 32), (unsigned)r.xid, (unsigned)r.pad);
                }
                const int ql = lane & 3;
                const bool odd = (ql & 1) != 0, high = (ql & 2) != 0;
#define XM_SWAP4(CTRL, V) make_uint4((unsigned)__builtin_amdgcn_update_dpp(0, (int)(V).x, CTRL, 0xf, 0xf, true), \
                                     (unsigned)__builtin_amdgcn_update_dpp(0, (int)(V).y, CTRL, 0xf, 0xf, true), \
                                     (unsigned)__builtin_amdgcn_update_dpp(0, (int)(V).z, CTRL, 0xf, 0xf, true), \
                                     (unsigned)__builtin_amdgcn_update_dpp(0, (int)(V).w, CTRL, 0xf, 0xf, true))
#pragma unroll
                for (int m = 0; m < 2; m++) {       // lane ^ 1: the off-diagonal pieces of every 2 x 2 block
                    const uint4 send = odd ? P[2 * m] : P[2 * m + 1];
                    const uint4 recv = XM_SWAP4(0xB1, send);
                    if (odd) P[2 * m] = recv; else P[2 * m + 1] = recv;
                }
#pragma unroll
                for (int c = 0; c < 2; c++) {       // lane ^ 2: the off-diagonal 2 x 2 blocks
                    const uint4 send = high ? P[c] : P[2 + c];
                    const uint4 recv = XM_SWAP4(0x4E, send);
                    if (high) P[c] = recv; else P[2 + c] = recv;
                }
#undef XM_SWAP4
                uint4 *out16 = reinterpret_cast<uint4 *>(A.midX);
#define XM_QSTORE(J) { const long long pj = __double_as_longlong(quad_bcast<J>(__longlong_as_double(pos)));                       \
                       const int vj = __builtin_amdgcn_update_dpp(0, valid ? 1 : 0, (J) | ((J) << 2) | ((J) << 4) | ((J) << 6), 0xf, 0xf, true); \
                       if (vj) out16[pj * 4 + ql] = P[J]; }
                XM_QSTORE(0) XM_QSTORE(1) XM_QSTORE(2) XM_QSTORE(3)
#undef XM_QSTORE
#endif
            });
        }
        done += total;
    }
    if (PHASE == 0 && threadIdx.x == 0) { ng[xpid] = (int)(done >> 40); nrec[xpid] = (long long)(done & ((1ull << 40) - 1)); }
}

#ifdef XMAP_CROSSCHECK      // (k_mid_dir: directory of the dense-table form)
// directory of the non-empty tiles of every x' (row of the dense table): count, then fill
template <bool FILL>
__global__ __launch_bounds__(256) void k_mid_dir(int n_nb, const int *tile_cnt, const long long *tile_off,
                                                 int *ng, const long long *dir_ptr, MidDir *dir, const int *nb_list,
                                                 const int *kcnt) {
    const int xpid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xpid >= n_nb) return;
    const int lane = lane_id();
    const size_t row = (size_t)xpid * n_nb;
    long long out = FILL ? dir_ptr[xpid] : 0;
    int total = 0;
    for (int b = 0; b < n_nb; b += 64) {
        const int xid = b + lane;
        const int c = (xid < n_nb) ? tile_cnt[row + xid] : 0;
        const unsigned long long m = __ballot(c > 0);
        if (FILL && c > 0) {
            MidDir d;
            d.x = nb_list[xid]; d.ne = 1 + kcnt[(size_t)d.x * 2 + 1]; d.cnt = c; d.pad = xid; d.off = tile_off[row + xid];
            dir[out + __popcll(m & lanemask_lt())] = d;
        }
        out += __popcll(m);
        total += __popcll(m);
    }
    if (!FILL && lane == 0) ng[xpid] = total;
}

#endif  // XMAP_CROSSCHECK
struct ColEnd { double sm, mu, f; int u; int pad; };     // one end of a column x: last edge (sim * mutu, mutu, frac; 0, 0, 1 for x itself), universe rank
struct Path2Args {
    PathArgs P;
    const ColEnd *cend;            // k_paths4: [n_nb][k + 1]
    const int *nb_id; const int *nb_list; int n_nb;
    const MidX *midX; const MidDir *dir; const long long *dir_ptr; const int *ng;
};

#ifdef XMAP_CROSSCHECK      // (flush_end: row update of k_paths2)
// merge a lane's register sums into the start's row (distinct ends per call)
__device__ __forceinline__ void flush_end(WaveAcc &W, bool active, int end, double s_hi, double s_lo, double c_hi, double c_lo) {
    bool first = false;
    if (active) {
        double *a = W.acc + (size_t)end * 4;
        double h0 = a[0], l0 = a[1], h1 = a[2], l1 = a[3];
        first = (h1 == 0.0);
        dd_add(h0, l0, s_hi); dd_add(h0, l0, s_lo);
        dd_add(h1, l1, c_hi); dd_add(h1, l1, c_lo);
        a[0] = h0; a[1] = l0; a[2] = h1; a[3] = l1;
    }
    unsigned long long m = __ballot(first);
    if (first) W.touched[W.nt + __popcll(m & lanemask_lt())] = end;
    W.nt += __popcll(m);
}

#endif  // XMAP_CROSSCHECK
// paths [start -] x' - t - s of one head (end s): lanes over the joint (t,s) of each t in NB_BB(x')
template <class ACC>
__device__ __forceinline__ void head_S(const PathArgs &A, ACC &W, int xp, bool has_e1, Carry e1) {
    const int lane = lane_id();
    const int nb = A.kcnt[(size_t)xp * 2];
    for (int q = 0; q < nb; q++) {
        const size_t o = ((size_t)xp * 2) * A.k + q;
        const int t = A.kcol[o];
        if (!(A.flags[t] & 2)) continue;
        const Carry c2 = has_e1 ? add_edge(e1, A.kval[o * 3], A.kval[o * 3 + 1], A.kval[o * 3 + 2])
                                : first_edge(A.kval[o * 3], A.kval[o * 3 + 1], A.kval[o * 3 + 2]);
        const long long s0 = A.src_ptr[t], s1 = A.src_ptr[t + 1];
        for (long long base = s0; base < s1; base += 64) {
            const long long p = base + lane;
            const bool act = (p < s1) && (A.src_flag[p] & 1);
            int s = 0;
            Carry c = c2;
            if (act) {
                s = A.src_idx[p];
                c = add_edge(c2, A.src_val[p * 3], A.src_val[p * 3 + 1], A.src_val[p * 3 + 2]);
            }
            W.add(act, s, c);
        }
    }
}

#ifdef XMAP_CROSSCHECK      // k_paths2 (round 1's tile-major enumeration, algo="mid"): a test formulation, libxmap_hip_xcheck.so only
// Tile-major reduction over the heads of one start.  Up to 64 heads (one per lane) are merged by item x: every
// head's tile directory is sorted by x, so the smallest current x over the lanes is the next tile column; all heads
// that own a tile (x', x) for it are reduced into the SAME register sums before the start's row is touched -- one
// row access per (start, x) instead of one per (head, x) (2.1x fewer at BASELINE configs[1], 25x for the starts
// with many heads).  [xlo, xhi) restricts the columns (work splitting of heavy starts).
__device__ __forceinline__ void heads_X(const Path2Args &B, WaveAcc &W, int start, long long h0, long long nH, int self,
                                        int xlo, int xhi) {
    const PathArgs &A = B.P;
    const int lane = lane_id();
    const int k = A.k;
    const int INF = 0x7fffffff;
    // this lane's head
    const long long h = h0 + lane;
    const bool hv = h < nH;
    double sm1 = 0.0, mu1 = 0.0, f1 = 1.0;
    bool has_e1 = false;
    long long dpos = 0, dend = 0;
    if (hv) {
        int xp = start;
        if (h >= self) {
            const long long rp = A.rnn_ptr[start] + (h - self);
            xp = A.rnn_idx[rp];
            const double sv = A.rnn_val[rp * 3], mu = A.rnn_val[rp * 3 + 1];
            sm1 = sv * mu; mu1 = mu; f1 = A.rnn_val[rp * 3 + 2];
            has_e1 = true;
        }
        const int xpid = B.nb_id[xp];
        dpos = B.dir_ptr[xpid];
        dend = B.dir_ptr[xpid + 1];
        if (xlo > 0) {   // lower bound of xlo in this head's directory (sorted by x)
            long long lo = dpos, hi = dend;
            while (lo < hi) {
                long long mid = (lo + hi) >> 1;
                if (B.dir[mid].x < xlo) lo = mid + 1; else hi = mid;
            }
            dpos = lo;
        }
    }
    MidDir cur;
    cur.x = INF; cur.ne = 0; cur.cnt = 0; cur.pad = 0; cur.off = 0;
    if (hv && dpos < dend) { cur = B.dir[dpos]; if (cur.x >= xhi) cur.x = INF; }
    for (;;) {
        int xmin = cur.x;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(xmin, m, 64); xmin = o < xmin ? o : xmin; }
        if (xmin == INF) break;
        const unsigned long long part = __ballot(cur.x == xmin);
        const int x = xmin;
        const int ne = rl32(cur.ne, __ffsll((long long)part) - 1);
        for (int b = 0; b < ne; b += 64) {
            const int idx = b + lane;
            const bool act = idx < ne;
            int end = x;
            double sm5 = 0.0, mu5 = 0.0, f5 = 1.0;
            const bool has5 = act && idx > 0;
            if (has5) {
                size_t o = ((size_t)x * 2 + 1) * k + (idx - 1);
                end = A.kcol[o];
                const double v = A.kval[o * 3], m = A.kval[o * 3 + 1];
                sm5 = v * m; mu5 = m; f5 = A.kval[o * 3 + 2];
            }
            double s_hi = 0.0, s_lo = 0.0, c_hi = 0.0, c_lo = 0.0;
            unsigned long long np = 0;
            unsigned long long pm = part;
            while (pm) {
                const int l = __ffsll((long long)pm) - 1;
                pm &= pm - 1;
                const int cnt = rl32(cur.cnt, l);
                const long long off = rl64(cur.off, l);
                const bool he1 = rl32((int)has_e1, l) != 0;
                const double hsm1 = rld(sm1, l), hmu1 = rld(mu1, l), hf1 = rld(f1, l);
                np += (unsigned long long)cnt;
                for (int r0 = 0; r0 < cnt; r0 += 64) {
                    MidX m;
                    m.sm2 = m.sm3 = m.sm4 = m.f2 = m.f3 = m.f4 = m.mu = 0.0;
                    if (r0 + lane < cnt) m = B.midX[off + r0 + lane];
                    const int nr = (cnt - r0) < 64 ? (cnt - r0) : 64;
                    // the part of a path's value that does not depend on the end is computed once per (head, record),
                    // on the record's lane (same operations in the same order as the per-path statement), and three
                    // doubles instead of seven are broadcast per step
                    double bsm, bc;
                    if (he1) { bsm = ((hsm1 + m.sm2) + m.sm3) + m.sm4; bc = ((hf1 * m.f2) * m.f3) * m.f4; }
                    else { bsm = (m.sm2 + m.sm3) + m.sm4; bc = (m.f2 * m.f3) * m.f4; }
                    const double bmu = m.mu + (he1 ? hmu1 : 0.0);
                    for (int r = 0; r < nr; r++) {
                        double sm = rld(bsm, r), c = rld(bc, r), mu = rld(bmu, r);
                        if (has5) { sm = sm + sm5; c = c * f5; mu = mu + mu5; }
                        const double sp = (mu != 0.0) ? 1.0 * sm / mu : 0.0;
                        dd_add(s_hi, s_lo, sp * c);
                        dd_add(c_hi, c_lo, c);
                    }
                }
            }
            flush_end(W, act, end, s_hi, s_lo, c_hi, c_lo);
            W.paths += np * (unsigned long long)__popcll(__ballot(act));
        }
        if (cur.x == xmin) {    // advance the heads that took part
            dpos++;
            cur.x = INF;
            if (dpos < dend) { cur = B.dir[dpos]; if (cur.x >= xhi) cur.x = INF; }
        }
    }
}

// 5 waves per SIMD (94 VGPRs, 68 B of scratch per lane) measured 6 % faster than the 4 the unconstrained allocation
// (112 VGPRs) allows, 6 (80 VGPRs, 128 B of scratch) 8 % slower: the kernel is bound by its random row updates, more
// waves keep more of them in flight
#ifdef B_TRACE
__device__ unsigned long long g_btrace[1 << 20][2];   // per unit: begin, end (wall_clock64, 100 MHz)
#endif
#ifndef B_WAVES
#define B_WAVES 5
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(B_WAVES, B_WAVES))) void k_paths2(Path2Args B) {
    __shared__ FinBuf fin[4];
    const PathArgs &A = B.P;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.n_slots) return;
    const int lane = lane_id();
    WaveAcc W;
    W.paths = 0;
    unsigned long long cand_total = 0;
    for (;;) {
        int u_ = 0;
        if (lane == 0) u_ = (int)atomicAdd(&A.counters[2], 1ull);
        const int unit = uniform(u_);
        if (unit >= A.n_units) break;  // every wave reaches this exit: the cursor only grows
        const int start = uniform(A.unit_start[unit]);
        const int c = uniform(A.unit_c[unit]);
        const int G = uniform(A.unit_G[unit]);
        const int row = uniform(A.unit_row[unit]);
#ifdef B_TRACE
        const unsigned long long tr0 = wall_clock64();
#endif
        if (row < 0) {
            W.acc = A.acc + (size_t)slot * A.I * 4;
            W.touched = A.touched + (size_t)slot * A.I;
        } else {
            W.acc = A.hacc + (size_t)row * A.I * 4;
            W.touched = A.htouched + (size_t)row * A.I;
        }
        W.nt = 0;
        int ent = 0;  // work entries of a start: role T; per head its (t,s) part; per (64-head batch, column range) the tiles
        if (A.flags[start] & 2) {   // role T: non-joint paths from t = start (final_nonjoint_extend, extender.py:124-140,:180)
            if (G == 1 || ent % G == c) {
                Carry none; none.sm = 0; none.mu = 0; none.c = 0;
                through_t(A, W, start, false, none);
            }
            ent++;
        }
        const long long r0 = uniform((int)A.rnn_ptr[start]), r1 = uniform((int)A.rnn_ptr[start + 1]);
        const int self = (A.cls[start] == 2) ? 1 : 0;   // head 0 = the start itself (target_path, extender.py:160-163)
        const long long nH = self + (r1 - r0);          // heads >= self: start in NN(x') (longest_path, :164-167)
        for (long long h = 0; h < nH; h++) {
            if (G == 1 || ent % G == c) {
                const bool has_e1 = h >= self;
                const int xp = has_e1 ? A.rnn_idx[r0 + h - self] : start;
                Carry e1; e1.sm = 0; e1.mu = 0; e1.c = 1.0;
                if (has_e1) e1 = first_edge(A.rnn_val[(r0 + h - self) * 3], A.rnn_val[(r0 + h - self) * 3 + 1],
                                            A.rnn_val[(r0 + h - self) * 3 + 2]);
                head_S(A, W, xp, has_e1, e1);
            }
            ent++;
        }
        const long long nbatch = (nH + 63) / 64;
        const int RX = (nbatch > 0) ? (int)((G + nbatch - 1) / nbatch) : 1;   // column ranges: nbatch * RX >= G entries
        const int n_nb = B.n_nb;
        for (long long bt = 0; bt < nbatch; bt++)
            for (int rx = 0; rx < RX; rx++) {
                if (G == 1 || ent % G == c) {
                    const int xlo = (rx == 0) ? 0 : B.nb_list[(long long)rx * n_nb / RX];
                    const int xhi = (rx == RX - 1) ? 0x7fffffff : B.nb_list[(long long)(rx + 1) * n_nb / RX];
                    heads_X(B, W, start, bt * 64, nH, self, xlo, xhi);
                }
                ent++;
            }
        if (row < 0) cand_total += finalize_start(A, fin[threadIdx.x >> 6], W.acc, W.touched, W.nt, start);
        else if (lane == 0) A.unit_nt[unit] = W.nt;
#ifdef B_TRACE
        if (lane == 0 && unit < (1 << 20)) { g_btrace[unit][0] = tr0; g_btrace[unit][1] = wall_clock64(); }
#endif
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], cand_total);
        atomicAdd(&A.counters[1], W.paths);
    }
}

#endif  // XMAP_CROSSCHECK

// ---- helpers of k_paths4 -------------------------------------------------------------------------------------
// a / b rounded to nearest for b > 0 and operands far from the ends of the exponent range: v_rcp_f64 + two Newton steps
// + one correction of the quotient, i.e. the sequence the compiler emits for `/` without v_div_scale / v_div_fmas'
// rescaling / v_div_fixup (which only act on operands near the ends of the range, zero, inf or nan)
__device__ __forceinline__ double div_mid(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// two-sum, rounding errors collected in lo (not renormalised: hi + lo is the sum to ~2^-104 like dd_add's pair)
__device__ __forceinline__ void acc2(double &hi, double &lo, double x) {
    const double s = hi + x;
    const double bb = s - hi;
    lo += (hi - (s - bb)) + (x - bb);
    hi = s;
}

// exchange inside a group of four adjacent lanes (DPP quad_perm: no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double quad_swap(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}


// =============================================================================================
// k_paths4 (default).  What the ablations of k_paths2 / k_paths3 at BASELINE configs[1] say (profiles/README.md, round 2):
// a column (start, x) costs a fixed price -- merge step, end list, row update, bookkeeping -- that outweighs its
// arithmetic (10 records x 22 ends on average), and the row updates are random 32-byte read-modify-writes.  Hence:
//   * ONE row update per column: the lanes of a step are  W ends x S record slices  with S = 4 / 2 / 1 for a column of
//     <= 16 / <= 32 / more ends, so every column is a single set of lanes whatever its width; the S slices of an end
//     are adjacent lanes and are added up by one or two DPP exchanges;
//   * rows are indexed by the rank of the end among the U items that can end a path at all (xmap_end_universe; ranks
//     in column order, so that the ends of a column are neighbours in the row): 2.7x shorter rows, 5x less scratch;
//   * the ends of a column come from one table of 32-byte records (k_col_ends: rank and last edge), not from three
//     dependent gathers; the row entries are requested before the records are reduced;
//   * prepared records of all participating heads are staged in LDS (128 per round) in sets of 64, one record per lane
//     whichever head it belongs to: the records of all the heads of a column are ONE round trip (a trip per head had been
//     2.5 dependent trips per column), requested together with the end records; the row entries are requested next, before
//     the records are prepared.  Loads are unconditional (clamped indices) and consumed at unconditional places -- see the
//     comment in heads_Q; division and sums as in k_paths3.
// Round 4: heads_Q reads both tables of a column as 16-byte pieces, one per lane (profiles/r04d_paths_pieces.txt: -2.7 %);
// -DQ_WHOLE: whole records per lane as in rounds 2-3 (also what the pipelined experiment heads_P uses)
#if !defined(Q_WHOLE) && !defined(Q_PIPE)
#define Q_RPIECE 1
#define Q_EPIECE 1
#endif
#ifdef Q_RPIECE
constexpr int Q_CAP = 64;                  // prepared records per round (the head records take the LDS of the other 64)
#else
constexpr int Q_CAP = 128;                 // prepared records per round
#endif
struct QLds {
    // (three arrays, not one record of four words per quad as the quad leaves them: that layout -- one 512-byte store, b64 +
    //  b128 reads at a 32-byte stride in the record loop -- measured 515 ms against 475, profiles/r04h_paths_columns.txt)
    double bsm[Q_CAP + 1], bc[Q_CAP + 1], bmu[Q_CAP + 1];      // (entry Q_CAP: the neutral record (0, 0, 1) of heads_Q's record loop)
#ifdef Q_RPIECE
    double hd[3][64];                      // first edge of every head of the batch: [0] sim * mutu, [1] frac, [2] mutu
#endif
#ifdef Q_EPIECE
    uint4 ep[128];                         // the chunk's end records as loaded: 16-byte pieces, two per end (the table's own layout)
#else
    double e_sm[64], e_mu[64], e_f[64];
    int e_u[64];                           // universe rank of the end, -1 = none
#endif
};

struct QAcc {
    double *acc; int *touched;             // the unit's row [U][4] and touched list [U]
    const int *urank;
    int nt;
    bool slot_rows;                        // (-DQ_STORE ablation: this unit's row has one slot per (column, end))
    double *junk;                          // 64 bytes nobody reads: where the lanes without an update store (heads_P)
#ifdef P_TRACE                             // (profiling build: shader cycles per segment of the column loop, summed per wave)
    unsigned long long pt[12], pt_last;
#endif
    unsigned long long paths;
    unsigned long long updates;            // read-modify-writes of row entries (the kernel's bound: DESIGN.md 4)
    __device__ __forceinline__ void add(bool active, int end, Carry p) {
        bool first = false;
        int u = 0;
        if (active) {
            u = urank[end];
            const double sp = (p.mu != 0.0) ? 1.0 * p.sm / p.mu : 0.0;   // calculate_path_confidence (extender.py:83-89)
            double *a = acc + (size_t)u * 4;
            double s_hi = a[0], s_lo = a[1], c_hi = a[2], c_lo = a[3];
            first = (c_hi == 0.0);
            acc2(s_hi, s_lo, sp * p.c);
            acc2(c_hi, c_lo, p.c);
            a[0] = s_hi; a[1] = s_lo; a[2] = c_hi; a[3] = c_lo;
        }
        const unsigned long long m = __ballot(first);
        if (first) touched[nt + __popcll(m & lanemask_lt())] = u;
        nt += __popcll(m);
        const int na = __popcll(__ballot(active));
        paths += na;
        updates += na;
    }
};

// An all-zero row entry: where the lanes of HOME ends read their "old" value from.  The home column of an end is the lowest
// column that lists it (ColEnd::u bit 30, k_col_home); columns are visited in ascending order and a unit visits its columns
// before anything else touches its row, so in the first head batch an end's entry is still zero when its home column
// comes by -- the update of a home end needs no load from the row: its lanes read this one cached line instead (the add
// of zero is exact, `first` comes out true by itself), and what the memory system sees is a store.
__device__ double g_zero_entry[4] = {0.0, 0.0, 0.0, 0.0};
__device__ double g_junk[8192][8];        // per wave slot (heads_P): target of the lanes that have nothing to store
#ifdef P_TRACE
__device__ unsigned long long g_ptrace[16];
#define PT(i) { const unsigned long long t_ = clock64(); W.pt[i] += t_ - W.pt_last; W.pt_last = t_; }
#else
#define PT(i)
#endif
constexpr int END_HOME = 1 << 30;
#ifdef Q_HIST                              // (probe build, profiles/tools/col_hist.py: column visits by heads / ends / records)
__device__ unsigned long long g_qhist[128];
#endif

// ONE (-DQ_ONE, the review's lever 2b): the batch is a start's only head (58.5 % of the column visits, profiles/r04h_paths_columns.txt)
// -- no merge step (the head's next column IS the column), no lane assignment loop (its records are one run), the head's first
// edge a per-lane constant instead of an LDS read per group
template <bool FASTDIV>
__device__ __forceinline__ void heads_Q(const Path2Args &B, QAcc &W, int start, long long h0, long long nH, int self, int xlo, int xhi,
                                        bool fresh) {
    __shared__ QLds stageq[4];          // one per wave of the block; DS operations of a wave execute in order
    const PathArgs &A = B.P;
    QLds &L = stageq[threadIdx.x >> 6];
    const int lane = lane_id();
    const int k = A.k;
    const int INF = 0x7fffffff;
    // this lane's head
    const long long h = h0 + lane;
    const bool hv = h < nH;
    double sm1 = 0.0, mu1 = 0.0, f1 = 1.0;      // (the neutral first edge: the start itself as head)
#ifdef Q_HE1
    bool has_e1 = false;
#endif
    long long dpos = 0, dend = 0;
    if (hv) {
        int xp = start;
        if (h >= self) {
            const long long rp = A.rnn_ptr[start] + (h - self);
            xp = A.rnn_idx[rp];
            const double sv = A.rnn_val[rp * 3], mu = A.rnn_val[rp * 3 + 1];
            sm1 = sv * mu; mu1 = mu; f1 = A.rnn_val[rp * 3 + 2];
#ifdef Q_HE1
            has_e1 = true;
#endif
        }
        const int xpid = B.nb_id[xp];
        dpos = B.dir_ptr[xpid];
        dend = B.dir_ptr[xpid + 1];
        if (xlo > 0) {   // lower bound of xlo in this head's directory (sorted by x)
            long long lo = dpos, hi = dend;
            while (lo < hi) {
                long long mid = (lo + hi) >> 1;
                if (B.dir[mid].x < xlo) lo = mid + 1; else hi = mid;
            }
            dpos = lo;
        }
    }
    MidDir cur;
    cur.x = INF; cur.ne = 0; cur.cnt = 0; cur.pad = 0; cur.off = 0;
    if (hv && dpos < dend) { cur = B.dir[dpos]; if (cur.x >= xhi) cur.x = INF; }
#ifdef Q_NO_DPP_MERGE
    const int nloc = 64;
#else
    const int nloc = (nH - h0) < 64 ? (int)(nH - h0) : 64;      // heads of this batch (lanes 0 .. nloc-1)
#endif
#ifdef Q_ONE
    const bool ONE = nloc == 1;         // (a wave-uniform branch in the same loop body: a second instantiation of the loop cost 30 %)
#else
    constexpr bool ONE = false;
#endif
    if (lane == 0) { L.bsm[Q_CAP] = 0.0; L.bc[Q_CAP] = 0.0; L.bmu[Q_CAP] = 1.0; }       // the neutral record of the record loop
#ifdef Q_RPIECE
    // Merged records as 16-byte PIECES (record r = the four lanes 4r .. 4r+3 = {sm2, sm3}, {sm4, f2}, {f3, f4}, {mu, -}): one
    // load instruction per group of 16 records instead of four over the same lines; the first edge of a record's head comes
    // from a 24-byte LDS record per head (one ds_read_b64 per lane: lane 0 of a quad needs sim * mutu, lane 2 frac, lane 3 mutu)
    L.hd[0][lane] = sm1; L.hd[1][lane] = f1; L.hd[2][lane] = mu1;
    const uint4 *recp = reinterpret_cast<const uint4 *>(B.midX);
    const int pq = lane & 3, prec = lane >> 2;
    const int pfield = pq == 0 ? 0 : (pq == 2 ? 1 : 2);
    // (ONE: the only head's first edge, the field this lane of a quad needs)
    const double H1s = __longlong_as_double(rl64(__double_as_longlong(sm1), 0)), H1f = __longlong_as_double(rl64(__double_as_longlong(f1), 0));
    const double H1m = __longlong_as_double(rl64(__double_as_longlong(mu1), 0));
    const double H1 = pq == 0 ? H1s : (pq == 2 ? H1f : H1m);
#endif
    for (;;) {
        // smallest column among the heads: xor butterfly inside each half of the wave (ds_swizzle: no address registers),
        // then the two halves
        int xmin = ONE ? rl32(cur.x, 0) : cur.x;
#define XM_DPP_MIN(CTRL) { const int o = __builtin_amdgcn_update_dpp(INF, xmin, CTRL, 0xf, 0xf, false); xmin = o < xmin ? o : xmin; }
#define XM_SWZ_MIN(PAT) { const int o = __builtin_amdgcn_ds_swizzle(xmin, PAT); xmin = o < xmin ? o : xmin; }
        if (ONE) {
        } else if (nloc <= 16) {      // the common batch of a few heads: DPP inside the first row of lanes, no LDS round trips
            XM_DPP_MIN(0xB1) XM_DPP_MIN(0x4E)                               // quad_perm [1,0,3,2], [2,3,0,1]
            if (nloc > 4) { XM_DPP_MIN(0x141) XM_DPP_MIN(0x140) }           // row_half_mirror, row_mirror
            xmin = rl32(xmin, 0);
        } else {
            XM_SWZ_MIN(0x041F) XM_SWZ_MIN(0x081F) XM_SWZ_MIN(0x101F) XM_SWZ_MIN(0x201F) XM_SWZ_MIN(0x401F)
            const int x0 = rl32(xmin, 0), x1 = rl32(xmin, 32);
            xmin = x0 < x1 ? x0 : x1;
        }
#undef XM_SWZ_MIN
#undef XM_DPP_MIN
        if (xmin == INF) break;
        const bool mine = ONE ? lane == 0 : cur.x == xmin;
        const unsigned long long part = ONE ? 1ull : __ballot(mine);
        const int first_l = ONE ? 0 : __ffsll((long long)part) - 1;
        const int ne = rl32(cur.ne, first_l);
        const ColEnd *ce = B.cend + (size_t)rl32(cur.pad, first_l) * (k + 1);
#ifdef Q_SHN
        // (-DQ_SHN, the review's lever 2a) slices from the RECORDS as well as from the ends: a slice level more is an exchange
        // (22 instructions) and saves steps (28 each) only from three records on; the count is at hand when one head has the
        // column.  30 % of the column visits have one or two records (profiles/r04h_paths_columns.txt) -- and the kernel takes
        // 478 ms with it against 475 without: not the default
        const int n1col = (part & (part - 1)) == 0 ? rl32(cur.cnt, first_l) : Q_CAP;
        const int shn = n1col <= 1 ? 0 : (n1col <= 2 ? 1 : 2);
#endif
        for (int b = 0; b < ne; b += 64) {
            const int nact = (ne - b) < 64 ? (ne - b) : 64;
#ifdef Q_SHN
            const int she = nact <= 16 ? 2 : (nact <= 32 ? 1 : 0);
            const int sh = she < shn ? she : shn;
#else
            const int sh = nact <= 16 ? 2 : (nact <= 32 ? 1 : 0);     // log2 of the record slices per end
#endif
            const int ns = 1 << sh;
            const int q = lane >> sh, slice = lane & (ns - 1);
            // Memory round trips of a column: {end records, first set of merged records} together, then the row entries
            // (under the preparation and reduction of the records).  Every load is unconditional (clamped index) and is
            // consumed at one unconditional place: a load whose use sits behind a branch stays "pending" on the other
            // path, and the compiler then waits for ALL outstanding loads at the next join, which serialises the trips.
#ifdef Q_EPIECE
            // the end records as 16-byte PIECES, one per lane (two per end): ONE load instruction for up to 32 ends where the
            // whole-record form needs two over the same lines; the pieces go to LDS as they come and are read back per (end, slice)
            const int np = 2 * nact;
            const uint4 *cep = reinterpret_cast<const uint4 *>(ce) + 2 * b;
            const uint4 e0 = cep[lane < np ? lane : np - 1];
#else
            const ColEnd e = ce[(b + lane < ne) ? b + lane : ne - 1];      // one 32-byte record per end (k_col_ends)
#endif
            unsigned long long pm = part;
            int pos = 0;                      // records of the current head already staged
            int set_n, my_h;
            long long my_rec;
#ifdef Q_RPIECE
            // a group = up to 16 records, four lanes each, of whichever participating heads they fall to
            auto assign = [&]() {
                if (ONE) {      // the head's records are one run: the next (up to) 16 of them
                    const int cnt = rl32(cur.cnt, 0);
                    const long long off = rl64(cur.off, 0);
                    int n = cnt - pos;
                    if (n > 16) n = 16;
                    my_h = 0; set_n = n;
                    my_rec = (off + pos) * 4 + (lane < 4 * n ? lane : 0);
                    pos += n;
                    if (pos == cnt) { pm = 0; pos = 0; }
                    return;
                }
                set_n = 0; my_h = 0;
                my_rec = rl64(cur.off, __ffsll((long long)pm) - 1) * 4;     // lanes beyond the group: any piece
                for (;;) {
                    pm = ((unsigned long long)(unsigned)uniform((int)(pm >> 32)) << 32) | (unsigned)uniform((int)pm);
                    pos = uniform(pos); set_n = uniform(set_n);
                    if (pm == 0 || set_n >= 16) break;
                    const int l = __ffsll((long long)pm) - 1;
                    const int cnt = rl32(cur.cnt, l);
                    const long long off = rl64(cur.off, l);
                    int n = cnt - pos;
                    if (n > 16 - set_n) n = 16 - set_n;
                    if (lane >= 4 * set_n && lane < 4 * (set_n + n)) { my_rec = (off + pos) * 4 + (lane - 4 * set_n); my_h = l; }
                    set_n += n; pos += n;
                    if (pos == cnt) { pm &= pm - 1; pos = 0; }
                }
            };
            // prepared form of a group's records, computed inside the quad (same operations in the same order)
            auto prepare = [&](const uint4 &v, int fill) {
                const double H = ONE ? H1 : L.hd[pfield][my_h];
                const double X = __longlong_as_double(((long long)v.y << 32) | v.x), Y = __longlong_as_double(((long long)v.w << 32) | v.z);
                const double X1 = quad_bcast<1>(X), Y1 = quad_bcast<1>(Y);      // sm4, f2 of the record
                const double v_sm = ((H + X) + Y) + X1;                           // lane 0: ((sm1 + sm2) + sm3) + sm4
                const double v_c = ((H * Y1) * X) * Y;                            // lane 2: ((f1 * f2) * f3) * f4
                const double v_mu = X + H;                                        // lane 3: mu + mu1
                const double val = pq == 0 ? v_sm : (pq == 2 ? v_c : v_mu);
                if (pq != 1 && prec < set_n) { if (pq == 0) L.bsm[fill + prec] = val; else if (pq == 2) L.bc[fill + prec] = val; else L.bmu[fill + prec] = val; }
            };
            assign();
            uint4 m0 = recp[my_rec];
#else
            // a set = up to 64 records, one per lane, of whichever participating head the lane falls to
            auto assign = [&]() {
                set_n = 0; my_h = 0;
                my_rec = rl64(cur.off, __ffsll((long long)pm) - 1);     // lanes beyond the set: any record
                for (;;) {
                    // (pm, pos and set_n are wave-uniform; said explicitly, or the loop is compiled as a divergent one)
                    pm = ((unsigned long long)(unsigned)uniform((int)(pm >> 32)) << 32) | (unsigned)uniform((int)pm);
                    pos = uniform(pos); set_n = uniform(set_n);
                    if (pm == 0 || set_n >= 64) break;
                    const int l = __ffsll((long long)pm) - 1;
                    const int cnt = rl32(cur.cnt, l);
                    const long long off = rl64(cur.off, l);
                    int n = cnt - pos;
                    if (n > 64 - set_n) n = 64 - set_n;
                    if (lane >= set_n && lane < set_n + n) { my_rec = off + pos + (lane - set_n); my_h = l; }
                    set_n += n; pos += n;
                    if (pos == cnt) { pm &= pm - 1; pos = 0; }
                }
            };
            // prepared form of a record: first edge of its head (from the head's lane) + the three middle edges
            auto prepare = [&](const MidX &m, int fill) {
                const double hsm1 = __shfl(sm1, my_h, 64), hmu1 = __shfl(mu1, my_h, 64), hf1 = __shfl(f1, my_h, 64);
                asm volatile("" :: "v"(m.sm2), "v"(m.sm3), "v"(m.sm4), "v"(m.f2), "v"(m.f3), "v"(m.f4), "v"(m.mu));
#ifdef Q_HE1        // (rounds 2-3: separate formulas for a head without a first edge, picked by a flag from the head's lane)
                const bool he1 = __shfl((int)has_e1, my_h, 64) != 0;
                double bsm, bc;
                if (he1) { bsm = ((hsm1 + m.sm2) + m.sm3) + m.sm4; bc = ((hf1 * m.f2) * m.f3) * m.f4; }
                else { bsm = (m.sm2 + m.sm3) + m.sm4; bc = (m.f2 * m.f3) * m.f4; }
                const double bmu = m.mu + (he1 ? hmu1 : 0.0);
#else
                // a head without a first edge (the start itself) carries the NEUTRAL edge (0, 0, 1): 0 + x and 1 * x are exact,
                // so one formula serves both kinds of head with the same bits (sim * mutu is never a zero: the zero filter)
                const double bsm = ((hsm1 + m.sm2) + m.sm3) + m.sm4;
                const double bc = ((hf1 * m.f2) * m.f3) * m.f4;
                const double bmu = m.mu + hmu1;
#endif
                if (lane < set_n) { L.bsm[fill + lane] = bsm; L.bc[fill + lane] = bc; L.bmu[fill + lane] = bmu; }
            };
            assign();
            MidX m0 = B.midX[my_rec];
#endif
#ifdef Q_EPIECE
            if (nact > 32) L.ep[64 + lane] = cep[64 + lane < np ? 64 + lane : np - 1];
            L.ep[lane] = e0;
            asm volatile("" ::: "memory");
            const bool ok = q < nact;
            const uint4 ea = L.ep[2 * (ok ? q : 0)], eb = L.ep[2 * (ok ? q : 0) + 1];
            const double sm5 = __longlong_as_double(((long long)ea.y << 32) | ea.x), mu5 = __longlong_as_double(((long long)ea.w << 32) | ea.z);
            const double f5 = __longlong_as_double(((long long)eb.y << 32) | eb.x);
            const int eur = (int)eb.z;
            const int eu = eur & (END_HOME - 1);
#else
            // ends: natural layout -> LDS -> (end, slice) layout
            {
                int eu_ = e.u;
                asm volatile("" : "+v"(eu_));          // (keeps the select below, and with it the wait for e, down here)
                L.e_u[lane] = (b + lane < ne) ? eu_ : -1; L.e_sm[lane] = e.sm; L.e_mu[lane] = e.mu; L.e_f[lane] = e.f;
            }
            asm volatile("" ::: "memory");
            const int eur = L.e_u[q];
            const bool ok = eur >= 0;
            const int eu = eur & (END_HOME - 1);
            const double sm5 = L.e_sm[q], mu5 = L.e_mu[q], f5 = L.e_f[q];
#endif
            const bool fl_ = ok && slice == 0;
#ifdef Q_STORE      // (ablation: every (column, end) a slot of its own -- a column visit is one contiguous run of stores, no load;
            //  the rows of the split heavy starts keep their U entries: the usual place there)
            double *a = W.acc + (W.slot_rows ? ((size_t)rl32(cur.pad, first_l) * (k + 1) + (size_t)(b + q)) : (size_t)(ok ? eu : 0)) * 4;
#else
            double *a = W.acc + (size_t)(ok ? eu : 0) * 4;
#endif
#ifdef Q_NOHOME
            const double *la = a;
#else
            const double *la = (ok && fresh && (eur & END_HOME)) ? g_zero_entry : a;
#endif
            for (;;) {
                // one round = up to Q_CAP prepared records (a column with more, < 1 % of them, updates its row once per round)
                double h0_ = 0.0, l0_ = 0.0, h1_ = 0.0, l1_ = 0.0;
#if !defined(Q_NOFLUSH) && !defined(Q_STORE) && !defined(Q_STOREU)
                h0_ = la[0]; l0_ = la[1]; h1_ = la[2]; l1_ = la[3];     // requested before the records are prepared and reduced
                la = a;                                                 // (a second round of the same column finds the first one's sums)
#endif
#ifdef Q_RPIECE
                int fill = 0;
                {   // up to four groups requested together (one trip for up to 64 records), then prepared
                    const int h0g = my_h, g0 = uniform(set_n);
                    uint4 m1 = m0, m2 = m0, m3 = m0;
                    int h1g = 0, h2g = 0, h3g = 0, g1 = 0, g2 = 0, g3 = 0;
                    if (pm) { assign(); m1 = recp[my_rec]; h1g = my_h; g1 = uniform(set_n); }
                    if (pm) { assign(); m2 = recp[my_rec]; h2g = my_h; g2 = uniform(set_n); }
                    if (pm) { assign(); m3 = recp[my_rec]; h3g = my_h; g3 = uniform(set_n); }
                    my_h = h0g; set_n = g0; prepare(m0, fill); fill += g0;
                    if (g1) { my_h = h1g; set_n = g1; prepare(m1, fill); fill += g1; }
                    if (g2) { my_h = h2g; set_n = g2; prepare(m2, fill); fill += g2; }
                    if (g3) { my_h = h3g; set_n = g3; prepare(m3, fill); fill += g3; }
                    fill = uniform(fill);
                }
#else
                prepare(m0, 0);
                int fill = set_n;
                while (pm && fill < Q_CAP) {
                    assign();
                    const MidX m = B.midX[my_rec];
                    prepare(m, fill);
                    fill += set_n;
                }
#endif
                asm volatile("" ::: "memory");
#ifdef Q_HIST
                if (lane == 0) {
                    const int c_n = fill <= 1 ? 0 : (fill <= 2 ? 1 : (fill <= 4 ? 2 : (fill <= 8 ? 3 : (fill <= 16 ? 4 : (fill <= 32 ? 5 : 6)))));
                    const int c_e = nact <= 4 ? 0 : (nact <= 8 ? 1 : (nact <= 16 ? 2 : (nact <= 32 ? 3 : 4)));
                    const int c_h = nloc == 1 ? 0 : (__popcll(part) == 1 ? 1 : 2);
                    atomicAdd(&g_qhist[(c_h * 5 + c_e) * 7 + c_n], 1ull);
                }
#endif
                double a_sh = 0.0, a_sl = 0.0, a_ch = 0.0, a_cl = 0.0;
                const int steps = (fill + ns - 1) >> sh;
#if defined(Q_NOCOMPUTE)
                if (ok && slice < fill) { a_sh = 1.0; a_ch = 1.0; }
#elif defined(Q_STEPLOOP1)      // (the record loop of rounds 2-3: one record per iteration, lanes without a record masked)
                for (int it = 0; it < steps; it++) {
                    const int r = (it << sh) + slice;
                    if (ok && r < fill) {
                        const double sm = L.bsm[r] + sm5;
                        const double c = L.bc[r] * f5;
                        const double mu = L.bmu[r] + mu5;
                        double sp;
                        if (FASTDIV) sp = div_mid(sm, mu);
                        else sp = (mu != 0.0) ? 1.0 * sm / mu : 0.0;      // calculate_path_confidence (extender.py:83-89)
                        acc2(a_sh, a_sl, sp * c);
                        acc2(a_ch, a_cl, c);
                    }
                }
#else
                // The record loop without a branch and two records per iteration (round 4).  A lane whose slice has no
                // record in a step takes the NEUTRAL record (0, 0, 1): its path weight c = 0 * f is zero, so both sums get
                // + 0 -- exact, no effect -- and the compare / mask / skip-branch instructions of a step are gone; lanes beyond
                // the column's ends compute on the last end's record (never stored).  Two records per iteration: no
                // loop-carried register copies, one LDS round trip for both.
                auto step = [&](double rsm, double rc, double rmu) {
                    const double sm = rsm + sm5;
                    const double c = rc * f5;
                    const double mu = rmu + mu5;
                    double sp;
                    if (FASTDIV) sp = div_mid(sm, mu);
                    else sp = (mu != 0.0) ? 1.0 * sm / mu : 0.0;          // calculate_path_confidence (extender.py:83-89)
                    acc2(a_sh, a_sl, sp * c);
                    acc2(a_ch, a_cl, c);
                };
                int it = 0;
                for (; it + 1 < steps; it += 2) {
                    const int ra = (it << sh) + slice, rb = ra + ns;      // ra < fill in every step but a round's last
                    const int rbc = rb < fill ? rb : Q_CAP;
                    const double s0 = L.bsm[ra], c0 = L.bc[ra], u0 = L.bmu[ra];
                    const double s1 = L.bsm[rbc], c1 = L.bc[rbc], u1 = L.bmu[rbc];
                    asm volatile("" :: "v"(s0), "v"(c0), "v"(u0), "v"(s1), "v"(c1), "v"(u1));      // (both records: one LDS round trip)
                    step(s0, c0, u0);
                    step(s1, c1, u1);
                }
                if (it < steps) {
                    const int ra = (it << sh) + slice;
                    const int rc_ = ra < fill ? ra : Q_CAP;
                    step(L.bsm[rc_], L.bc[rc_], L.bmu[rc_]);
                }
#endif
                // the slices of an end sit in adjacent lanes
                if (sh >= 1) {
                    const double o_sh = quad_swap<0xB1>(a_sh), o_sl = quad_swap<0xB1>(a_sl);      // lane ^ 1
                    const double o_ch = quad_swap<0xB1>(a_ch), o_cl = quad_swap<0xB1>(a_cl);
                    acc2(a_sh, a_sl, o_sh); a_sl += o_sl;
                    acc2(a_ch, a_cl, o_ch); a_cl += o_cl;
                }
                if (sh == 2) {
                    const double o_sh = quad_swap<0x4E>(a_sh), o_sl = quad_swap<0x4E>(a_sl);      // lane ^ 2
                    const double o_ch = quad_swap<0x4E>(a_ch), o_cl = quad_swap<0x4E>(a_cl);
                    acc2(a_sh, a_sl, o_sh); a_sl += o_sl;
                    acc2(a_ch, a_cl, o_ch); a_cl += o_cl;
                }
                asm volatile("" :: "v"(h0_), "v"(l0_), "v"(h1_), "v"(l1_) : "memory");
                bool first = false;
#if defined(Q_NOFLUSH)
                if (fl_ && a_sh == 1.2345e300 && a_cl == 7.7e-300) W.acc[0] = a_sh + a_sl + a_ch + a_cl;
#elif defined(Q_STORE) || defined(Q_STOREU)    // (Q_STOREU: the usual scattered places, stores only)
                if (fl_) { a[0] = a_sh; a[1] = a_sl; a[2] = a_ch; a[3] = a_cl; }
#else
#ifndef Q_ST1
                // The entry's two 16-byte halves leave from the end's first TWO slice lanes in ONE store instruction (round 4; -DQ_ST1:
                // both from the first lane): the CU's memory path charges an instruction by the lines it touches
                // (profiles/ta_rate.hip), and the two stores of the one-lane form touch the same ~15 lines twice: 490-492 -> 478-481 ms.
                // (The two LOADS split the same way: no change -- the second load of the one-lane form hits L1.  The finalisation
                //  with an entry per lane pair: +4 ms -- half as many line requests, but the division twice per entry.)
                if (fl_) {
                    first = (h1_ == 0.0);
                    acc2(h0_, l0_, a_sh); l0_ += a_sl;
                    acc2(h1_, l1_, a_ch); l1_ += a_cl;
                }
                if (sh >= 1) {
                    const double p_h = quad_swap<0xB1>(h1_), p_l = quad_swap<0xB1>(l1_);      // (slice 1 <- slice 0)
                    double *dst = a + (slice == 0 ? 0 : 2);
                    const double v0 = slice == 0 ? h0_ : p_h, v1 = slice == 0 ? l0_ : p_l;
                    if (ok && slice <= 1) { dst[0] = v0; dst[1] = v1; }
                } else if (fl_) {
                    a[0] = h0_; a[1] = l0_; a[2] = h1_; a[3] = l1_;
                }
#else
                if (fl_) {
                    first = (h1_ == 0.0);
                    acc2(h0_, l0_, a_sh); l0_ += a_sl;
                    acc2(h1_, l1_, a_ch); l1_ += a_cl;
                    a[0] = h0_; a[1] = l0_; a[2] = h1_; a[3] = l1_;
                }
#endif
#endif
                const unsigned long long fm = __ballot(first);
                if (first) W.touched[W.nt + __popcll(fm & lanemask_lt())] = eu;
                W.nt += __popcll(fm);
                W.paths += (unsigned long long)fill * (unsigned long long)nact;
                W.updates += (unsigned long long)nact;
                if (!pm) break;
                assign();
#ifdef Q_RPIECE
                m0 = recp[my_rec];
#else
                m0 = B.midX[my_rec];
#endif
            }
        }
        if (mine) {    // advance the heads that took part
            dpos++;
            cur.x = INF;
            if (dpos < dend) { cur = B.dir[dpos]; if (cur.x >= xhi) cur.x = INF; }
        }
    }
}

#ifdef Q_PIPE
// ---- the column loop of k_paths4, software-pipelined (round 4; -DQ_PIPE: measured, NOT the default) ----------------------
// The hypothesis it was built on (profiles/r04a_paths_row_ablations.txt: 329 ms without any row access, + 49 ms for the row
// loads, + 73 ms for the scattered row stores, + 75 ms for the finalisation -- the parts ADD UP; removing 45 % of the row
// loads outright bought nothing): the serial chain of a wave.  A column of heads_Q is three DEPENDENT memory trips --
// directory entry of the advanced heads -> {end records, merged records} -> row entries -- plus the acknowledgement of
// its row stores, which the next column's first wait also waits for (vmcnt retires loads and stores in issue order).
// RESULT (profiles/r04b_paths_pipelined.txt): the trips ARE hidden -- a wave of this loop waits 2.0 % of its time for the
// end records, 1.4 % for the directory entries, 0.2 % for the row entries (-DP_TRACE, profiles/tools/trace_p.py) -- and the
// kernel takes as long as before: 539-544 ms at four waves per SIMD against 526 ms for heads_Q at five on the same box.
// What the stamps show instead is waves queueing to ISSUE their vector-memory instructions (14.8 % of a wave's time for
// the eight loads of a fetch, 9.8 % for a flush): the limit is work, not latency -- per column and CU about 375 cycles
// of vector ALU issue and about 450 cycles of the CU's memory path (profiles/ta_rate.hip, ta_rate2.hip: ~5 cycles per
// distinct 128-byte line of an instruction, whatever its width: the 15 row lines a column loads and stores again are
// 40 % of it), out of the 740 the kernel takes; the two pipes overlap poorly with 16-20 waves per CU.  Hence neither
// latency hiding nor occupancy moves the kernel, only fewer instructions / lines per column do.  Kept as the evidence.
// How the trips of consecutive columns overlap here:
//   * the advance of the heads that take part in column i+1 is requested when column i+1 is fetched, and consumed one
//     column later (nxt): a whole column of time;
//   * column i+1 -- merge step, lane assignment of its first set of records, its end records and those records -- is
//     fetched before the record loop of column i runs, and consumed at the top of the next iteration;
//   * the row entries of column i are requested at the top of its iteration as before; its stores are issued behind
//     the loads of column i+1, so no wait on the chain includes their acknowledgement.
// Every load is unconditional (clamped index; lanes without an advance read directory entry 0) and is consumed at one
// unconditional place.  A column whose records do not fit one set of 64, or with more than 64 ends (top_k > 63), is
// worked through synchronously as in heads_Q and the next column is fetched after it (< 1 % of the columns).
// The first column is peeled (body() is inlined twice): at the loop header the compiler merges the counter state of the
// entry with that of the back edge, and an entry without the stores of a previous column would turn the counted
// waits of the steady state into waits for those stores.
template <bool FASTDIV>
__device__ __forceinline__ void heads_P(const Path2Args &B, QAcc &W, int start, long long h0, long long nH, int self, int xlo, int xhi,
                                        bool fresh) {
    __shared__ QLds stagep[4];          // one per wave of the block; DS operations of a wave execute in order
    const PathArgs &A = B.P;
    QLds &L = stagep[threadIdx.x >> 6];
    const int lane = lane_id();
    const int k = A.k;
    const int INF = 0x7fffffff;
    // this lane's head
    const long long h = h0 + lane;
    const bool hv = h < nH;
    double sm1 = 0.0, mu1 = 0.0, f1 = 1.0;
    bool has_e1 = false;
    long long dpos = 0, dend = 0;
    if (hv) {
        int xp = start;
        if (h >= self) {
            const long long rp = A.rnn_ptr[start] + (h - self);
            xp = A.rnn_idx[rp];
            const double sv = A.rnn_val[rp * 3], mu = A.rnn_val[rp * 3 + 1];
            sm1 = sv * mu; mu1 = mu; f1 = A.rnn_val[rp * 3 + 2];
            has_e1 = true;
        }
        const int xpid = B.nb_id[xp];
        dpos = B.dir_ptr[xpid];
        dend = B.dir_ptr[xpid + 1];
        if (xlo > 0) {   // lower bound of xlo in this head's directory (sorted by x)
            long long lo = dpos, hi = dend;
            while (lo < hi) {
                long long mid = (lo + hi) >> 1;
                if (B.dir[mid].x < xlo) lo = mid + 1; else hi = mid;
            }
            dpos = lo;
        }
    }
    MidDir cur;
    cur.x = INF; cur.ne = 0; cur.cnt = 0; cur.pad = 0; cur.off = 0;
    if (hv && dpos < dend) { cur = B.dir[dpos]; if (cur.x >= xhi) cur.x = INF; }
    const int nloc = (nH - h0) < 64 ? (int)(nH - h0) : 64;      // heads of this batch (lanes 0 .. nloc-1)

    // smallest column among the heads (wave-uniform result)
    auto col_min = [&](int xmin) -> int {
#define XM_DPP_MIN(CTRL) { const int o = __builtin_amdgcn_update_dpp(INF, xmin, CTRL, 0xf, 0xf, false); xmin = o < xmin ? o : xmin; }
#define XM_SWZ_MIN(PAT) { const int o = __builtin_amdgcn_ds_swizzle(xmin, PAT); xmin = o < xmin ? o : xmin; }
        if (nloc <= 16) {      // the common batch of a few heads: DPP inside the first row of lanes, no LDS round trips
            XM_DPP_MIN(0xB1) XM_DPP_MIN(0x4E)                               // quad_perm [1,0,3,2], [2,3,0,1]
            if (nloc > 4) { XM_DPP_MIN(0x141) XM_DPP_MIN(0x140) }           // row_half_mirror, row_mirror
            xmin = rl32(xmin, 0);
        } else {
            XM_SWZ_MIN(0x041F) XM_SWZ_MIN(0x081F) XM_SWZ_MIN(0x101F) XM_SWZ_MIN(0x201F) XM_SWZ_MIN(0x401F)
            const int x0 = rl32(xmin, 0), x1 = rl32(xmin, 32);
            xmin = x0 < x1 ? x0 : x1;
        }
#undef XM_SWZ_MIN
#undef XM_DPP_MIN
        return xmin;
    };

    // ---- lane assignment of record sets.  k_cnt / k_off: the tiles of the participating heads of the column whose sets
    // are being dealt (a snapshot of cur taken when the column was fetched: cur itself moves on)
    int k_cnt = 0;
    long long k_off = 0;
    unsigned long long pm = 0;       // participating heads with records not yet dealt
    int pos = 0;                     // records of the first of them already dealt
    int set_n = 0, my_h = 0;
    long long my_rec = 0;
    // a set = up to 64 records, one per lane, of whichever participating head the lane falls to
    auto assign = [&]() {
        set_n = 0; my_h = 0;
        my_rec = rl64(k_off, __ffsll((long long)pm) - 1);     // lanes beyond the set: any record
        for (;;) {
            // (pm, pos and set_n are wave-uniform; said explicitly, or the loop is compiled as a divergent one)
            pm = ((unsigned long long)(unsigned)uniform((int)(pm >> 32)) << 32) | (unsigned)uniform((int)pm);
            pos = uniform(pos); set_n = uniform(set_n);
            if (pm == 0 || set_n >= 64) break;
            const int l = __ffsll((long long)pm) - 1;
            const int cnt = rl32(k_cnt, l);
            const long long off = rl64(k_off, l);
            int n = cnt - pos;
            if (n > 64 - set_n) n = 64 - set_n;
            if (lane >= set_n && lane < set_n + n) { my_rec = off + pos + (lane - set_n); my_h = l; }
            set_n += n; pos += n;
            if (pos == cnt) { pm &= pm - 1; pos = 0; }
        }
    };
    // prepared form of a record: first edge of its head (from the head's lane) + the three middle edges
    auto prepare = [&](const MidX &m, int fill) {
        const bool he1 = __shfl((int)has_e1, my_h, 64) != 0;
        const double hsm1 = __shfl(sm1, my_h, 64), hmu1 = __shfl(mu1, my_h, 64), hf1 = __shfl(f1, my_h, 64);
        asm volatile("" :: "v"(m.sm2), "v"(m.sm3), "v"(m.sm4), "v"(m.f2), "v"(m.f3), "v"(m.f4), "v"(m.mu));
        double bsm, bc;
        if (he1) { bsm = ((hsm1 + m.sm2) + m.sm3) + m.sm4; bc = ((hf1 * m.f2) * m.f3) * m.f4; }
        else { bsm = (m.sm2 + m.sm3) + m.sm4; bc = (m.f2 * m.f3) * m.f4; }
        const double bmu = m.mu + (he1 ? hmu1 : 0.0);
        if (lane < set_n) { L.bsm[fill + lane] = bsm; L.bc[fill + lane] = bc; L.bmu[fill + lane] = bmu; }
    };

    // ---- the column that has been fetched (and, once body() has taken its scalars, the one in work)
    bool have = false;
    unsigned long long part = 0;
    int ne = 0, col = 0;
    ColEnd e;                        // end record of this lane (first 64 ends)
    MidX m0;                         // this lane's record of the first set
    e.sm = 0.0; e.mu = 0.0; e.f = 1.0; e.u = -1; e.pad = 0;
    m0.sm2 = m0.sm3 = m0.sm4 = m0.f2 = m0.f3 = m0.f4 = m0.mu = 0.0; m0.xid = 0; m0.pad = 0;
    MidDir nxt = cur;                // the advanced heads' next directory entry, requested with the column, used a column later
    bool adv = false, took = false;
    // Fetch the next column: complete the advance requested a column ago, merge step, first set of records, loads.
    // The loads are issued whether or not a column is left (valid dummy addresses): one shape of the memory-operation
    // queue on every path is what keeps the compiler's counted waits counted.
    auto fetch = [&]() {
        PT(7)
        if (adv) { cur = nxt; if (cur.x >= xhi) cur.x = INF; }      // (selects: the one place nxt is consumed)
        else if (took) cur.x = INF;
#ifdef P_TRACE
        asm volatile("" : "+v"(cur.x), "+v"(cur.cnt));
#endif
        PT(8)
        const int xmin = col_min(cur.x);
        have = xmin != INF;
        const bool mine = have && cur.x == xmin;
        part = __ballot(mine);
        const int fl = have ? __ffsll((long long)part) - 1 : 0;
        ne = rl32(cur.ne, fl);
        col = rl32(cur.pad, fl);
        if (!have) { ne = 1; col = 0; }
        k_cnt = cur.cnt; k_off = have ? cur.off : 0;
        pm = part; pos = 0;
        set_n = 0; my_h = 0; my_rec = 0;
        if (have) assign();
        PT(9)
        m0 = B.midX[my_rec];
        e = (B.cend + (size_t)col * (k + 1))[(lane < ne) ? lane : ne - 1];      // one 32-byte record per end (k_col_ends)
        took = mine;
        adv = mine && (dpos + 1 < dend);
        if (mine) dpos++;
        nxt = B.dir[adv ? dpos : 0];
        PT(10)
    };

    auto body = [&]() {
        // the column in work takes its scalars out of the fetch state
        const int c_ne = uniform(ne), c_col = uniform(col);
        const unsigned long long c_part = ((unsigned long long)(unsigned)uniform((int)(part >> 32)) << 32) | (unsigned)uniform((int)part);
        const ColEnd *ce = B.cend + (size_t)c_col * (k + 1);
        const int b_last = ((c_ne - 1) >> 6) << 6;       // first end of the last chunk of 64 ends (0 unless top_k > 63)
        // ---- the chunk of ends in work: lanes = ends x record slices
        int nact = 0, sh = 0, slice = 0, eu = 0;
        bool ok = false, fl_ = false;
        double sm5 = 0.0, mu5 = 0.0, f5 = 1.0;
        double *a = W.acc;
        const double *la = W.acc;
        auto ends_ctx = [&](int b) {
            nact = (c_ne - b) < 64 ? (c_ne - b) : 64;
            sh = nact <= 16 ? 2 : (nact <= 32 ? 1 : 0);     // log2 of the record slices per end
            const int q = lane >> sh;
            slice = lane & ((1 << sh) - 1);
            // ends: natural layout -> LDS -> (end, slice) layout
            {
                int eu_ = e.u;
                asm volatile("" : "+v"(eu_));          // (keeps the select below, and with it the wait for e, down here)
                L.e_u[lane] = (b + lane < c_ne) ? eu_ : -1; L.e_sm[lane] = e.sm; L.e_mu[lane] = e.mu; L.e_f[lane] = e.f;
            }
            asm volatile("" ::: "memory");
            const int eur = L.e_u[q];
            ok = eur >= 0;
            eu = eur & (END_HOME - 1);
            sm5 = L.e_sm[q]; mu5 = L.e_mu[q]; f5 = L.e_f[q];
            fl_ = ok && slice == 0;
            a = W.acc + (size_t)(ok ? eu : 0) * 4;
#ifdef Q_NOHOME
            la = a;
#else
            la = (ok && fresh && (eur & END_HOME)) ? g_zero_entry : a;
#endif
        };
        // the row entries of the chunk's ends and the prepared records of one round (up to Q_CAP: a column with more,
        // < 1 % of them, updates its row once per round)
        double h0_ = 0.0, l0_ = 0.0, h1_ = 0.0, l1_ = 0.0;
        int fill = 0;
        auto round_in = [&]() {
            h0_ = la[0]; l0_ = la[1]; h1_ = la[2]; l1_ = la[3];     // requested before the records are prepared and reduced
            la = a;                                                 // (a second round of the same ends finds the first one's sums)
            prepare(m0, 0);
            fill = set_n;
            while (pm && fill < Q_CAP) {
                assign();
                const MidX m = B.midX[my_rec];
                prepare(m, fill);
                fill += set_n;
            }
            asm volatile("" ::: "memory");
        };
        // record loop, slices added up, ONE update of the row
        auto round_out = [&]() {
            double a_sh = 0.0, a_sl = 0.0, a_ch = 0.0, a_cl = 0.0;
            const int steps = (fill + (1 << sh) - 1) >> sh;
            for (int it = 0; it < steps; it++) {
                const int r = (it << sh) + slice;
                if (ok && r < fill) {
                    const double sm = L.bsm[r] + sm5;
                    const double c = L.bc[r] * f5;
                    const double mu = L.bmu[r] + mu5;
                    double sp;
                    if (FASTDIV) sp = div_mid(sm, mu);
                    else sp = (mu != 0.0) ? 1.0 * sm / mu : 0.0;      // calculate_path_confidence (extender.py:83-89)
                    acc2(a_sh, a_sl, sp * c);
                    acc2(a_ch, a_cl, c);
                }
            }
            // the slices of an end sit in adjacent lanes
            if (sh >= 1) {
                const double o_sh = quad_swap<0xB1>(a_sh), o_sl = quad_swap<0xB1>(a_sl);      // lane ^ 1
                const double o_ch = quad_swap<0xB1>(a_ch), o_cl = quad_swap<0xB1>(a_cl);
                acc2(a_sh, a_sl, o_sh); a_sl += o_sl;
                acc2(a_ch, a_cl, o_ch); a_cl += o_cl;
            }
            if (sh == 2) {
                const double o_sh = quad_swap<0x4E>(a_sh), o_sl = quad_swap<0x4E>(a_sl);      // lane ^ 2
                const double o_ch = quad_swap<0x4E>(a_ch), o_cl = quad_swap<0x4E>(a_cl);
                acc2(a_sh, a_sl, o_sh); a_sl += o_sl;
                acc2(a_ch, a_cl, o_ch); a_cl += o_cl;
            }
            PT(4)
            asm volatile("" :: "v"(h0_), "v"(l0_), "v"(h1_), "v"(l1_) : "memory");
            PT(5)
            // (every lane adds; the lanes that own an end store: the masked region is the stores alone, so that no
            //  skip branch -- a second path with a different number of queued stores -- appears around them)
            const bool first = fl_ && (h1_ == 0.0);
            acc2(h0_, l0_, a_sh); l0_ += a_sl;
            acc2(h1_, l1_, a_ch); l1_ += a_cl;
            const unsigned long long fm = __ballot(first);
            // The stores are UNCONDITIONAL: lanes without an update write the wave's junk entry (one address: one
            // request).  A masked store would sit behind a skip branch (the compiler keeps `s_cbranch_execz` around
            // vector-memory instructions), i.e. two paths with different numbers of queued stores, and the counted
            // waits of the next column would fall back to the smaller count -- to waiting for these stores.
            double *dst = fl_ ? a : W.junk;
            int *tp = first ? W.touched + (W.nt + __popcll(fm & lanemask_lt())) : (int *)W.junk + 8;
            dst[0] = h0_; dst[1] = l0_; dst[2] = h1_; dst[3] = l1_;
            *tp = eu;
            W.nt += __popcll(fm);
            W.paths += (unsigned long long)fill * (unsigned long long)nact;
            W.updates += (unsigned long long)nact;
        };
        PT(0)
        ends_ctx(0);
        PT(1)
        round_in();
        PT(2)
        if (!(pm == 0 && b_last == 0)) {
            // every (chunk of ends, round of records) of the column but its last, one after the other (top_k > 63, or more
            // than Q_CAP records: rare).  Kept off the common path: a loop around round_in() would merge this side's
            // counter state into the common one at its header.
            int b = 0;
            do {
                round_out();
                if (pm == 0) {
                    b += 64;
                    e = ce[(b + lane < c_ne) ? b + lane : c_ne - 1];
                    pm = c_part; pos = 0;
                    assign();
                    m0 = B.midX[my_rec];
                    ends_ctx(b);
                } else {
                    assign();
                    m0 = B.midX[my_rec];
                }
                // (round_in() in another order and behind opaque statements: the same code here would be merged with the
                //  common one, and this side's counter state with it)
                asm volatile("s_nop 0" ::: "memory");
                prepare(m0, 0);
                fill = set_n;
                while (pm && fill < Q_CAP) {
                    assign();
                    const MidX m = B.midX[my_rec];
                    prepare(m, fill);
                    fill += set_n;
                }
                asm volatile("s_nop 0" ::: "memory");
                h0_ = la[0]; l0_ = la[1]; h1_ = la[2]; l1_ = la[3];
                la = a;
                asm volatile("s_nop 0" ::: "memory");
            } while (!(pm == 0 && b == b_last));
        }
        // the column's final round: the next column is fetched before its record loop -- the trips of that column run
        // under the loop and the row update of this one
        fetch();
        PT(3)
        round_out();
        PT(6)
    };

    fetch();
    if (have) {
        body();
        while (have) body();
    }
}
#endif  // Q_PIPE

// Waves per SIMD of k_paths4: five with the serial column loop (96 VGPRs).  The pipelined loop (-DQ_PIPE) keeps the next
// column's end record, merged record and directory entry in registers while the current column is reduced: 126 VGPRs,
// four waves.
#ifndef P_WAVES
#ifdef Q_PIPE
#define P_WAVES 4
#else
#define P_WAVES 5
#endif
#endif
template <bool FASTDIV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P_WAVES, P_WAVES))) void k_paths4(Path2Args B) {
    __shared__ FinBuf fin[4];
    const PathArgs &A = B.P;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= A.n_slots) return;
    const int lane = lane_id();
    QAcc W;
    W.paths = 0; W.updates = 0; W.urank = A.urank;
    W.junk = g_junk[slot & 8191];
#ifdef P_TRACE
    for (int i = 0; i < 12; i++) W.pt[i] = 0;
    W.pt_last = clock64();
    const unsigned long long pt_begin = W.pt_last;
#endif
    unsigned long long cand_total = 0;
    for (;;) {
        int u_ = 0;
        if (lane == 0) u_ = (int)atomicAdd(&A.counters[2], 1ull);
        const int unit = uniform(u_);
        if (unit >= A.n_units) break;  // every wave reaches this exit: the cursor only grows
        const int start = uniform(A.unit_start[unit]);
        const int c = uniform(A.unit_c[unit]);
        const int G = uniform(A.unit_G[unit]);
        const int row = uniform(A.unit_row[unit]);
        W.slot_rows = row < 0;
        if (row < 0) {
            W.acc = A.acc + (size_t)slot * (size_t)A.row_stride * 4;
            W.touched = A.touched + (size_t)slot * A.U;
        } else {
            W.acc = A.hacc + (size_t)row * A.U * 4;
            W.touched = A.htouched + (size_t)row * A.U;
        }
        W.nt = 0;
        // work entries of a start, numbered: role T; per head its (t,s) part; per (64-head batch, column range) the tiles.
        // The tiles are walked FIRST (the numbering is what deals the entries to the G units of a heavy start, not the
        // order): while the first head batch runs nothing else has touched the unit's row, which is what lets the home
        // ends of a column be stored without a load (heads_Q); sums are exact, so the order does not show in the result.
        const bool role_t = (A.flags[start] & 2) != 0;
        const long long r0 = uniform((int)A.rnn_ptr[start]), r1 = uniform((int)A.rnn_ptr[start + 1]);
        const int self = (A.cls[start] == 2) ? 1 : 0;   // head 0 = the start itself (target_path, extender.py:160-163)
        const long long nH = self + (r1 - r0);          // heads >= self: start in NN(x') (longest_path, :164-167)
        const long long nbatch = (nH + 63) / 64;
        const int RX = (nbatch > 0) ? (int)((G + nbatch - 1) / nbatch) : 1;   // column ranges: nbatch * RX >= G entries
        const int n_nb = B.n_nb;
        long long ent = (role_t ? 1 : 0) + nH;
        for (long long bt = 0; bt < nbatch; bt++)
            for (int rx = 0; rx < RX; rx++) {
                if (G == 1 || ent % G == c) {
                    const int xlo = (rx == 0) ? 0 : B.nb_list[(long long)rx * n_nb / RX];
                    const int xhi = (rx == RX - 1) ? 0x7fffffff : B.nb_list[(long long)(rx + 1) * n_nb / RX];
#ifdef Q_PIPE        // (the software-pipelined column loop: measured, not faster -- see the comment at heads_P)
                    heads_P<FASTDIV>(B, W, start, bt * 64, nH, self, xlo, xhi, bt == 0);
#else
                    heads_Q<FASTDIV>(B, W, start, bt * 64, nH, self, xlo, xhi, bt == 0);
#endif
                }
                ent++;
            }
        ent = 0;
        if (role_t) {   // role T: non-joint paths from t = start (final_nonjoint_extend, extender.py:124-140,:180)
            if (G == 1 || ent % G == c) {
                Carry none; none.sm = 0; none.mu = 0; none.c = 0;
                through_t(A, W, start, false, none);
            }
            ent++;
        }
        for (long long h = 0; h < nH; h++) {
            if (G == 1 || ent % G == c) {
                const bool has_e1 = h >= self;
                const int xp = has_e1 ? A.rnn_idx[r0 + h - self] : start;
                Carry e1; e1.sm = 0; e1.mu = 0; e1.c = 1.0;
                if (has_e1) e1 = first_edge(A.rnn_val[(r0 + h - self) * 3], A.rnn_val[(r0 + h - self) * 3 + 1],
                                            A.rnn_val[(r0 + h - self) * 3 + 2]);
                head_S(A, W, xp, has_e1, e1);
            }
            ent++;
        }
#ifdef EXP_NOFIN     // (ablation, profiles/tools/a_variants.sh with FILE=stage_b: no finalisation -- wrong results, timing only)
        if (row < 0) cand_total += W.nt;
#else
        if (row < 0) cand_total += finalize_start(A, fin[threadIdx.x >> 6], W.acc, W.touched, W.nt, start);
#endif
        else if (lane == 0) A.unit_nt[unit] = W.nt;
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], cand_total);
        atomicAdd(&A.counters[1], W.paths);
        atomicAdd(&A.counters[4], W.updates);
#ifdef P_TRACE
        for (int i = 0; i < 12; i++) atomicAdd(&g_ptrace[i], W.pt[i]);
        atomicAdd(&g_ptrace[12], clock64() - pt_begin);
        atomicAdd(&g_ptrace[13], 1ull);
#endif
    }
}

// precondition of the bare division sequence (div_mid): every kept pair has a positive, finite mutuality within 2^+-100
// and a product sim * mutu that is zero or within 2^+-400 -- then a path's mutuality sum is never zero and no operand is
// near the ends of the exponent range.  What stage A produces always qualifies; records fed by a caller are checked.
__global__ __launch_bounds__(256) void k_edge_ranges(long long n, const double *sim, const int *mutu, int *bad) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double m = (double)mutu[p], sm = fabs(sim[p] * m);
    const bool ok = (m >= 1.0) && (sm == 0.0 || (sm > 0x1p-400 && sm < 0x1p400));      // mutu is an int32 count: >= 1 is "positive"
    if (!ok) atomicOr(bad, 1);
}

// the ends of every column x (non-bridge record): x itself, then NN(x) in list order, as 32-byte records
// home column of every end = the lowest column x whose end list {x} + NN(x) holds it (home[] preset to INT_MAX)
__global__ __launch_bounds__(256) void k_col_home(int n_nb, int k, const int *nb_list, const int *kcnt, const int *kcol, int *home) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_nb * (k + 1)) return;
    const int xid = (int)(t / (k + 1)), idx = (int)(t % (k + 1));
    const int x = nb_list[xid];
    int e = -1;
    if (idx == 0) e = x;
    else if (idx - 1 < kcnt[(size_t)x * 2 + 1]) e = kcol[((size_t)x * 2 + 1) * k + (idx - 1)];
    if (e >= 0) atomicMin(&home[e], x);
}

__global__ __launch_bounds__(256) void k_col_ends(int n_nb, int k, const int *nb_list, const int *kcnt, const int *kcol, const double *kval,
                                                  const int *urank, const int *home, ColEnd *cend) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_nb * (k + 1)) return;
    const int xid = (int)(t / (k + 1)), idx = (int)(t % (k + 1));
    const int x = nb_list[xid];
    ColEnd e;
    e.sm = 0.0; e.mu = 0.0; e.f = 1.0; e.u = -1; e.pad = 0;
    int item = -1;
    if (idx == 0) item = x;
    else if (idx - 1 < kcnt[(size_t)x * 2 + 1]) {
        const size_t o = ((size_t)x * 2 + 1) * k + (idx - 1);
        const double v = kval[o * 3], m = kval[o * 3 + 1];
        e.sm = v * m; e.mu = m; e.f = kval[o * 3 + 2];
        item = kcol[o];
    }
    if (item >= 0) e.u = urank[item] | (home[item] == x ? END_HOME : 0);      // (the ends of a column are distinct items)
    cend[t] = e;
}

// items that can end a path: the s of every src record, the x of every attach record, x and NN(x) of every non-bridge record
__global__ __launch_bounds__(256) void k_mark_ends(int I, int k, const uint8_t *cls, const int *kcnt, const int *kcol, long long n_src,
                                                   const int *src_idx, long long n_att, const int *att_idx, int *mark) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_src) mark[src_idx[t]] = 1;
    if (t < n_att) mark[att_idx[t]] = 1;
    if (t < (long long)I * k) {
        const int x = (int)(t / k), q = (int)(t % k);
        if (cls[x] == 2) {
            if (q == 0) mark[x] = 1;
            if (q < kcnt[(size_t)x * 2 + 1]) mark[kcol[((size_t)x * 2 + 1) * k + q]] = 1;
        }
    }
}
__global__ __launch_bounds__(256) void k_end_ranks(int I, const int *mark, const long long *rank64, int *urank, int *uitem) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    const int r = (int)rank64[i];
    urank[i] = mark[i] ? r : -1;
    if (mark[i]) uitem[r] = i;
}

// heavy starts: add the G partial rows into the first one (double-double merge), then finalise.  One block of
// MERGE_WAVES waves per job: the touched entries of a partial row are distinct, so the waves take 64 of them at a
// time side by side (a single wave per start had left a chain of G - 1 serial merges: 68 ms at BASELINE configs[1]);
// the finalisation pass is shared the same way, every wave keeping the best of its share, wave 0 the best of those.
// Two levels for the starts with more than MERGE_GROUP rows (the heaviest has 133): level 1 folds every group of
// MERGE_GROUP consecutive rows into the group's first row (one block per group), level 2 the group heads into row 0.
constexpr int MERGE_WAVES = 16;
constexpr int MERGE_GROUP = 12;

// rows of one start: add row (acc_s, touched_s[0..nt_s)) into (acc_d, touched_d, *s_nt); all waves of the block
// (four sets of 64 entries per wave and step with all their loads in flight -- 120 VGPRs, one block per CU -- made the two
//  kernels slower: 5.35 + 6.07 ms against 4.23 + 5.45, round 4)
__device__ __forceinline__ void merge_row(double *acc_d, int *touched_d, int *s_nt, double *acc_s, const int *touched_s, int nt_s) {
#ifndef Q_MRG1    // An entry per lane PAIR (round 4): the even lane adds the (value, error) pair of the sums, the odd lane that of the
                  // weights -- the two halves are independent, and every load / store instruction touches each line once:
                  // k_merge_groups 4.07 -> 3.55 ms, k_merge 5.62 -> 5.03 ms (rocprof, profiles/tools/merge_ab.sh).  -DQ_MRG1: an entry per lane
    {
        const int lane = lane_id(), w = threadIdx.x >> 6;
        const int half = (lane & 1) * 2;
        for (int b0 = 32 * w; b0 < nt_s; b0 += 32 * MERGE_WAVES) {
            const int b = b0 + (lane >> 1);
            bool first = false;
            int e = 0;
            if (b < nt_s) {
                e = touched_s[b];
                double *s = acc_s + (size_t)e * 4 + half, *d = acc_d + (size_t)e * 4 + half;
                double hi = d[0], lo = d[1];
                first = half == 2 && hi == 0.0;
                dd_add(hi, lo, s[0]); dd_add(hi, lo, s[1]);
                d[0] = hi; d[1] = lo;
                s[0] = 0.0; s[1] = 0.0;
            }
            const unsigned long long m = __ballot(first);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(s_nt, __popcll(m));
            base = rl32(base, 0);
            if (first) touched_d[base + __popcll(m & lanemask_lt())] = e;
        }
        __syncthreads();
        return;
    }
#endif
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int b0 = 64 * w; b0 < nt_s; b0 += 64 * MERGE_WAVES) {
        const int b = b0 + lane;
        bool first = false;
        int e = 0;
        if (b < nt_s) {
            e = touched_s[b];
            double *s = acc_s + (size_t)e * 4, *d = acc_d + (size_t)e * 4;
            double s_hi = d[0], s_lo = d[1], c_hi = d[2], c_lo = d[3];
            first = (c_hi == 0.0);
            dd_add(s_hi, s_lo, s[0]); dd_add(s_hi, s_lo, s[1]);
            dd_add(c_hi, c_lo, s[2]); dd_add(c_hi, c_lo, s[3]);
            d[0] = s_hi; d[1] = s_lo; d[2] = c_hi; d[3] = c_lo;
            s[0] = 0.0; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
        }
        const unsigned long long m = __ballot(first);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(s_nt, __popcll(m));
        base = rl32(base, 0);
        if (first) touched_d[base + __popcll(m & lanemask_lt())] = e;
    }
    __syncthreads();      // the destination row and its touched list are complete before the next row (entries repeat)
}

__global__ __launch_bounds__(64 * MERGE_WAVES) void k_merge_groups(PathArgs A, int n_heavy, const int *heavy_unit0) {
    __shared__ int s_nt;
    const int h = blockIdx.x;
    if (h >= n_heavy) return;
    const int u0 = heavy_unit0[h];
    const int G = A.unit_G[u0], r0 = A.unit_row[u0];
    if (G <= MERGE_GROUP) return;
    for (int g = blockIdx.y; g * MERGE_GROUP < G; g += gridDim.y) {
        const int b = g * MERGE_GROUP;
        const int e = (b + MERGE_GROUP) < G ? (b + MERGE_GROUP) : G;
        if (threadIdx.x == 0) s_nt = A.unit_nt[u0 + b];
        __syncthreads();
        for (int c = b + 1; c < e; c++)
            merge_row(A.hacc + (size_t)(r0 + b) * A.U * 4, A.htouched + (size_t)(r0 + b) * A.U, &s_nt,
                      A.hacc + (size_t)(r0 + c) * A.U * 4, A.htouched + (size_t)(r0 + c) * A.U, A.unit_nt[u0 + c]);
        if (threadIdx.x == 0) A.unit_nt[u0 + b] = s_nt;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64 * MERGE_WAVES) void k_merge(PathArgs A, int n_heavy, const int *heavy_unit0) {
    __shared__ FinBuf fin[MERGE_WAVES];
    __shared__ int s_nt, s_ns[MERGE_WAVES], s_full;
    __shared__ unsigned long long s_off;
    const int h = blockIdx.x;
    if (h >= n_heavy) return;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int u0 = heavy_unit0[h];
    const int start = A.unit_start[u0], G = A.unit_G[u0], r0 = A.unit_row[u0];
    double *acc0 = A.hacc + (size_t)r0 * A.U * 4;
    int *touched0 = A.htouched + (size_t)r0 * A.U;
    if (threadIdx.x == 0) s_nt = A.unit_nt[u0];
    __syncthreads();
    const int stride = G > MERGE_GROUP ? MERGE_GROUP : 1;     // group heads (k_merge_groups ran) or all rows
    for (int c = stride; c < G; c += stride)
        merge_row(acc0, touched0, &s_nt, A.hacc + (size_t)(r0 + c) * A.U * 4, A.htouched + (size_t)(r0 + c) * A.U,
                  A.unit_nt[u0 + c]);
    const int nt = s_nt;
    if (w == 0) {
        if (lane == 0) A.n_cand[start] = nt;
        unsigned long long off;
        const bool full = fin_list_offset(A, nt, start, off);
        if (lane == 0) { s_off = off; s_full = full ? 1 : 0; }
    }
    __syncthreads();
    const int ns = finalize_slice(A, fin[w], acc0, touched0, nt, s_off, s_full != 0, w, MERGE_WAVES);
    if (lane == 0) s_ns[w] = ns;
    __syncthreads();
    if (w == 0) {       // the best of the waves' best
        volatile double *bv = fin[0].v;
        volatile int *be = fin[0].e;
        int nbuf = 0;
        for (int o = 0; o < MERGE_WAVES; o++) {
            const int n = s_ns[o];
            int te = 0;
            double tv = 0.0;
            if (lane < n) { te = ((volatile int *)fin[o].oe)[lane]; tv = ((volatile double *)fin[o].ov)[lane]; }
            if (lane < n) { be[nbuf + lane] = te; bv[nbuf + lane] = tv; }
            nbuf += n;
        }
        fin_cut(fin[0], nbuf, A.top_end + (size_t)start * XMAP_TOPC, A.top_val + (size_t)start * XMAP_TOPC);
        if (lane == 0) atomicAdd(&A.counters[0], (unsigned long long)nt);
    }
}

// ---- per-start path counts (scheduling weights): T(s) tails of s, sums over src(t), heads of x' -------------
__global__ __launch_bounds__(256) void k_w_tails(int I, const long long *att_ptr, const int *att_idx, const int *kcnt,
                                                 long long *T) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= I) return;
    long long a0 = att_ptr[s], a1 = att_ptr[s + 1], t = 0;
    for (long long ap = a0; ap < a1; ap++) t += 1 + kcnt[(size_t)att_idx[ap] * 2 + 1];
    T[s] = (a1 > a0) ? t + 1 : 0;
}
__global__ __launch_bounds__(256) void k_w_src(int I, const long long *src_ptr, const int *src_idx, const uint8_t *src_flag,
                                               const long long *T, long long *ST_all, long long *ST_j) {
    int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= I) return;
    int lane = lane_id();
    long long a = 0, j = 0;
    for (long long p = src_ptr[t] + lane; p < src_ptr[t + 1]; p += 64) {
        long long v = T[src_idx[p]];
        a += v;
        if (src_flag[p] & 1) j += v;
    }
    a = wave_sum_ll(a);
    j = wave_sum_ll(j);
    if (lane == 0) { ST_all[t] = a; ST_j[t] = j; }
}
__global__ __launch_bounds__(256) void k_w_heads(int I, int k, const uint8_t *cls, const int *kcnt, const int *kcol,
                                                 const uint8_t *flags, const long long *ST_j, long long *HX) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= I) return;
    long long h = 0;
    if (cls[x] == 2) {
        int nb = kcnt[(size_t)x * 2];
        for (int q = 0; q < nb; q++) {
            int t = kcol[((size_t)x * 2) * k + q];
            if (flags[t] & 2) h += ST_j[t];
        }
    }
    HX[x] = h;
}
__global__ __launch_bounds__(256) void k_w_starts(int I, const uint8_t *flags, const long long *rnn_ptr, const int *rnn_idx,
                                                  const long long *ST_all, const long long *HX, long long *P) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= I) return;
    long long p = ((flags[s] & 2) ? ST_all[s] : 0) + HX[s];
    for (long long rp = rnn_ptr[s]; rp < rnn_ptr[s + 1]; rp++) p += HX[rnn_idx[rp]];
    P[s] = p;
}

}  // namespace xmap

using namespace xmap;

extern "C" {
#ifdef Q_HIST
int xmap_debug_qhist(unsigned long long *host, int reset) {
    if (reset) { unsigned long long z[128] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(xmap::g_qhist), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xmap::g_qhist), 128 * sizeof(unsigned long long));
}
#endif
#ifdef P_TRACE
int xmap_debug_ptrace(unsigned long long *host, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(xmap::g_ptrace), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xmap::g_ptrace), 16 * sizeof(unsigned long long));
}
#endif
#ifdef B_TRACE
int xmap_debug_btrace(unsigned long long *host, long long n_units) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(xmap::g_btrace), (size_t)n_units * 16);
}
#endif

int xmap_bridge_flags(void *stream, const xmap_sim *S, const int32_t *prefix_cls, uint8_t *bb) {
    XM_ARG(S && prefix_cls && bb);
    if (S->n_items == 0) return XMAP_OK;
    k_bridge_flags<<<dim3((unsigned)((S->n_items + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        S->n_items, (const long long *)S->row_ptr, S->col, prefix_cls, bb);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_knn_classify(void *stream, const xmap_sim *S, int top_k, const uint8_t *bb, const int32_t *suffix_cls,
                      const uint32_t *contains_mask, uint8_t *cls, int32_t *kcnt, int32_t *kcol, double *kval,
                      int32_t row_lo, int32_t row_hi) {
    XM_ARG(S && bb && suffix_cls && contains_mask && cls && kcnt && kcol && kval);
    XM_ARG(top_k >= 1 && 2 * top_k <= K_CH / 2);
    XM_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= S->n_items);
    if (row_hi == row_lo) return XMAP_OK;
    KnnArgs A;
    A.I = S->n_items; A.k = top_k; A.row_lo = row_lo;
    A.row_ptr = (const long long *)S->row_ptr; A.col = S->col; A.sim = S->sim; A.mutu = S->mutu; A.nij = S->nij;
    A.info = S->info; A.frac = S->frac; A.bb = bb; A.suffix_cls = suffix_cls; A.contains_mask = contains_mask;
    A.cls = cls; A.kcnt = kcnt; A.kcol = kcol; A.kval = kval;
    if (2 * top_k <= K_CH_SMALL / 2) {
        k_knn_classify<K_CH_SMALL, 1><<<dim3((unsigned)(row_hi - row_lo)), dim3(K_THREADS), 0, (hipStream_t)stream>>>(A);
        XM_LAUNCH_CHECK();
        k_knn_classify<K_CH, 2><<<dim3((unsigned)(row_hi - row_lo)), dim3(K_THREADS), 0, (hipStream_t)stream>>>(A);
    } else {      // (lists too long for the small instance's carry-over: every row on the large one)
        k_knn_classify<K_CH, 0><<<dim3((unsigned)(row_hi - row_lo)), dim3(K_THREADS), 0, (hipStream_t)stream>>>(A);
    }
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

static int reverse_common(void *stream, bool fill, const xmap_sim *S, int mode, int top_k, const uint8_t *bb,
                          const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol, const double *kval,
                          const int32_t *suffix_cls, const uint32_t *contains_mask, const uint8_t *flags,
                          const int64_t *attach_ptr, const void *thr, int32_t *long_rows, uint8_t *eflag, int32_t *rcnt,
                          const int64_t *rptr, int32_t *ridx, double *rval, uint8_t *rflag, int32_t row_lo, int32_t row_hi,
                          int32_t *rcnt2 = nullptr) {
    XM_ARG(S && bb && cls && kcnt && kcol && kval && suffix_cls && contains_mask && flags);
    XM_ARG((mode >= 0 && mode <= 2) || (mode == 3 && !fill && rcnt2 && eflag));
    XM_ARG(mode != 1 || attach_ptr);
    XM_ARG(row_lo >= 0 && row_lo <= row_hi && row_hi <= S->n_items);
    if (row_hi == row_lo) return XMAP_OK;
    RevArgs A;
    A.I = S->n_items; A.k = top_k; A.mode = mode; A.thr = (const KnnThr *)thr; A.long_rows = long_rows; A.eflag = eflag;
    A.row_lo = row_lo; A.row_hi = row_hi;
    const char *rl = getenv("XMAP_REV_LONG");
    A.rev_long = (rl && atoi(rl) > 0) ? atoi(rl) : REV_LONG;
    A.row_ptr = (const long long *)S->row_ptr; A.col = S->col; A.sim = S->sim; A.mutu = S->mutu; A.nij = S->nij;
    A.info = S->info; A.frac = S->frac; A.bb = bb; A.cls = cls; A.kcnt = kcnt; A.kcol = kcol; A.kval = kval;
    A.suffix_cls = suffix_cls; A.contains_mask = contains_mask; A.flags = flags;
    A.attach_ptr = (const long long *)attach_ptr;
    A.rcnt2 = rcnt2;
    A.rcnt = rcnt; A.rptr = (const long long *)rptr; A.ridx = ridx; A.rval = rval; A.rflag = rflag;
    const int n_rows = row_hi - row_lo;
    dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;

    if (long_rows && !fill) {     // the count pass lists the long rows, the fill pass that follows reuses the list
        XM_HIP(hipMemsetAsync(long_rows, 0, sizeof(int32_t), st));
        k_rev_long_rows<<<dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st>>>(row_lo, row_hi, (const long long *)S->row_ptr,
                                                                                  A.rev_long, long_rows);
        XM_LAUNCH_CHECK();
    }
    if (fill) k_reverse<true><<<grid, block, 0, st>>>(A);
    else k_reverse<false><<<grid, block, 0, st>>>(A);
    XM_LAUNCH_CHECK();
    if (long_rows) {
        if (fill) k_reverse_long<true><<<dim3(512), dim3(64 * REV_WAVES), 0, st>>>(A);
        else k_reverse_long<false><<<dim3(512), dim3(64 * REV_WAVES), 0, st>>>(A);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

int xmap_knn_thresholds(void *stream, int32_t n_items, int top_k, const int32_t *kcnt, const int32_t *kcol, const double *kval,
                        void *thr) {
    XM_ARG(kcnt && kcol && kval && thr);
    if (n_items == 0) return XMAP_OK;
    k_knn_thresholds<<<dim3((unsigned)((2ll * n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, top_k, kcnt, kcol, kval, (KnnThr *)thr);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_reverse_count(void *stream, const xmap_sim *S, int mode, int top_k, const uint8_t *bb, const uint8_t *cls,
                       const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                       const uint32_t *contains_mask, const uint8_t *flags, const int64_t *attach_ptr, const void *thr,
                       int32_t *long_rows, uint8_t *eflag, int32_t *rcnt, int32_t row_lo, int32_t row_hi) {
    XM_ARG(rcnt);
    return reverse_common(stream, false, S, mode, top_k, bb, cls, kcnt, kcol, kval, suffix_cls, contains_mask, flags,
                          attach_ptr, thr, long_rows, eflag, rcnt, nullptr, nullptr, nullptr, nullptr, row_lo, row_hi);
}

int xmap_reverse_count_att_rnn(void *stream, const xmap_sim *S, int top_k, const uint8_t *bb, const uint8_t *cls,
                               const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                               const uint32_t *contains_mask, const uint8_t *flags, const void *thr, int32_t *long_rows,
                               uint8_t *eflag, int32_t *rcnt_att, int32_t *rcnt_rnn, int32_t row_lo, int32_t row_hi) {
    XM_ARG(rcnt_att && rcnt_rnn && eflag);
    return reverse_common(stream, false, S, 3, top_k, bb, cls, kcnt, kcol, kval, suffix_cls, contains_mask, flags, nullptr, thr,
                          long_rows, eflag, rcnt_att, nullptr, nullptr, nullptr, nullptr, row_lo, row_hi, rcnt_rnn);
}

int xmap_reverse_fill(void *stream, const xmap_sim *S, int mode, int top_k, const uint8_t *bb, const uint8_t *cls,
                      const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                      const uint32_t *contains_mask, const uint8_t *flags, const int64_t *attach_ptr, const void *thr,
                      int32_t *long_rows, uint8_t *eflag, const int64_t *rptr, int32_t *ridx, double *rval, uint8_t *rflag,
                      int32_t row_lo, int32_t row_hi) {
    XM_ARG(rptr && ridx && rval);
    return reverse_common(stream, true, S, mode, top_k, bb, cls, kcnt, kcol, kval, suffix_cls, contains_mask, flags,
                          attach_ptr, thr, long_rows, eflag, nullptr, rptr, ridx, rval, rflag, row_lo, row_hi);
}

int xmap_topc_from_lists(void *stream, int32_t n_items, const int64_t *xs_ptr, const int32_t *xs_end, const double *xs_val,
                         int32_t *n_cand, int32_t *top_end, double *top_val) {
    XM_ARG(xs_ptr && xs_end && xs_val && n_cand && top_end && top_val);
    if (n_items == 0) return XMAP_OK;
    k_topc_lists<<<dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, (const long long *)xs_ptr, xs_end, xs_val, n_cand, top_end, top_val);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_path_weights(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                      const int64_t *src_ptr, const int32_t *src_idx, const uint8_t *src_flag, const int64_t *rnn_ptr,
                      const int32_t *rnn_idx, int64_t *tmp /*[4][I]*/, int64_t *paths /*[I]*/) {
    XM_ARG(cls && kcnt && kcol && flags && att_ptr && src_ptr && rnn_ptr && tmp && paths);
    if (n_items == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    long long *T = (long long *)tmp, *STa = T + n_items, *STj = STa + n_items, *HX = STj + n_items;
    unsigned g1 = (unsigned)((n_items + 255) / 256), g4 = (unsigned)((n_items + 3) / 4);
    k_w_tails<<<dim3(g1), dim3(256), 0, st>>>(n_items, (const long long *)att_ptr, att_idx, kcnt, T);
    XM_LAUNCH_CHECK();
    k_w_src<<<dim3(g4), dim3(256), 0, st>>>(n_items, (const long long *)src_ptr, src_idx, src_flag, T, STa, STj);
    XM_LAUNCH_CHECK();
    k_w_heads<<<dim3(g1), dim3(256), 0, st>>>(n_items, top_k, cls, kcnt, kcol, flags, STj, HX);
    XM_LAUNCH_CHECK();
    k_w_starts<<<dim3(g1), dim3(256), 0, st>>>(n_items, flags, (const long long *)rnn_ptr, rnn_idx, STa, HX,
                                                (long long *)paths);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

static int extend_paths_impl(const Path2Args *mid, void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const double *kval, const uint8_t *flags, const int64_t *att_ptr,
                      const int32_t *att_idx, const double *att_val, const int64_t *src_ptr, const int32_t *src_idx,
                      const double *src_val, const uint8_t *src_flag, const int64_t *rnn_ptr, const int32_t *rnn_idx,
                      const double *rnn_val, int32_t n_units, const int32_t *unit_start, const int32_t *unit_c,
                      const int32_t *unit_G, const int32_t *unit_row, int32_t *unit_nt, int32_t n_heavy,
                      const int32_t *heavy_unit0, int32_t n_slots, double *acc, int32_t *touched, double *hacc,
                      int32_t *htouched, int32_t *n_cand, int32_t *top_end, double *top_val, int64_t xs_cap,
                      int64_t *xs_off, int32_t *xs_end, double *xs_val, int64_t *d_counters, int64_t *h_counters) {
    XM_ARG(cls && kcnt && kcol && kval && flags && att_ptr && src_ptr && rnn_ptr);
    XM_ARG(acc && touched && n_cand && top_end && top_val && d_counters);
    XM_ARG(n_slots > 0 && n_units >= 0 && n_heavy >= 0);
    XM_ARG(n_units == 0 || (unit_start && unit_c && unit_G && unit_row && unit_nt));
    XM_ARG(n_heavy == 0 || (heavy_unit0 && hacc && htouched));
    XM_ARG(xs_cap == 0 || (xs_off && xs_end && xs_val));
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(hipMemsetAsync(d_counters, 0, 4 * sizeof(int64_t), st));
    if (n_units > 0) {
        PathArgs A;
        A.I = n_items; A.k = top_k;
        A.cls = cls; A.kcnt = kcnt; A.kcol = kcol; A.kval = kval; A.flags = flags;
        A.att_ptr = (const long long *)att_ptr; A.att_idx = att_idx; A.att_val = att_val;
        A.src_ptr = (const long long *)src_ptr; A.src_idx = src_idx; A.src_val = src_val; A.src_flag = src_flag;
        A.rnn_ptr = (const long long *)rnn_ptr; A.rnn_idx = rnn_idx; A.rnn_val = rnn_val;
        A.n_units = n_units; A.unit_start = unit_start; A.unit_c = unit_c; A.unit_G = unit_G; A.unit_row = unit_row;
        A.unit_nt = unit_nt;
        A.n_slots = n_slots; A.acc = acc; A.touched = touched; A.hacc = hacc; A.htouched = htouched;
        A.n_cand = n_cand; A.top_end = top_end; A.top_val = top_val;
        A.xs_cap = xs_cap; A.xs_off = (long long *)xs_off; A.xs_end = xs_end; A.xs_val = xs_val;
        A.counters = (unsigned long long *)d_counters;
        A.U = n_items; A.urank = nullptr; A.uitem = nullptr; A.row_stride = n_items;
        int slots = n_slots < n_units ? n_slots : n_units;
        A.n_slots = slots;
#ifdef XMAP_CROSSCHECK
        if (mid) {
            Path2Args B = *mid;
            B.P = A;
            k_paths2<<<dim3((unsigned)((slots + 3) / 4)), dim3(256), 0, st>>>(B);
        } else
#endif
        {
            k_paths<<<dim3((unsigned)((slots + 3) / 4)), dim3(256), 0, st>>>(A);
        }
        XM_LAUNCH_CHECK();
        if (n_heavy > 0) {
            k_merge_groups<<<dim3((unsigned)n_heavy, 16), dim3(64 * MERGE_WAVES), 0, st>>>(A, n_heavy, heavy_unit0);
            XM_LAUNCH_CHECK();
            k_merge<<<dim3((unsigned)n_heavy), dim3(64 * MERGE_WAVES), 0, st>>>(A, n_heavy, heavy_unit0);
            XM_LAUNCH_CHECK();
        }
    }
    if (h_counters) {
        XM_HIP(hipMemcpyAsync(h_counters, d_counters, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (xs_cap > 0 && h_counters[0] > xs_cap) {
            set_error("candidate buffer too small: need %lld entries, have %lld", (long long)h_counters[0],
                      (long long)xs_cap);
            return XMAP_ERR_CAPACITY;
        }
    }
    return XMAP_OK;
}

int xmap_extend_paths(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const double *kval, const uint8_t *flags, const int64_t *att_ptr,
                      const int32_t *att_idx, const double *att_val, const int64_t *src_ptr, const int32_t *src_idx,
                      const double *src_val, const uint8_t *src_flag, const int64_t *rnn_ptr, const int32_t *rnn_idx,
                      const double *rnn_val, int32_t n_units, const int32_t *unit_start, const int32_t *unit_c,
                      const int32_t *unit_G, const int32_t *unit_row, int32_t *unit_nt, int32_t n_heavy,
                      const int32_t *heavy_unit0, int32_t n_slots, double *acc, int32_t *touched, double *hacc,
                      int32_t *htouched, int32_t *n_cand, int32_t *top_end, double *top_val, int64_t xs_cap,
                      int64_t *xs_off, int32_t *xs_end, double *xs_val, int64_t *d_counters, int64_t *h_counters) {
    return extend_paths_impl(nullptr, stream, n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val, src_flag, rnn_ptr, rnn_idx, rnn_val, n_units, unit_start, unit_c, unit_G, unit_row, unit_nt, n_heavy, heavy_unit0, n_slots, acc, touched, hacc, htouched, n_cand, top_end, top_val, xs_cap, xs_off, xs_end, xs_val, d_counters, h_counters);
}

#ifdef XMAP_CROSSCHECK
int xmap_extend_paths2(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const double *kval, const uint8_t *flags, const int64_t *att_ptr,
                      const int32_t *att_idx, const double *att_val, const int64_t *src_ptr, const int32_t *src_idx,
                      const double *src_val, const uint8_t *src_flag, const int64_t *rnn_ptr, const int32_t *rnn_idx,
                      const double *rnn_val, int32_t n_units, const int32_t *unit_start, const int32_t *unit_c,
                      const int32_t *unit_G, const int32_t *unit_row, int32_t *unit_nt, int32_t n_heavy,
                      const int32_t *heavy_unit0, int32_t n_slots, double *acc, int32_t *touched, double *hacc,
                      int32_t *htouched, int32_t *n_cand, int32_t *top_end, double *top_val, int64_t xs_cap,
                      int64_t *xs_off, int32_t *xs_end, double *xs_val, int64_t *d_counters, int64_t *h_counters,
                       const int32_t *nb_id, const int32_t *nb_list, int32_t n_nb, const void *midX, const void *dir,
                       const int64_t *dir_ptr, const int32_t *ng) {
    XM_ARG(nb_id && nb_list && midX && dir && dir_ptr && ng && n_nb > 0);
    Path2Args B;
    memset(&B, 0, sizeof(B));
    B.nb_id = nb_id; B.nb_list = nb_list; B.n_nb = n_nb; B.midX = (const MidX *)midX; B.dir = (const MidDir *)dir;
    B.dir_ptr = (const long long *)dir_ptr; B.ng = ng;
    return extend_paths_impl(&B, stream, n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val, src_flag, rnn_ptr, rnn_idx, rnn_val, n_units, unit_start, unit_c, unit_G, unit_row, unit_nt, n_heavy, heavy_unit0, n_slots, acc, touched, hacc, htouched, n_cand, top_end, top_val, xs_cap, xs_off, xs_end, xs_val, d_counters, h_counters);
}

#endif  // XMAP_CROSSCHECK

static MidArgs mid_args(int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                        const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                        const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                        const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id) {
    MidArgs A;
    memset(&A, 0, sizeof(A));
    A.I = n_items; A.k = top_k; A.cls = cls; A.kcnt = kcnt; A.kcol = kcol; A.kval = kval; A.flags = flags;
    A.att_ptr = (const long long *)att_ptr; A.att_idx = att_idx; A.att_val = att_val;
    A.src_ptr = (const long long *)src_ptr; A.src_idx = src_idx; A.src_val = src_val; A.src_flag = src_flag;
    A.n_nb = n_nb; A.nb_list = nb_list; A.nb_id = nb_id;
    return A;
}

// row-wise construction (k_mid_rows); the tile counters of a column range of the row live in LDS
static_assert((size_t)XMAP_MID_ROWS_SPAN * 4 + (size_t)MIDROW_WAVES * 64 * 4 + 3072 + 256 <= 160 * 1024,
              "k_mid_rows: tile counters + the neighbour table + the waves' walk state must fit the LDS of a gfx950 CU");
// the compacted joint lists of one call (temporaries of the caller's scope)
static int mid_joints(hipStream_t st, MidArgs &A, bool ranges) {
    int *jcnt = nullptr, *joff = nullptr;
    long long *jptr = nullptr;
    const int I = A.I;
    XM_HIP(xm_malloc_async((void **)&jcnt, sizeof(int) * (size_t)(I > 0 ? I : 1), st));
    XM_HIP(xm_malloc_async((void **)&jptr, sizeof(long long) * (size_t)(I + 1), st));
    int64_t nj = 0;
    if (I > 0) {
        k_joint_list<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, A.src_ptr, A.src_flag, jcnt, nullptr, nullptr);
        XM_LAUNCH_CHECK();
    }
    int rc = xmap_exclusive_scan_i32_to_i64(st, jcnt, (int64_t *)jptr, I, &nj);      // (one synchronisation: the size of joff)
    if (rc) return rc;
    XM_HIP(xm_malloc_async((void **)&joff, sizeof(int) * (size_t)(nj > 0 ? nj : 1), st));
    if (I > 0) {
        k_joint_list<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, A.src_ptr, A.src_flag, nullptr, jptr, joff);
        XM_LAUNCH_CHECK();
    }
    A.jptr = jptr; A.joff = joff;
    {   // (attach lists belong to the non-bridge items' first lists: at most k entries each)
        const long long bound = (long long)A.n_nb * A.k;
        int *axid = nullptr;
        XM_HIP(xm_malloc_async((void **)&axid, sizeof(int) * (size_t)(bound > 0 ? bound : 1), st));
        if (bound > 0) {
            k_att_columns<<<dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, st>>>(bound, I, A.att_ptr, A.att_idx, A.nb_id, axid);
            XM_LAUNCH_CHECK();
        }
        A.axid = axid;
    }
    if (ranges && A.n_nb > 0) {      // rows wider than the LDS span: where the columns of a row's records are kept between its ranges
        long long *jrec = nullptr, *rowrec = nullptr, *xoff = nullptr;
        int *xl = nullptr;
        XM_HIP(xm_malloc_async((void **)&jrec, sizeof(long long) * (size_t)(I > 0 ? I : 1), st));
        XM_HIP(xm_malloc_async((void **)&rowrec, sizeof(long long) * (size_t)A.n_nb, st));
        XM_HIP(xm_malloc_async((void **)&xoff, sizeof(long long) * ((size_t)A.n_nb + 1), st));
        k_joint_records<<<dim3((unsigned)((I + 3) / 4)), dim3(256), 0, st>>>(I, jptr, joff, A.src_ptr, A.src_idx, A.att_ptr, jrec);
        XM_LAUNCH_CHECK();
        k_row_records<<<dim3((unsigned)((A.n_nb + 3) / 4)), dim3(256), 0, st>>>(A.n_nb, A.k, A.nb_list, A.kcnt, A.kcol, A.flags, jrec, rowrec);
        XM_LAUNCH_CHECK();
        int64_t total = 0;
        rc = xmap_exclusive_scan_i64(st, (const int64_t *)rowrec, (int64_t *)xoff, A.n_nb, &total);
        if (rc) return rc;
        XM_HIP(xm_malloc_async((void **)&xl, sizeof(int) * (size_t)(total > 0 ? total : 1), st));
        A.xoff = xoff; A.xl = xl;
    }
    return XMAP_OK;
}
static int mid_rows_lds(int32_t n_nb, size_t *bytes, int *span) {
    int cap = XMAP_MID_ROWS_SPAN;
    if (const char *e = getenv("XMAP_MID_ROWS_SPAN")) {      // tests: several column ranges on small inputs
        const int v = atoi(e);
        if (v >= 1 && v < cap) cap = v;
    }
    *span = n_nb < cap ? n_nb : cap;
    *bytes = sizeof(int32_t) * (size_t)(*span > 0 ? *span : 1);
    return XMAP_OK;
}

int xmap_mid_rows_count(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                        const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                        const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                        const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                        int32_t *ng /*[n_nb]*/, int64_t *nrec /*[n_nb]*/) {
    XM_ARG(cls && kcnt && kcol && kval && flags && att_ptr && src_ptr && nb_list && nb_id && ng && nrec);
    if (n_nb == 0) return XMAP_OK;
    size_t lds;
    int span;
    int rc = mid_rows_lds(n_nb, &lds, &span);
    if (rc) return rc;
    MidArgs A = mid_args(n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val,
                         src_flag, n_nb, nb_list, nb_id);
    XM_ARG(src_flag);
    XM_SCOPE(stream);
    rc = mid_joints((hipStream_t)stream, A, span < n_nb);
    if (rc) return rc;
    if (span >= n_nb) {
        XM_HIP(hipFuncSetAttribute((const void *)k_mid_rows<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_mid_rows<0, true><<<dim3((unsigned)n_nb), dim3(64 * MIDROW_WAVES), lds, (hipStream_t)stream>>>(A, span, ng, (long long *)nrec,
                                                                                                       nullptr, nullptr, nullptr);
    } else {
        XM_HIP(hipFuncSetAttribute((const void *)k_mid_rows<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_mid_rows<0, false><<<dim3((unsigned)n_nb), dim3(64 * MIDROW_WAVES), lds, (hipStream_t)stream>>>(A, span, ng, (long long *)nrec,
                                                                                                        nullptr, nullptr, nullptr);
    }
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_mid_rows_place(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                        const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                        const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                        const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                        const int64_t *dir_ptr /*[n_nb+1]*/, const int64_t *rec_ptr /*[n_nb+1]*/, void *dir, void *midX) {
    XM_ARG(cls && kcnt && kcol && kval && flags && att_ptr && src_ptr && nb_list && nb_id);
    XM_ARG(dir_ptr && rec_ptr && dir && midX);
    if (n_nb == 0) return XMAP_OK;
    size_t lds;
    int span;
    int rc = mid_rows_lds(n_nb, &lds, &span);
    if (rc) return rc;
    MidArgs A = mid_args(n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val,
                         src_flag, n_nb, nb_list, nb_id);
    A.midX = (MidX *)midX;
    XM_ARG(src_flag);
    XM_SCOPE(stream);
    rc = mid_joints((hipStream_t)stream, A, span < n_nb);
    if (rc) return rc;
    if (span >= n_nb) {
        XM_HIP(hipFuncSetAttribute((const void *)k_mid_rows<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_mid_rows<1, true><<<dim3((unsigned)n_nb), dim3(64 * MIDROW_WAVES), lds, (hipStream_t)stream>>>(
            A, span, nullptr, nullptr, (const long long *)dir_ptr, (const long long *)rec_ptr, (MidDir *)dir);
    } else {
        XM_HIP(hipFuncSetAttribute((const void *)k_mid_rows<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_mid_rows<1, false><<<dim3((unsigned)n_nb), dim3(64 * MIDROW_WAVES), lds, (hipStream_t)stream>>>(
            A, span, nullptr, nullptr, (const long long *)dir_ptr, (const long long *)rec_ptr, (MidDir *)dir);
    }
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

#ifdef XMAP_CROSSCHECK
int xmap_mid_tally(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                   const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                   const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                   const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                   int32_t *tile_cnt /*[n_nb*n_nb], zeroed here*/, int32_t *ng /*[n_nb]*/) {
    XM_ARG(cls && kcnt && kcol && kval && flags && att_ptr && src_ptr && nb_list && nb_id && tile_cnt && ng);
    if (n_nb == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    MidArgs A = mid_args(n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val,
                         src_flag, n_nb, nb_list, nb_id);
    A.tile_cnt = tile_cnt;
    XM_HIP(hipMemsetAsync(tile_cnt, 0, sizeof(int32_t) * (size_t)n_nb * (size_t)n_nb, st));
    const long long waves = (long long)n_nb * top_k;
    k_mid_build<false><<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st>>>(A);
    XM_LAUNCH_CHECK();
    k_mid_dir<false><<<dim3((unsigned)((n_nb + 3) / 4)), dim3(256), 0, st>>>(n_nb, tile_cnt, nullptr, ng, nullptr, nullptr,
                                                                             nullptr, nullptr);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_mid_place(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                   const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                   const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                   const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                   int32_t *tile_cnt, const int64_t *tile_off /*[n_nb*n_nb+1]*/, const int64_t *dir_ptr /*[n_nb+1]*/,
                   void *dir /*24 B per tile*/, void *midX /*64 B per record*/) {
    XM_ARG(cls && kcnt && kcol && kval && flags && att_ptr && src_ptr && nb_list && nb_id);
    XM_ARG(tile_cnt && tile_off && dir_ptr && dir && midX);
    if (n_nb == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    MidArgs A = mid_args(n_items, top_k, cls, kcnt, kcol, kval, flags, att_ptr, att_idx, att_val, src_ptr, src_idx, src_val,
                         src_flag, n_nb, nb_list, nb_id);
    A.tile_cnt = tile_cnt; A.tile_off = (const long long *)tile_off; A.midX = (MidX *)midX;
    k_mid_dir<true><<<dim3((unsigned)((n_nb + 3) / 4)), dim3(256), 0, st>>>(n_nb, tile_cnt, (const long long *)tile_off,
                                                                            nullptr, (const long long *)dir_ptr, (MidDir *)dir, nb_list, kcnt);
    XM_LAUNCH_CHECK();
    XM_HIP(hipMemsetAsync(tile_cnt, 0, sizeof(int32_t) * (size_t)n_nb * (size_t)n_nb, st));   // now the placement cursors
    const long long waves = (long long)n_nb * top_k;
    k_mid_build<true><<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st>>>(A);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

#endif  // XMAP_CROSSCHECK

int xmap_edge_ranges(void *stream, const xmap_sim *S, int32_t *h_fast_ok) {
    XM_SCOPE(stream);
    XM_ARG(S && h_fast_ok);
    *h_fast_ok = 1;
    if (S->n_items == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    long long n = 0;
    XM_HIP(hipMemcpyAsync(&n, S->row_ptr + S->n_items, sizeof(long long), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    if (n == 0) return XMAP_OK;
    if (S->frac) { *h_fast_ok = 0; return XMAP_OK; }      // caller-supplied fractions: generic records, take the checked division
    int *bad = nullptr;
    XM_HIP(xm_malloc_async((void **)&bad, sizeof(int), st));
    XM_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    k_edge_ranges<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, S->sim, S->mutu, bad);
    XM_LAUNCH_CHECK();
    int h = 0;
    XM_HIP(hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    XM_HIP(xm_free_async(bad, st));
    *h_fast_ok = h ? 0 : 1;
    return XMAP_OK;
}

int xmap_end_universe(void *stream, const xmap_ext_tables *T, int32_t *mark /*[I] scratch*/, int64_t *rank /*[I+1] scratch*/,
                      int32_t *urank /*[I]*/, int32_t *uitem /*[I]*/, int64_t *h_n_ends) {
    XM_ARG(T && mark && rank && urank && uitem && h_n_ends);
    const int I = T->n_items, k = T->top_k;
    *h_n_ends = 0;
    if (I == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    long long h_n[2];
    XM_HIP(hipMemcpyAsync(&h_n[0], T->src_ptr + I, sizeof(long long), hipMemcpyDeviceToHost, st));
    XM_HIP(hipMemcpyAsync(&h_n[1], T->att_ptr + I, sizeof(long long), hipMemcpyDeviceToHost, st));
    XM_HIP(hipMemsetAsync(mark, 0, sizeof(int32_t) * (size_t)I, st));
    XM_HIP(hipStreamSynchronize(st));
    long long n = (long long)I * k;
    if (h_n[0] > n) n = h_n[0];
    if (h_n[1] > n) n = h_n[1];
    k_mark_ends<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(I, k, T->cls, T->kcnt, T->kcol, h_n[0], T->src_idx, h_n[1],
                                                                          T->att_idx, mark);
    XM_LAUNCH_CHECK();
    int rc = xmap_exclusive_scan_i32_to_i64(stream, mark, rank, I, h_n_ends);
    if (rc) return rc;
    k_end_ranks<<<dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st>>>(I, mark, (const long long *)rank, urank, uitem);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_extend_cols_slots(int32_t *h_n_slots) {
    XM_ARG(h_n_slots);
    int dev = 0;
    hipDeviceProp_t prop;
    XM_HIP(hipGetDevice(&dev));
    XM_HIP(hipGetDeviceProperties(&prop, dev));
    *h_n_slots = prop.multiProcessorCount * 4 * P_WAVES;
    return XMAP_OK;
}

int xmap_extend_cols(void *stream, const xmap_ext_tables *T, const xmap_path_units *Un, const xmap_path_rows *R,
                      const xmap_path_out *O, int fast_div, int64_t *d_counters, int64_t *h_counters) {
    XM_SCOPE(stream);
    XM_ARG(T && Un && R && O && d_counters);
    XM_ARG(T->cls && T->kcnt && T->kcol && T->kval && T->flags && T->att_ptr && T->src_ptr && T->rnn_ptr);
    XM_ARG(T->n_ends >= 0 && (T->n_items == 0 || (T->urank && T->uitem)));
    XM_ARG(R->n_slots > 0 && R->acc && R->touched && O->n_cand && O->top_end && O->top_val);
    XM_ARG(Un->n_units >= 0 && Un->n_heavy >= 0);
    XM_ARG(Un->n_units == 0 || (Un->unit_start && Un->unit_c && Un->unit_G && Un->unit_row && Un->unit_nt));
    XM_ARG(Un->n_units == 0 || T->n_nb == 0 || (T->nb_id && T->nb_list && T->midX && T->dir && T->dir_ptr));
    XM_ARG(Un->n_heavy == 0 || (Un->heavy_unit0 && R->hacc && R->htouched));
    XM_ARG(O->xs_cap == 0 || (O->xs_off && O->xs_end && O->xs_val));
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(hipMemsetAsync(d_counters, 0, 8 * sizeof(int64_t), st));
    if (Un->n_units > 0) {
        Path2Args B;
        memset(&B, 0, sizeof(B));
        PathArgs &A = B.P;
        A.I = T->n_items; A.k = T->top_k;
        A.cls = T->cls; A.kcnt = T->kcnt; A.kcol = T->kcol; A.kval = T->kval; A.flags = T->flags;
        A.att_ptr = (const long long *)T->att_ptr; A.att_idx = T->att_idx; A.att_val = T->att_val;
        A.src_ptr = (const long long *)T->src_ptr; A.src_idx = T->src_idx; A.src_val = T->src_val; A.src_flag = T->src_flag;
        A.rnn_ptr = (const long long *)T->rnn_ptr; A.rnn_idx = T->rnn_idx; A.rnn_val = T->rnn_val;
        A.n_units = Un->n_units; A.unit_start = Un->unit_start; A.unit_c = Un->unit_c; A.unit_G = Un->unit_G;
        A.unit_row = Un->unit_row; A.unit_nt = Un->unit_nt;
        A.acc = R->acc; A.touched = R->touched; A.hacc = R->hacc; A.htouched = R->htouched;
        A.n_cand = O->n_cand; A.top_end = O->top_end; A.top_val = O->top_val;
        A.xs_cap = O->xs_cap; A.xs_off = (long long *)O->xs_off; A.xs_end = O->xs_end; A.xs_val = O->xs_val;
        A.counters = (unsigned long long *)d_counters;
        A.U = T->n_ends; A.urank = T->urank; A.uitem = T->uitem;
        A.n_slots = R->n_slots < Un->n_units ? R->n_slots : Un->n_units;
        B.nb_id = T->nb_id; B.nb_list = T->nb_list; B.n_nb = T->n_nb; B.midX = (const MidX *)T->midX; B.dir = (const MidDir *)T->dir;
        B.dir_ptr = (const long long *)T->dir_ptr; B.ng = nullptr;
        A.row_stride = T->n_ends;
#ifdef Q_STORE      // ablation build: rows of one slot per (column, end); the caller sizes the rows (XMAP_ABL_ROW_ENTRIES)
        if (const char *env = getenv("XMAP_ABL_ROW_ENTRIES")) A.row_stride = atoll(env);
        XM_ARG(A.row_stride >= (long long)T->n_nb * (T->top_k + 1) && A.row_stride >= T->n_ends);
#endif
        ColEnd *cend = nullptr;
        int *home = nullptr;
        if (T->n_nb > 0) {
            const long long n = (long long)T->n_nb * (T->top_k + 1);
            XM_HIP(xm_malloc_async((void **)&cend, sizeof(ColEnd) * (size_t)n, st));
            XM_HIP(xm_malloc_async((void **)&home, sizeof(int) * (size_t)T->n_items, st));
            XM_HIP(hipMemsetAsync(home, 0x7f, sizeof(int) * (size_t)T->n_items, st));
            const dim3 cgrid((unsigned)((n + 255) / 256));
            k_col_home<<<cgrid, dim3(256), 0, st>>>(T->n_nb, T->top_k, T->nb_list, T->kcnt, T->kcol, home);
            XM_LAUNCH_CHECK();
            k_col_ends<<<cgrid, dim3(256), 0, st>>>(T->n_nb, T->top_k, T->nb_list, T->kcnt, T->kcol, T->kval, T->urank, home, cend);
            XM_LAUNCH_CHECK();
            XM_HIP(xm_free_async(home, st));
        }
        B.cend = cend;
        const dim3 grid((unsigned)((A.n_slots + 3) / 4)), block(256);
        if (fast_div) k_paths4<true><<<grid, block, 0, st>>>(B);
        else k_paths4<false><<<grid, block, 0, st>>>(B);
        XM_LAUNCH_CHECK();
        if (cend) XM_HIP(xm_free_async(cend, st));
        if (Un->n_heavy > 0) {
            k_merge_groups<<<dim3((unsigned)Un->n_heavy, 16), dim3(64 * MERGE_WAVES), 0, st>>>(A, Un->n_heavy, Un->heavy_unit0);
            XM_LAUNCH_CHECK();
            k_merge<<<dim3((unsigned)Un->n_heavy), dim3(64 * MERGE_WAVES), 0, st>>>(A, Un->n_heavy, Un->heavy_unit0);
            XM_LAUNCH_CHECK();
        }
    }
    if (h_counters) {
        XM_HIP(hipMemcpyAsync(h_counters, d_counters, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (O->xs_cap > 0 && h_counters[0] > O->xs_cap) {
            set_error("candidate buffer too small: need %lld entries, have %lld", (long long)h_counters[0], (long long)O->xs_cap);
            return XMAP_ERR_CAPACITY;
        }
    }
    return XMAP_OK;
}
}
