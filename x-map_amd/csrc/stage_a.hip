// stage_a.hip -- item-item cosine / adjusted-cosine over sparse user x item ratings
// (baseliner_calculate_sim_pipeline, reference utils/assist.py:66-77; core/baselinerSim.py).
//
// Design (DESIGN.md section "Stage A"):
//   * row-wise Gustavson over the CSC (item -> raters) and CSR (user -> profile) copies that stay
//     resident in HBM; a work unit is (item i, hash partition q of i's partner space);
//   * ONE WAVE owns a unit: it walks the raters of i in ascending user index (the order the
//     reference's reduceByKey concatenates co-raters in) and, per rater, spreads the profile over
//     the 64 lanes with coalesced 4-byte index + 4-byte rating loads (bit 31 of the index carries
//     `rating >= item average`, so mutuality needs no further gather);
//   * per-partner partials (n_ij, mutuality, fp64 dot) live in a wave-private open-addressing
//     table in LDS (1024 slots, 20 KB); lanes of one step hold distinct partners, so the
//     accumulate is a plain LDS read-modify-write and the fp64 sum runs in rater order --
//     deterministic, symmetric in (i,j) bit for bit, and equal to the reference's own sum for
//     every pair with fewer than 8 co-raters (all pairs in cosine mode);
//   * finalisation (cosine, significance weighting, zero filter) is done from LDS; a count pass
//     sizes the CSR output, a fill pass writes it (col, sim fp64, mutu, n_ij).
#include "common.h"
#include "item_stats.h"

namespace xmap {

#ifdef XMAP_CROSSCHECK      // (constants of the complete-rows pair kernel)
constexpr int A_THREADS = 64;   // one wave = one unit = one workgroup (no block-level barrier is used)
constexpr int A_WAVES = A_THREADS / 64;
constexpr int LOG_SLOTS = 10;
constexpr int SLOTS = 1 << LOG_SLOTS;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
#endif

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_user_stats(long long U, const long long *ptr, const float *rating,
                                                    double *avg, double *norm2) {
    long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    long long a = ptr[u], b = ptr[u + 1];
    double s = 0.0, q = 0.0;
    for (long long e = a; e < b; e++) {
        double r = (double)rating[e];
        s += r;
        q += r * r;
    }
    avg[u] = (b > a) ? s / (double)(b - a) : 0.0;
    norm2[u] = sqrt(q);
}

// CSC (item -> raters) from the CSR: count, scan, fill.  The order of an item's raters is whatever the cursor
// atomics produce -- every consumer sums exactly (integer-valued or double-double), so no order is needed.
// Popular items (8e4 raters at BASELINE configs[1]) would serialise that many atomics on one word, which is what
// bounds a naive version (~90 same-address atomics per microsecond): every workgroup therefore first counts its
// entries in a direct-mapped LDS cache of CSC_SLOTS (item, count) slots and goes to memory once per occupied slot;
// entries whose slot is taken by another item use the global word directly.
constexpr int CSC_SLOTS = 4096;
constexpr int CSC_CHUNK = 8192;    // entries per workgroup of the count kernel
constexpr int CSC_USERS = 512;     // users per workgroup of the fill kernel
__device__ __forceinline__ int csc_slot(int it) { return (int)(mix32((uint32_t)it) & (CSC_SLOTS - 1)); }

__global__ __launch_bounds__(256) void k_csc_count(long long nnz, const int *uitem, int *cnt) {
    __shared__ int tag[CSC_SLOTS], loc[CSC_SLOTS];
    for (int t = threadIdx.x; t < CSC_SLOTS; t += 256) { tag[t] = -1; loc[t] = 0; }
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * CSC_CHUNK;
    for (int q = threadIdx.x; q < CSC_CHUNK; q += 256) {
        const long long e = e0 + q;
        if (e >= nnz) break;
        const int it = uitem[e];
        const int sl = csc_slot(it);
        const int old = atomicCAS(&tag[sl], -1, it);
        if (old == -1 || old == it) atomicAdd(&loc[sl], 1); else atomicAdd(&cnt[it], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CSC_SLOTS; t += 256)
        if (loc[t]) atomicAdd(&cnt[tag[t]], loc[t]);
}

// wave per user (the rater id is the user), CSC_USERS users per workgroup
__global__ __launch_bounds__(256) void k_csc_fill(long long U, const long long *uptr, const int *uitem, const float *urating,
                                                  const long long *iptr, int *cur, int *iuser, float *irating) {
    __shared__ int tag[CSC_SLOTS], loc[CSC_SLOTS], base[CSC_SLOTS];
    for (int t = threadIdx.x; t < CSC_SLOTS; t += 256) { tag[t] = -1; loc[t] = 0; }
    __syncthreads();
    const long long u0 = (long long)blockIdx.x * CSC_USERS;
    const long long u1 = min(U, u0 + CSC_USERS);
    const int w = threadIdx.x >> 6, lane = lane_id();
    // pass 1: claim slots, count this workgroup's entries per cached item
    for (long long u = u0 + w; u < u1; u += 4)
        for (long long e = uptr[u] + lane; e < uptr[u + 1]; e += 64) {
            const int it = uitem[e];
            const int sl = csc_slot(it);
            const int old = atomicCAS(&tag[sl], -1, it);
            if (old == -1 || old == it) atomicAdd(&loc[sl], 1);
        }
    __syncthreads();
    // pass 2: one range of the item's segment per occupied slot
    for (int t = threadIdx.x; t < CSC_SLOTS; t += 256) {
        if (loc[t]) base[t] = atomicAdd(&cur[tag[t]], loc[t]);
        loc[t] = 0;
    }
    __syncthreads();
    // pass 3: positions inside the reserved ranges (LDS counter), or straight from the global cursor
    for (long long u = u0 + w; u < u1; u += 4)
        for (long long e = uptr[u] + lane; e < uptr[u + 1]; e += 64) {
            const int it = uitem[e];
            const int sl = csc_slot(it);
            const int off = (tag[sl] == it) ? base[sl] + atomicAdd(&loc[sl], 1) : atomicAdd(&cur[it], 1);
            const long long p = iptr[it] + off;
            iuser[p] = (int)u;
            irating[p] = urating[e];
        }
}

// four items per wave: the ones with at most 64 raters (99 % at BASELINE configs[1]; the median item has 10) together,
// one per 16-lane group; the others one after the other on the whole wave
// items [lo, hi) (a rank's share when the items are sharded: the per-item results are all-gathered afterwards)
__global__ __launch_bounds__(256) void k_item_stats(int I, int lo, int hi, const long long *iptr, const int *iuser, const float *irating,
                                                    const double *u_avg, double *info, double *norms, int *ia_user,
                                                    double *partial = nullptr) {
    const int i0 = lo + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (i0 >= hi) return;
    const int lane = lane_id();
    {
        const int i = i0 + (lane >> 4);
        const bool on = i < hi && iptr[i + 1] - iptr[i] <= 64;
        item_stats_group<16>(on, i, lane & 15, I, iptr, CscSrc{iuser, irating, u_avg}, info, norms, ia_user, partial);
    }
    for (int t = 0; t < 4; t++) {
        const int i = i0 + t;
        if (i >= hi) break;
        if (iptr[i + 1] - iptr[i] <= 64) continue;
        item_stats_group<64>(true, i, lane, I, iptr, CscSrc{iuser, irating, u_avg}, info, norms, ia_user, partial);
    }
}

// user-sharded input: the item sums of the ranks' user shares [n_parts][I][5] = (sum r, sum r^2, adjusted
// norm^2 as an exact (value, error) pair, raters) are added up in rank order -- the adjusted norm exactly -- and finished
// as k_item_stats finishes them
__global__ __launch_bounds__(256) void k_item_merge(int I, int n_parts, const double *parts, double *info, double *norms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0, q = 0.0, a2 = 0.0, a2lo = 0.0, n = 0.0;
    for (int r = 0; r < n_parts; r++) {
        const double *o = parts + ((size_t)r * I + i) * 5;
        s += o[0]; q += o[1]; n += o[4];
        dd_add(a2, a2lo, o[2]);
        dd_add(a2, a2lo, o[3]);
    }
    const double avg = (n > 0.0) ? 1.0 * s / n : 0.0;
    info[(size_t)i * 4 + 0] = avg;
    info[(size_t)i * 4 + 1] = sqrt(q);
    info[(size_t)i * 4 + 2] = sqrt(a2);
    info[(size_t)i * 4 + 3] = 1.0 * n;
    norms[i] = sqrt(q);
    norms[(size_t)I + i] = sqrt(a2);
}

#ifdef XMAP_CROSSCHECK      // ---- the complete-rows formulation of round 1 (algo="rows"): a test formulation, built into libxmap_hip_xcheck.so only
__global__ __launch_bounds__(256) void k_pack_user_side(long long nnz, const int *uitem, const float *urating,
                                                        const double *info, int *ua_item) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int j = uitem[e];
    unsigned ge = ((double)urating[e] >= info[(size_t)j * 4]) ? 0x80000000u : 0u;
    ua_item[e] = (int)((unsigned)j | ge);
}

// W_i = sum over raters of (profile length - 1); Q_i = ceil(min(W_i, I-1) / target)
__global__ __launch_bounds__(256) void k_plan(int I, const long long *iptr, const int *iuser, const long long *uptr,
                                              int target, int *Q, long long *W, unsigned long long *contrib) {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    int lane = lane_id();
    long long p0 = iptr[i], p1 = iptr[i + 1];
    long long w = 0;
    for (long long p = p0 + lane; p < p1; p += 64) {
        int u = iuser[p];
        w += uptr[u + 1] - uptr[u] - 1;
    }
    w = wave_sum_ll(w);
    if (lane == 0) {
        long long m = w < (long long)(I - 1) ? w : (long long)(I - 1);
        Q[i] = (int)((m + target - 1) / target);
        W[i] = w;
        if (w) atomicAdd(contrib, (unsigned long long)w);
    }
}

__global__ __launch_bounds__(256) void k_fill_units(int I, const int *Q, const long long *unit_ptr, int *unit_item,
                                                    int *unit_q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    long long b = unit_ptr[i];
    int q = Q[i];
    for (int k = 0; k < q; k++) {
        unit_item[b + k] = i;
        unit_q[b + k] = k;
    }
}

__global__ __launch_bounds__(256) void k_row_ptr(int I, const long long *unit_ptr, const long long *unit_off,
                                                 long long *row_ptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > I) return;
    row_ptr[i] = unit_off[unit_ptr[i]];
}

// ---------------------------------------------------------------------------------------------
struct PairArgs {
    const long long *uptr;
    const int *ua_item;
    const float *urating;
    const long long *iptr;
    const int *ia_user;
    const float *irating;
    const double *u_avg;
    const double *info;
    const int *Q;
    const int *unit_item;
    const int *unit_q;
    long long unit_lo, unit_hi;
    int cap;
    // count pass
    int *unit_cnt;
    unsigned long long *counters;  // [0] kept, [1] evaluated, [2] overflow
    // fill pass
    const long long *unit_off;
    int *col;
    double *sim;
    int *mutu;
    int *nij;
};

template <int METHOD, bool WRITE>
__global__ __launch_bounds__(A_THREADS) void k_pair_sim(PairArgs A) {
    __shared__ uint32_t s_key[A_WAVES][SLOTS];
    __shared__ uint32_t s_cnt[A_WAVES][SLOTS];
    __shared__ uint32_t s_mut[A_WAVES][SLOTS];
    __shared__ double s_dot[A_WAVES][SLOTS];
    // adjusted-cosine terms are summed error-free (double-double): the low words live here
    __shared__ double s_lo[METHOD == XMAP_ADJUST_COSINE ? A_WAVES : 1][METHOD == XMAP_ADJUST_COSINE ? SLOTS : 1];

    const int w = threadIdx.x >> 6;
    const int lane = lane_id();
    const long long unit = A.unit_lo + (long long)blockIdx.x * A_WAVES + w;
    if (unit >= A.unit_hi) return;  // no block-level barrier is used below: waves are independent

    uint32_t *key = s_key[w];
    uint32_t *cnt = s_cnt[w];
    uint32_t *mut = s_mut[w];
    double *dot = s_dot[w];
    double *dlo = s_lo[METHOD == XMAP_ADJUST_COSINE ? w : 0];
    for (int s = lane; s < SLOTS; s += 64) {
        key[s] = EMPTY;
        cnt[s] = 0;
        mut[s] = 0;
        dot[s] = 0.0;
        if (METHOD == XMAP_ADJUST_COSINE) dlo[s] = 0.0;
    }

    const int i = uniform(A.unit_item[unit]);
    const int q = uniform(A.unit_q[unit]);
    const int Qi = uniform(A.Q[i]);
    const int p0 = uniform((int)A.iptr[i]);
    const int p1 = uniform((int)A.iptr[i + 1]);
    int ovf = 0;

    for (int base = p0; base < p1; base += 64) {
        // lane-parallel fetch of up to 64 raters of item i (ascending user index)
        int p = base + lane;
        int uw = 0, e0 = 0, e1 = 0;
        float r = 0.f;
        double au = 0.0;
        if (p < p1) {
            uw = A.ia_user[p];
            r = A.irating[p];
            int u = uw & 0x7fffffff;
            e0 = (int)A.uptr[u];
            e1 = (int)A.uptr[u + 1];
            if (METHOD == XMAP_ADJUST_COSINE) au = A.u_avg[u];
        }
        const int nr = (p1 - base) < 64 ? (p1 - base) : 64;
        // software prefetch of the next rater's first 64 profile entries
        int nb0 = rl32(e0, 0), nb1 = rl32(e1, 0);
        int njw = 0;
        float nrj = 0.f;
        if (nb0 + lane < nb1) {
            njw = A.ua_item[nb0 + lane];
            nrj = A.urating[nb0 + lane];
        }
        for (int t = 0; t < nr; ++t) {
            const int b0 = nb0, b1 = nb1;
            int jw = njw;
            float rj = nrj;
            if (t + 1 < nr) {
                nb0 = rl32(e0, t + 1);
                nb1 = rl32(e1, t + 1);
                if (nb0 + lane < nb1) {
                    njw = A.ua_item[nb0 + lane];
                    nrj = A.urating[nb0 + lane];
                }
            }
            if (b1 - b0 < 2) continue;  // users with >= 2 ratings only (baselinerSim.py:184-185)
            const double ri = (double)rlf(r, t);
            const unsigned gei = ((unsigned)rl32(uw, t)) >> 31;
            const double a = (METHOD == XMAP_ADJUST_COSINE) ? rld(au, t) : 0.0;
            for (int c0 = b0; c0 < b1; c0 += 64) {
                const int e = c0 + lane;
                if (c0 != b0 && e < b1) {
                    jw = A.ua_item[e];
                    rj = A.urating[e];
                }
                if (e < b1) {
                    const int j = jw & 0x7fffffff;
                    bool mine = (j != i);
                    if (mine && Qi > 1) mine = (int)__umulhi(mix32((uint32_t)j), (uint32_t)Qi) == q;
                    if (mine) {
                        uint32_t h = ((uint32_t)j * 0x9E3779B1u) >> (32 - LOG_SLOTS);
                        int probes = 0;
                        bool ok = true;
                        for (;;) {
                            uint32_t prev = atomicCAS(&key[h], EMPTY, (uint32_t)j);
                            if (prev == EMPTY || prev == (uint32_t)j) break;
                            h = (h + 1) & (SLOTS - 1);
                            if (++probes >= SLOTS) { ok = false; break; }
                        }
                        if (ok) {
                            const unsigned gej = ((unsigned)jw) >> 31;
                            cnt[h] += 1;
                            mut[h] += (gej == gei) ? 1u : 0u;
                            if (METHOD == XMAP_COSINE) {
                                dot[h] += (1.0 * ri) * (double)rj;   // exact for integer ratings
                            } else {
                                double hi = dot[h], lo = dlo[h];
                                dd_add(hi, lo, (ri - a) * ((double)rj - a));
                                dot[h] = hi;
                                dlo[h] = lo;
                            }
                        } else {
                            ovf = 1;
                        }
                    }
                }
            }
        }
    }

    // finalise from LDS: cosine (:91-95), significance weighting (:84-89), zero filter (:198,:207)
    const int c1 = (METHOD == XMAP_COSINE) ? 1 : 2;
    const double norm_i = A.info[(size_t)i * 4 + c1];
    long long out = WRITE ? A.unit_off[unit] : 0;
    int kept = 0, evald = 0;
    for (int s0 = 0; s0 < SLOTS; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t kj = key[s];
        const bool occ = kj != EMPTY;
        bool keep = false;
        double simv = 0.0;
        int n = 0, m = 0;
        if (occ) {
            n = (int)cnt[s];
            m = (int)mut[s];
            const double np = norm_i * A.info[(size_t)kj * 4 + c1];
            const double cs = (np != 0.0) ? 1.0 * dot[s] / np : 0.0;
            const int mn = n < A.cap ? n : A.cap;
            simv = 1.0 * cs * (double)mn / (double)A.cap;
            keep = (simv != 0.0) && (m != 0);  // frac_mutu != 0 <=> mutu != 0
        }
        const unsigned long long km = __ballot(keep);
        evald += __popcll(__ballot(occ));
        if (WRITE && keep) {
            const long long o = out + __popcll(km & lanemask_lt());
            A.col[o] = (int)kj;
            A.sim[o] = simv;
            A.mutu[o] = m;
            A.nij[o] = n;
        }
        const int c = __popcll(km);
        out += c;
        kept += c;
    }
    if (!WRITE && lane == 0) {
        A.unit_cnt[unit] = kept;
        atomicAdd(&A.counters[0], (unsigned long long)kept);
        atomicAdd(&A.counters[1], (unsigned long long)evald);
    }
    if (__ballot(ovf) && lane == 0) atomicOr(&A.counters[2], 1ull);
}

static int launch_pair(hipStream_t st, int method, bool write, const PairArgs &A) {
    long long n = A.unit_hi - A.unit_lo;
    if (n <= 0) return XMAP_OK;
    dim3 grid((unsigned)((n + A_WAVES - 1) / A_WAVES)), block(A_THREADS);
    if (method == XMAP_COSINE) {
        if (write) k_pair_sim<XMAP_COSINE, true><<<grid, block, 0, st>>>(A);
        else k_pair_sim<XMAP_COSINE, false><<<grid, block, 0, st>>>(A);
    } else {
        if (write) k_pair_sim<XMAP_ADJUST_COSINE, true><<<grid, block, 0, st>>>(A);
        else k_pair_sim<XMAP_ADJUST_COSINE, false><<<grid, block, 0, st>>>(A);
    }
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

static PairArgs make_args(const xmap_ratings *R, int cap, const double *u_avg, const double *info, const int32_t *ua_item,
                          const int32_t *ia_user, const int32_t *Q, const int32_t *unit_item, const int32_t *unit_q,
                          int64_t lo, int64_t hi) {
    PairArgs A;
    memset(&A, 0, sizeof(A));
    A.uptr = (const long long *)R->user_ptr;
    A.ua_item = ua_item;
    A.urating = R->user_rating;
    A.iptr = (const long long *)R->item_ptr;
    A.ia_user = ia_user;
    A.irating = R->item_rating;
    A.u_avg = u_avg;
    A.info = info;
    A.Q = Q;
    A.unit_item = unit_item;
    A.unit_q = unit_q;
    A.unit_lo = lo;
    A.unit_hi = hi;
    A.cap = cap;
    return A;
}

#endif  // XMAP_CROSSCHECK

}  // namespace xmap

using namespace xmap;

extern "C" {

int xmap_build_csc(void *stream, int64_t n_users, int32_t n_items, int64_t nnz, const int64_t *user_ptr,
                   const int32_t *user_item, const float *user_rating, int32_t *cnt /*[I] scratch*/,
                   int64_t *item_ptr /*[I+1]*/, int32_t *item_user, float *item_rating) {
    XM_ARG(user_ptr && cnt && item_ptr && (nnz == 0 || (user_item && user_rating && item_user && item_rating)));
    XM_ARG(nnz < 0x7fffffffLL);
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(n_items > 0 ? n_items : 1), st));
    if (nnz > 0) {
        k_csc_count<<<dim3((unsigned)((nnz + CSC_CHUNK - 1) / CSC_CHUNK)), dim3(256), 0, st>>>(nnz, user_item, cnt);
        XM_LAUNCH_CHECK();
    }
    int rc = xmap_exclusive_scan_i32_to_i64(stream, cnt, item_ptr, n_items, nullptr);
    if (rc) return rc;
    XM_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(n_items > 0 ? n_items : 1), st));
    if (nnz > 0) {
        k_csc_fill<<<dim3((unsigned)((n_users + CSC_USERS - 1) / CSC_USERS)), dim3(256), 0, st>>>(
            n_users, (const long long *)user_ptr, user_item, user_rating, (const long long *)item_ptr, cnt, item_user,
            item_rating);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

int xmap_user_stats(void *stream, const xmap_ratings *R, double *u_avg, double *u_norm2) {
    XM_ARG(R && u_avg && u_norm2);
    if (R->n_users == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    k_user_stats<<<dim3((unsigned)((R->n_users + 255) / 256)), dim3(256), 0, st>>>(
        R->n_users, (const long long *)R->user_ptr, R->user_rating, u_avg, u_norm2);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_item_stats(void *stream, const xmap_ratings *R, const double *u_avg, double *info, double *norms,
                    int32_t *ua_item, int32_t *ia_user, int32_t item_lo, int32_t item_hi) {
    XM_ARG(R && u_avg && info && ((ua_item != nullptr) == (ia_user != nullptr)));
    XM_ARG(R->nnz < 0x7fffffffLL && item_lo >= 0 && item_lo <= item_hi && item_hi <= R->n_items);
    hipStream_t st = (hipStream_t)stream;
    if (item_hi > item_lo) {
        k_item_stats<<<dim3((unsigned)((item_hi - item_lo + 15) / 16)), dim3(256), 0, st>>>(
            R->n_items, item_lo, item_hi, (const long long *)R->item_ptr, R->item_user, R->item_rating, u_avg, info, norms, ia_user);
        XM_LAUNCH_CHECK();
    }
#ifdef XMAP_CROSSCHECK      // (the packed per-rating arrays are read by the complete-rows formulation only)
    if (R->nnz > 0 && ua_item) {
        k_pack_user_side<<<dim3((unsigned)((R->nnz + 255) / 256)), dim3(256), 0, st>>>(
            R->nnz, R->user_item, R->user_rating, info, ua_item);
        XM_LAUNCH_CHECK();
    }
#else
    XM_ARG(ua_item == nullptr);
#endif
    return XMAP_OK;
}

int xmap_item_partials(void *stream, const xmap_ratings *R, const double *u_avg, double *partial /*[I][5]*/) {
    XM_ARG(R && u_avg && partial && R->nnz < 0x7fffffffLL);
    hipStream_t st = (hipStream_t)stream;
    if (R->n_items > 0) {
        k_item_stats<<<dim3((unsigned)((R->n_items + 15) / 16)), dim3(256), 0, st>>>(
            R->n_items, 0, R->n_items, (const long long *)R->item_ptr, R->item_user, R->item_rating, u_avg, nullptr, nullptr, nullptr,
            partial);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

int xmap_item_merge(void *stream, int32_t n_items, int32_t n_parts, const double *parts, double *info, double *norms) {
    XM_ARG(parts && info && norms && n_items >= 0 && n_parts >= 1);
    if (n_items > 0) {
        k_item_merge<<<dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(n_items, n_parts, parts, info, norms);
        XM_LAUNCH_CHECK();
    }
    return XMAP_OK;
}

#ifdef XMAP_CROSSCHECK
int xmap_sim_plan(void *stream, const xmap_ratings *R, int32_t slot_target, int32_t *Q, int64_t *W, int64_t *unit_ptr,
                  int64_t *h_n_units, int64_t *h_contrib) {
    XM_SCOPE(stream);
    XM_ARG(R && Q && W && unit_ptr && slot_target > 0 && slot_target <= SLOTS);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d_contrib = nullptr;
    XM_HIP(xm_malloc_async((void **)&d_contrib, sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(d_contrib, 0, sizeof(unsigned long long), st));
    if (R->n_items > 0) {
        k_plan<<<dim3((unsigned)((R->n_items + 3) / 4)), dim3(256), 0, st>>>(
            R->n_items, (const long long *)R->item_ptr, R->item_user, (const long long *)R->user_ptr, slot_target, Q,
            (long long *)W, d_contrib);
        XM_LAUNCH_CHECK();
    }
    int rc = xmap_exclusive_scan_i32_to_i64(stream, Q, unit_ptr, R->n_items, h_n_units);
    if (rc) return rc;
    unsigned long long hc = 0;
    XM_HIP(hipMemcpyAsync(&hc, d_contrib, sizeof(hc), hipMemcpyDeviceToHost, st));
    XM_HIP(hipStreamSynchronize(st));
    XM_HIP(xm_free_async(d_contrib, st));
    if (h_contrib) *h_contrib = (int64_t)hc;
    return XMAP_OK;
}

int xmap_sim_units(void *stream, int32_t n_items, const int32_t *Q, const int64_t *unit_ptr, int32_t *unit_item,
                   int32_t *unit_q) {
    XM_ARG(Q && unit_ptr && unit_item && unit_q);
    if (n_items == 0) return XMAP_OK;
    k_fill_units<<<dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, Q, (const long long *)unit_ptr, unit_item, unit_q);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_sim_row_ptr(void *stream, int32_t n_items, const int64_t *unit_ptr, const int64_t *unit_off, int64_t *row_ptr) {
    XM_ARG(unit_ptr && unit_off && row_ptr);
    k_row_ptr<<<dim3((unsigned)((n_items + 256) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, (const long long *)unit_ptr, (const long long *)unit_off, (long long *)row_ptr);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_sim_count(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg, const double *info,
                   const int32_t *ua_item, const int32_t *ia_user, const int32_t *Q, const int32_t *unit_item,
                   const int32_t *unit_q, int64_t unit_lo, int64_t unit_hi, int32_t *unit_cnt, int64_t *d_counters,
                   int64_t *h_counters) {
    XM_ARG(R && u_avg && info && ua_item && ia_user && Q && unit_item && unit_q && unit_cnt && d_counters);
    XM_ARG(method == XMAP_COSINE || method == XMAP_ADJUST_COSINE);
    XM_ARG(cap > 0 && R->nnz < 0x7fffffffLL);
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(hipMemsetAsync(d_counters, 0, 4 * sizeof(int64_t), st));
    PairArgs A = make_args(R, cap, u_avg, info, ua_item, ia_user, Q, unit_item, unit_q, unit_lo, unit_hi);
    A.unit_cnt = unit_cnt;
    A.counters = (unsigned long long *)d_counters;
    int rc = launch_pair(st, method, false, A);
    if (rc) return rc;
    if (h_counters) {
        XM_HIP(hipMemcpyAsync(h_counters, d_counters, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (h_counters[2]) {
            set_error("pair-table overflow: lower slot_target and re-plan");
            return XMAP_ERR_OVERFLOW;
        }
    }
    return XMAP_OK;
}

int xmap_sim_fill(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg, const double *info,
                  const int32_t *ua_item, const int32_t *ia_user, const int32_t *Q, const int32_t *unit_item,
                  const int32_t *unit_q, int64_t unit_lo, int64_t unit_hi, const int64_t *unit_off, int32_t *col,
                  double *sim, int32_t *mutu, int32_t *nij) {
    XM_SCOPE(stream);
    XM_ARG(R && u_avg && info && ua_item && ia_user && Q && unit_item && unit_q && unit_off);
    XM_ARG(method == XMAP_COSINE || method == XMAP_ADJUST_COSINE);
    XM_ARG(cap > 0 && R->nnz < 0x7fffffffLL);
    PairArgs A = make_args(R, cap, u_avg, info, ua_item, ia_user, Q, unit_item, unit_q, unit_lo, unit_hi);
    A.unit_off = (const long long *)unit_off;
    A.col = col;
    A.sim = sim;
    A.mutu = mutu;
    A.nij = nij;
    unsigned long long *dummy = nullptr;
    hipStream_t st = (hipStream_t)stream;
    XM_HIP(xm_malloc_async((void **)&dummy, 4 * sizeof(unsigned long long), st));
    XM_HIP(hipMemsetAsync(dummy, 0, 4 * sizeof(unsigned long long), st));
    A.counters = dummy;
    int rc = launch_pair(st, method, true, A);
    XM_HIP(xm_free_async(dummy, st));
    return rc;
}
#endif  // XMAP_CROSSCHECK
}
