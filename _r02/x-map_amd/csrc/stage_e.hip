// stage_e.hip -- RecommenderPrivacy.nonprivate_neighbor_selection (reference
// core/recommenderPrivacy.py:22-35,141-152): per item the mapping_range neighbours with the largest |similarity| out of
// the RecommenderSim rows (Engine.rec_sim).  The reference sorts each neighbour list with a stable sort on -|sim|, so
// equal similarities keep the order the pairs arrived in, which Spark does not define; the canonical order here is
// (|sim| desc, neighbour index asc), as everywhere else.
//
//   k_rec_select : one wave per row.  The row streams through in chunks of 64 entries; the wave keeps the best
//                  `keep` (<= 64) entries as a sorted list in registers, one entry per lane (the same DPP
//                  shift-insert as the dense variant's ranking, stage_d.hip); a chunk costs one compare + ballot,
//                  a candidate one 64-bit compare + popcount, a wave shift and a readlane.
#include "common.h"

namespace xmap {

__device__ __forceinline__ unsigned dpp_shr1_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
__device__ __forceinline__ double dpp_shr1_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = dpp_shr1_u32((unsigned)(b & 0xffffffffull)), hi = dpp_shr1_u32((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void k_rec_select(int I, const long long *row_ptr, const int *col, const double *sim,
                                                    const double *ls, int keep, int *out_cnt, int *out_col,
                                                    double *out_sim, double *out_ls) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= I) return;
    const int lane = lane_id();
    const long long p0 = row_ptr[row], p1 = row_ptr[row + 1];
    // the list: entry p on lane p, sorted best first; empty entries have |sim| = -1 (below every candidate) and index -1
    double Ls = -1.0, Ll = 0.0;
    int Lc = -1;
    const unsigned long long kmask = keep >= 64 ? ~0ull : ((1ull << keep) - 1ull);
    double thr = -1.0;       // |sim| of the keep-th entry (uniform)
    int thr_c = -1;          // ... and its index
    for (long long base = p0; base < p1; base += 64) {
        const long long p = base + lane;
        const bool in = p < p1;
        const double s = in ? sim[p] : 0.0, l = in ? ls[p] : 0.0;
        const int c = in ? col[p] : 0x7fffffff;
        const double a = fabs(s);
        unsigned long long m = __ballot(in && (a > thr || (a == thr && (unsigned)c < (unsigned)thr_c)));
        while (m) {
            const int t = __ffsll((long long)m) - 1;
            m &= m - 1;
            const double cs = rld(s, t), cl = rld(l, t);
            const int cc = rl32(c, t);
            const double ca = fabs(cs);
            // position = number of kept entries that order before the candidate
            const double la = fabs(Ls);
            const bool before = (Lc >= 0) && (la > ca || (la == ca && Lc < cc));
            const int pos = __popcll(__ballot(before) & kmask);
            if (pos >= keep) continue;     // an earlier candidate of this chunk raised the bar
            const double ss = dpp_shr1_f64(Ls), sl = dpp_shr1_f64(Ll);
            const int sc = (int)dpp_shr1_u32((unsigned)Lc);
            Ls = lane > pos ? ss : (lane == pos ? cs : Ls);
            Ll = lane > pos ? sl : (lane == pos ? cl : Ll);
            Lc = lane > pos ? sc : (lane == pos ? cc : Lc);
            thr_c = rl32(Lc, keep - 1);
            thr = (thr_c >= 0) ? fabs(rld(Ls, keep - 1)) : -1.0;
        }
    }
    const int cnt = __popcll(__ballot(Lc >= 0) & kmask);
    if (lane == 0) out_cnt[row] = cnt;
    if (lane < keep) {
        const size_t o = (size_t)row * keep + lane;
        out_col[o] = (lane < cnt) ? Lc : -1;
        out_sim[o] = (lane < cnt) ? Ls : 0.0;
        out_ls[o] = (lane < cnt) ? Ll : 0.0;
    }
}

// ---- RecommenderPrediction.item_based_prediction (core/recommenderPrediction.py:26-105) ------------------------------------
// One thread per test pair (user, item): the evidence of the item's selected neighbours -- every rating the user gave a
// neighbour: (sim * (rating - neighbour average), |sim|, time) in neighbour-list order, a neighbour's ratings in list order --
// then  base + sum(ev0) / sum(ev1)  (Python sums: left to right) and the decayed form: evidence sorted by time (stable),
// equal times share a rank, weight exp(-alpha (now - rank)) from a host-made table (numpy's exp), sums in sorted order.
// Results are bound_rating()'s: round half up, clamp to [0, 5].
constexpr int PRED_EV = 64;        // evidence entries per pair (mapping_range neighbours x the user's ratings of each)
__device__ __forceinline__ double bound_rating(double r) {
    long long v = (long long)(r + 0.5);             // int(): towards zero
    v = v < 5 ? v : 5;
    v = v > 0 ? v : 0;
    return 1.0 * (double)v;
}

__global__ __launch_bounds__(128) void k_predict(long long n_test, const int *tu, const int *ti, const long long *nb_ptr,
                                                 const int *nb_item, const double *nb_sim, const long long *rt_ptr, const int *rt_user,
                                                 const double *rt_rating, const double *rt_time, const double *avg, const double *wtab,
                                                 int n_w, double *out_plain, double *out_decay, int *status) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_test) return;
    const int u = tu[t], it = ti[t];
    if (it < 0) { status[t] = 1; out_plain[t] = 0.0; out_decay[t] = 0.0; return; }       // item without neighbours: ()
    const double base = avg[it];
    double e0[PRED_EV], e1[PRED_EV], tm[PRED_EV];
    int n = 0;
    bool over = false;
    for (long long p = nb_ptr[it]; p < nb_ptr[it + 1] && !over; p++) {
        const int nb = nb_item[p];
        const double nsim = nb_sim[p], navg = avg[nb];
        if (u < 0) continue;
        long long lo = rt_ptr[nb], hi = rt_ptr[nb + 1];
        const long long end = hi;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (rt_user[mid] < u) lo = mid + 1; else hi = mid; }
        for (; lo < end && rt_user[lo] == u; lo++) {
            if (n == PRED_EV) { over = true; break; }
            e0[n] = nsim * (rt_rating[lo] - navg); e1[n] = fabs(nsim); tm[n] = rt_time[lo];
            n++;
        }
    }
    if (over || n + 1 > n_w) { status[t] = 2; out_plain[t] = 0.0; out_decay[t] = 0.0; return; }       // the host decides this pair
    double plain = base, decayed = base;
    if (n > 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < n; k++) { s0 += e0[k]; s1 += e1[k]; }
        plain = base + s0 / s1;
        int ord[PRED_EV];
        for (int k = 0; k < n; k++) {                 // stable insertion sort by time
            int q = k;
            while (q > 0 && tm[ord[q - 1]] > tm[k]) { ord[q] = ord[q - 1]; q--; }
            ord[q] = k;
        }
        int rank[PRED_EV], r = 0;
        for (int k = 0; k < n; k++) {
            if (k == 0 || tm[ord[k]] != tm[ord[k - 1]]) r++;
            rank[k] = r;
        }
        const int now = r + 1;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < n; k++) {
            const double w = wtab[now - rank[k]];
            a += e0[ord[k]] * w; b += e1[ord[k]] * w;
        }
        decayed = base + a / b;
    }
    status[t] = 0;
    out_plain[t] = bound_rating(plain);
    out_decay[t] = bound_rating(decayed);
}

}  // namespace xmap
using namespace xmap;

extern "C" {

int xmap_predict(void *stream, int64_t n_test, const int32_t *test_user, const int32_t *test_item, const int64_t *nb_ptr,
                 const int32_t *nb_item, const double *nb_sim, const int64_t *rt_ptr, const int32_t *rt_user, const double *rt_rating,
                 const double *rt_time, const double *item_avg, const double *wtab, int32_t n_w, double *out_plain,
                 double *out_decay, int32_t *status) {
    XM_ARG(test_user && test_item && nb_ptr && nb_item && nb_sim && rt_ptr && rt_user && rt_rating && rt_time && item_avg && wtab);
    XM_ARG(out_plain && out_decay && status && n_test >= 0 && n_w >= 2);
    if (n_test == 0) return XMAP_OK;
    k_predict<<<dim3((unsigned)((n_test + 127) / 128)), dim3(128), 0, (hipStream_t)stream>>>(
        n_test, test_user, test_item, (const long long *)nb_ptr, nb_item, nb_sim, (const long long *)rt_ptr, rt_user, rt_rating, rt_time,
        item_avg, wtab, n_w, out_plain, out_decay, status);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_rec_select(void *stream, int32_t n_items, const int64_t *row_ptr, const int32_t *col, const double *sim,
                    const double *ls, int32_t keep, int32_t *out_cnt, int32_t *out_col, double *out_sim, double *out_ls) {
    XM_ARG(row_ptr && col && sim && ls && out_cnt && out_col && out_sim && out_ls);
    XM_ARG(keep >= 1 && keep <= 64 && n_items >= 0);
    if (n_items == 0) return XMAP_OK;
    k_rec_select<<<dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        n_items, (const long long *)row_ptr, col, sim, ls, keep, out_cnt, out_col, out_sim, out_ls);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}
