// stage_c.hip -- replacement selection + AlterEgo profile aggregation (generator_pipeline,
// reference utils/assist.py:136-150, :210-215; core/generator.py).  HBM-bound scans over the CSR.
#include "common.h"

namespace xmap {

// cross_private_mapping degenerates to arg-max |xsim| under Python 3 (generator.py:38,66-67,91-97);
// cross_nonprivate_mapping picks top4[np.random.randint(0, len-1)] (generator.py:100-111), the draw is the host's.
// map_to_dict (assist.py:210-215): {choice: start}, last writer in ascending start order wins = max start.
__global__ __launch_bounds__(256) void k_select_map(int I, int private_flag, const int *n_cand, const int *top_end,
                                                    const int *picks, int *n_top, int *choice, int *map) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= I) return;
    int n = n_cand[s];
    int keep = private_flag ? XMAP_TOPC : 4;
    int m = n < keep ? n : keep;
    n_top[s] = m;
    int c = -1;
    if (m > 0) {
        int idx = private_flag ? 0 : (picks ? picks[s] : 0);
        if (idx < 0 || idx >= m) idx = 0;
        c = top_end[(size_t)s * XMAP_TOPC + idx];
        atomicMax(&map[c], s);
    }
    choice[s] = c;
}

// build_alterEgo (generator.py:140-157), one profile entry per lane: four profiles of up to 16 ratings per wave, one per
// 16-lane group (90 % of the users at BASELINE configs[1]), longer ones on the whole wave in chunks of 64 -- coalesced
// loads of the profile; the first-seen grouping of avoid_duplicate_ratings (generator.py:123-138) by broadcasting every
// entry through its group (sums in profile order, like the reference's list); rows written in runs.  Rounds 1-2 gave every
// user ONE thread that walked the profile alone and re-scanned it per mapped entry: 1.1 ms for the fill pass, 3 ms under a
// long run (driver run of round 2), and a tail of users with long profiles.
// PROF_SHARDS counters for the users with output rows (one word would serialise an atomic per wave).
constexpr int PROF_SHARDS = 64;

template <bool FILL>
__device__ __forceinline__ void alterego_group16(bool on, long long u, long long a, int d, int gl, int gbase, const int *item,
                                                 const float *rating, const long long *time, const uint8_t *flags,
                                                 const int *map, int *cnt_t, int *cnt_m, const long long *off_t,
                                                 const long long *off_m, long long n_t_total, int *out_user, int *out_item,
                                                 double *out_rating, long long *out_time, int &any) {
    const bool act = on && gl < d;
    int it = -1, m = -1;
    float r = 0.f;
    bool is_t = false;
    if (act) {
        it = item[a + gl];
        r = rating[a + gl];
        is_t = (flags[it] & 2) != 0;          // "T:" in iid: pass-through row
        m = map[it];
    }
    bool first = act && m >= 0;
    double s = 0.0;
    int n = 0;
    for (int t = 0; t < 16; t++) {            // entry t of the group's profile, broadcast
        const int mt = __shfl(m, gbase + t, 64);
        const float rt = __shfl(r, gbase + t, 64);
        if (act && m >= 0 && mt == m) {
            if (t < gl) first = false;
            s += (double)rt; n++;
        }
    }
    const unsigned long long gmask = 0xffffull << gbase;
    const unsigned long long bt = __ballot(act && is_t) & gmask, bm = __ballot(first) & gmask;
    const unsigned long long lt = lanemask_lt();
    if (!FILL) {
        if (on && gl == 0) {
            const int ct = __popcll(bt), cm = __popcll(bm);
            cnt_t[u] = ct; cnt_m[u] = cm;
            any = (ct + cm) > 0;
        }
        return;
    }
    if (!on) return;
    if (act && is_t) {
        const long long o = off_t[u] + __popcll(bt & lt);
        out_user[o] = (int)u; out_item[o] = it; out_rating[o] = (double)r; out_time[o] = time[a + gl];
    }
    if (first) {
        const long long o = n_t_total + off_m[u] + __popcll(bm & lt);
        out_user[o] = (int)u; out_item[o] = m; out_rating[o] = s / (double)n;      // np.mean of the group (fp64, generator.py:134)
        out_time[o] = time[a + gl];
    }
}

// a profile of any length on the whole wave, 64 entries at a time; every chunk is compared with all chunks (the earlier
// ones decide "seen before", all of them add to the mean)
template <bool FILL>
__device__ __forceinline__ void alterego_wave(long long u, long long a, int d, int lane, const int *item, const float *rating,
                                              const long long *time, const uint8_t *flags, const int *map, int *cnt_t,
                                              int *cnt_m, const long long *off_t, const long long *off_m, long long n_t_total,
                                              int *out_user, int *out_item, double *out_rating, long long *out_time, int &any) {
    int ct = 0, cm = 0;
    for (int cb = 0; cb < d; cb += 64) {
        const int e = cb + lane;
        const bool act = e < d;
        int it = -1, m = -1;
        float r = 0.f;
        bool is_t = false;
        if (act) {
            it = item[a + e];
            r = rating[a + e];
            is_t = (flags[it] & 2) != 0;
            m = map[it];
        }
        bool first = act && m >= 0;
        double s = 0.0;
        int n = 0;
        for (int ob = 0; ob < d; ob += 64) {
            int om = m;
            float orr = r;
            if (ob != cb) {
                om = -1; orr = 0.f;
                if (ob + lane < d) { om = map[item[a + ob + lane]]; orr = rating[a + ob + lane]; }
            }
            const int lim = min(64, d - ob);
            for (int t = 0; t < lim; t++) {
                const int mt = __shfl(om, t, 64);
                const float rt = __shfl(orr, t, 64);
                if (act && m >= 0 && mt == m) {
                    if (ob + t < e) first = false;
                    s += (double)rt; n++;
                }
            }
        }
        const unsigned long long bt = __ballot(act && is_t), bm = __ballot(first);
        const unsigned long long lt = lanemask_lt();
        if (FILL) {
            if (act && is_t) {
                const long long o = off_t[u] + ct + __popcll(bt & lt);
                out_user[o] = (int)u; out_item[o] = it; out_rating[o] = (double)r; out_time[o] = time[a + e];
            }
            if (first) {
                const long long o = n_t_total + off_m[u] + cm + __popcll(bm & lt);
                out_user[o] = (int)u; out_item[o] = m; out_rating[o] = s / (double)n;
                out_time[o] = time[a + e];
            }
        }
        ct += __popcll(bt); cm += __popcll(bm);
    }
    if (!FILL && lane == 0) {
        cnt_t[u] = ct; cnt_m[u] = cm;
        any = (ct + cm) > 0;
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_alterego_grp(long long U, const long long *ptr, const int *item, const float *rating,
                                                      const long long *time, const uint8_t *flags, const int *map, int *cnt_t,
                                                      int *cnt_m, const long long *off_t, const long long *off_m,
                                                      long long n_t_total, int *out_user, int *out_item, double *out_rating,
                                                      long long *out_time, unsigned long long *n_prof) {
    const long long u0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (u0 >= U) return;
    const int lane = lane_id();
    int any = 0;
    {
        const int g = lane >> 4;
        const long long u = u0 + g;
        long long a = 0;
        int d = 17;
        if (u < U) { a = ptr[u]; d = (int)min(ptr[u + 1] - a, 17ll); }
        alterego_group16<FILL>(u < U && d <= 16, u, a, d, lane & 15, g << 4, item, rating, time, flags, map, cnt_t, cnt_m, off_t,
                               off_m, n_t_total, out_user, out_item, out_rating, out_time, any);
    }
    for (int q = 0; q < 4; q++) {
        const long long u = u0 + q;
        if (u >= U) break;
        const long long a = ptr[u];
        const long long dl = ptr[u + 1] - a;
        if (dl <= 16) continue;
        int any1 = 0;
        alterego_wave<FILL>(u, a, (int)dl, lane, item, rating, time, flags, map, cnt_t, cnt_m, off_t, off_m, n_t_total, out_user,
                            out_item, out_rating, out_time, any1);
        any += any1;        // (lane 0 carries it)
    }
    if (!FILL && n_prof) {
        const int tot = (int)wave_sum_ll((long long)any);
        if (lane == 0 && tot) atomicAdd(&n_prof[blockIdx.x & (PROF_SHARDS - 1)], (unsigned long long)tot);
    }
}

}  // namespace xmap

using namespace xmap;

extern "C" {

int xmap_select_map(void *stream, int32_t n_items, int private_flag, const int32_t *n_cand, const int32_t *top_end,
                    const int32_t *picks, int32_t *n_top, int32_t *choice, int32_t *map_src2tgt) {
    XM_ARG(n_cand && top_end && n_top && choice && map_src2tgt);
    hipStream_t st = (hipStream_t)stream;
    if (n_items == 0) return XMAP_OK;
    XM_HIP(hipMemsetAsync(map_src2tgt, 0xff, sizeof(int32_t) * (size_t)n_items, st));
    k_select_map<<<dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st>>>(n_items, private_flag, n_cand, top_end,
                                                                                  picks, n_top, choice, map_src2tgt);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_alterego_count(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, int32_t *cnt_t, int32_t *cnt_m,
                        int64_t *d_profiles /* [64] zeroed by the caller, or NULL */) {
    XM_ARG(R && map_src2tgt && cnt_t && cnt_m);
    if (R->n_users == 0) return XMAP_OK;
    k_alterego_grp<false><<<dim3((unsigned)((R->n_users + 15) / 16)), dim3(256), 0, (hipStream_t)stream>>>(
        R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating, (const long long *)R->user_time,
        R->flags, map_src2tgt, cnt_t, cnt_m, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
        (unsigned long long *)d_profiles);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_alterego_fill(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, const int64_t *off_t,
                       const int64_t *off_m, int64_t n_t_total, int32_t *out_user, int32_t *out_item, double *out_rating,
                       int64_t *out_time) {
    XM_ARG(R && map_src2tgt && off_t && off_m && out_user && out_item && out_rating && out_time);
    if (R->n_users == 0) return XMAP_OK;
    k_alterego_grp<true><<<dim3((unsigned)((R->n_users + 15) / 16)), dim3(256), 0, (hipStream_t)stream>>>(
        R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating, (const long long *)R->user_time,
        R->flags, map_src2tgt, nullptr, nullptr, (const long long *)off_t, (const long long *)off_m, n_t_total, out_user,
        out_item, out_rating, (long long *)out_time, nullptr);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}
