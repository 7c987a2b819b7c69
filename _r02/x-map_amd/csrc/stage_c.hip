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

// build_alterEgo (generator.py:140-157): one thread per user; profiles are short, the grouping of
// avoid_duplicate_ratings (generator.py:123-138) is done by re-scanning the profile (first-seen order).
template <bool FILL>
__global__ __launch_bounds__(256) void k_alterego(long long U, const long long *ptr, const int *item, const float *rating,
                                                  const long long *time, const uint8_t *flags, const int *map,
                                                  int *cnt_t, int *cnt_m, const long long *off_t,
                                                  const long long *off_m, long long n_t_total, int *out_user,
                                                  int *out_item, double *out_rating, long long *out_time) {
    long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    long long a = ptr[u], b = ptr[u + 1];
    long long ot = FILL ? off_t[u] : 0, om = FILL ? n_t_total + off_m[u] : 0;
    int ct = 0, cm = 0;
    for (long long e = a; e < b; e++) {
        int it = item[e];
        if (flags[it] & 2) {  // "T:" in iid: pass-through row
            if (FILL) {
                out_user[ot] = (int)u; out_item[ot] = it; out_rating[ot] = (double)rating[e]; out_time[ot] = time[e];
                ot++;
            }
            ct++;
        }
        int m = map[it];
        if (m < 0) continue;
        bool first = true;
        for (long long f = a; f < e; f++)
            if (map[item[f]] == m) { first = false; break; }
        if (!first) continue;
        if (FILL) {
            double s = 0.0;
            int n = 0;
            for (long long f = e; f < b; f++)
                if (map[item[f]] == m) { s += (double)rating[f]; n++; }
            out_user[om] = (int)u; out_item[om] = m; out_rating[om] = s / (double)n;      // np.mean of the group (fp64, generator.py:134)
            out_time[om] = time[e];
            om++;
        }
        cm++;
    }
    if (!FILL) { cnt_t[u] = ct; cnt_m[u] = cm; }
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

int xmap_alterego_count(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, int32_t *cnt_t, int32_t *cnt_m) {
    XM_ARG(R && map_src2tgt && cnt_t && cnt_m);
    if (R->n_users == 0) return XMAP_OK;
    k_alterego<false><<<dim3((unsigned)((R->n_users + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating, (const long long *)R->user_time,
        R->flags, map_src2tgt, cnt_t, cnt_m, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_alterego_fill(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, const int64_t *off_t,
                       const int64_t *off_m, int64_t n_t_total, int32_t *out_user, int32_t *out_item, double *out_rating,
                       int64_t *out_time) {
    XM_ARG(R && map_src2tgt && off_t && off_m && out_user && out_item && out_rating && out_time);
    if (R->n_users == 0) return XMAP_OK;
    k_alterego<true><<<dim3((unsigned)((R->n_users + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        R->n_users, (const long long *)R->user_ptr, R->user_item, R->user_rating, (const long long *)R->user_time,
        R->flags, map_src2tgt, nullptr, nullptr, (const long long *)off_t, (const long long *)off_m, n_t_total, out_user,
        out_item, out_rating, (long long *)out_time);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}
