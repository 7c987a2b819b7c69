// plan.hip -- the planning steps of stage B as kernels of the library (round 1 did them with torch ops on the device:
// sort, cumsum, repeat_interleave, nonzero).  A stable LSD radix sort of (key, value) pairs, the index of the non-bridge
// records, the work units of the path enumeration (heaviest first, heavy starts cut into chunks) and the order of the end
// universe (the ends of a column side by side).
#include "common.h"

namespace xmap {

// ---- stable LSD radix sort, 4 bits per pass (the lists sorted here have 1e5 - 1e6 entries: a few tens of microseconds
// per pass; a pass = block histograms -> one exclusive scan of the digit-major table -> ordered scatter) ---------------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

// per block: cnt[d][t] = keys of thread t (its RS_ITEMS consecutive keys) with digit d; then, per digit, the exclusive
// prefix over the threads (in place) and the block total
__device__ __forceinline__ void rs_local(const unsigned long long *keys, long long n, int shift, unsigned short (*cnt)[RS_THREADS],
                                         int *tot) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)t * RS_ITEMS;
    for (int d = 0; d < 16; d++) cnt[d][t] = 0;
    for (int i = 0; i < RS_ITEMS; i++)
        if (base + i < n) cnt[(keys[base + i] >> shift) & 15][t]++;
    __syncthreads();
    if (t < 16) {           // one lane per digit walks the 256 thread counts (tiny)
        int run = 0;
        for (int u = 0; u < RS_THREADS; u++) { const int c = cnt[t][u]; cnt[t][u] = (unsigned short)run; run += c; }
        tot[t] = run;
    }
    __syncthreads();
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const unsigned long long *keys, long long n, int shift, int n_blocks, int *hist) {
    __shared__ unsigned short cnt[16][RS_THREADS];
    __shared__ int tot[16];
    rs_local(keys, n, shift, cnt, tot);
    if (threadIdx.x < 16) hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = tot[threadIdx.x];      // digit-major
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const unsigned long long *keys, const int *vals, long long n, int shift,
                                                           int n_blocks, const long long *hist_off, unsigned long long *keys_out,
                                                           int *vals_out) {
    __shared__ unsigned short cnt[16][RS_THREADS];
    __shared__ int tot[16];
    rs_local(keys, n, shift, cnt, tot);
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)t * RS_ITEMS;
    int seen[16];
#pragma unroll
    for (int d = 0; d < 16; d++) seen[d] = 0;
    for (int i = 0; i < RS_ITEMS; i++) {
        if (base + i >= n) break;
        const unsigned long long k = keys[base + i];
        const int d = (int)((k >> shift) & 15);
        int mine = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) if (e == d) { mine = seen[e]; seen[e]++; }
        const long long pos = hist_off[(size_t)d * n_blocks + blockIdx.x] + cnt[d][t] + mine;
        keys_out[pos] = k;
        vals_out[pos] = vals[base + i];
    }
}

// sorts (keys, vals) ascending by the low `bits` bits of the key, stable; result in (keys, vals); tmp buffers of n
int radix_sort_pairs(hipStream_t st, unsigned long long *keys, int *vals, unsigned long long *keys_tmp, int *vals_tmp,
                     long long n, int bits) {
    if (n <= 1) return XMAP_OK;
    const int n_blocks = (int)((n + RS_TILE - 1) / RS_TILE);
    int *hist = nullptr;
    long long *off = nullptr;
    XM_HIP(xm_malloc_async((void **)&hist, sizeof(int) * (size_t)16 * n_blocks, st));
    XM_HIP(xm_malloc_async((void **)&off, sizeof(long long) * ((size_t)16 * n_blocks + 1), st));
    unsigned long long *ka = keys, *kb = keys_tmp;
    int *va = vals, *vb = vals_tmp;
    int passes = 0;
    for (int shift = 0; shift < bits; shift += 4, passes++) {
        k_rs_hist<<<dim3((unsigned)n_blocks), dim3(RS_THREADS), 0, st>>>(ka, n, shift, n_blocks, hist);
        XM_LAUNCH_CHECK();
        int rc = xmap_exclusive_scan_i32_to_i64(st, hist, (int64_t *)off, (int64_t)16 * n_blocks, nullptr);
        if (rc) return rc;
        k_rs_scatter<<<dim3((unsigned)n_blocks), dim3(RS_THREADS), 0, st>>>(ka, va, n, shift, n_blocks, off, kb, vb);
        XM_LAUNCH_CHECK();
        unsigned long long *tk = ka; ka = kb; kb = tk;
        int *tv = va; va = vb; vb = tv;
    }
    if (passes & 1) {       // the result sits in the tmp buffers
        XM_HIP(hipMemcpyAsync(keys, ka, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToDevice, st));
        XM_HIP(hipMemcpyAsync(vals, va, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    XM_HIP(xm_free_async(hist, st));
    XM_HIP(xm_free_async(off, st));
    return XMAP_OK;
}

// ---- index of the non-bridge records ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flag_cls(int I, const uint8_t *cls, int want, int *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < I) flag[i] = cls[i] == want;
}
__global__ __launch_bounds__(256) void k_nb_fill(int I, const int *flag, const long long *rank, int *nb_list, int *nb_id) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    nb_id[i] = flag[i] ? (int)rank[i] : -1;
    if (flag[i]) nb_list[rank[i]] = i;
}

// ---- work units of the path enumeration ------------------------------------------------------------------------------
// G[s] = chunks of start s (0: no paths / outside the range), cost key for the order
__global__ __launch_bounds__(256) void k_plan_chunks(int I, const long long *P, int lo, int hi, long long chunk, int *G, long long *counters) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= I) return;
    const long long p = (s >= lo && s < hi) ? P[s] : 0;
    const int g = p > chunk ? (int)((p + chunk - 1) / chunk) : (p > 0 ? 1 : 0);
    G[s] = g;
    if (g > 1) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)g);       // dedicated rows
    if (p > 0) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)p);       // paths in the range
}
__global__ __launch_bounds__(256) void k_flag_pos(int I, const int *G, int *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < I) flag[i] = G[i] > 0;
}
__global__ __launch_bounds__(256) void k_plan_keys(int I, const long long *P, const int *G, const long long *rank, unsigned long long *keys,
                                                   int *vals) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= I || G[s] == 0) return;
    const long long cost = P[s] / G[s];                      // paths per unit of this start
    const unsigned long long cap = (1ull << 44) - 1;
    keys[rank[s]] = cap - ((unsigned long long)cost < cap ? (unsigned long long)cost : cap);      // ascending key = descending cost
    vals[rank[s]] = s;
}
__global__ __launch_bounds__(256) void k_plan_gather(int n, const int *order, const int *G, int *g_sorted, int *gh_sorted, int *isheavy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = G[order[i]];
    g_sorted[i] = g;
    gh_sorted[i] = g > 1 ? g : 0;
    isheavy[i] = g > 1;
}
__global__ __launch_bounds__(256) void k_plan_expand(int n, const int *order, const int *g_sorted, const long long *unit_off,
                                                     const long long *row_off, const long long *heavy_rank, int *unit_start, int *unit_c,
                                                     int *unit_G, int *unit_row, int *heavy_unit0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = g_sorted[i], s = order[i];
    const long long u0 = unit_off[i];
    for (int c = 0; c < g; c++) {
        unit_start[u0 + c] = s; unit_c[u0 + c] = c; unit_G[u0 + c] = g;
        unit_row[u0 + c] = g > 1 ? (int)(row_off[i] + c) : -1;
    }
    if (g > 1) heavy_unit0[heavy_rank[i]] = (int)u0;
}

// ---- order of the end universe: every end with the first column whose end list holds it ---------------------------------
__global__ __launch_bounds__(256) void k_end_home(int n_nb, int k, const int *nb_list, const int *kcnt, const int *kcol, int *home) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_nb * (k + 1)) return;
    const int xid = (int)(t / (k + 1)), idx = (int)(t % (k + 1));
    const int x = nb_list[xid];
    int e = -1;
    if (idx == 0) e = x;
    else if (idx - 1 < kcnt[(size_t)x * 2 + 1]) e = kcol[((size_t)x * 2 + 1) * k + (idx - 1)];
    if (e >= 0) atomicMin(&home[e], x);
}
__global__ __launch_bounds__(256) void k_home_keys(int nU, const int *uitem, const int *home, unsigned long long *keys) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nU) return;
    const int e = uitem[r];
    keys[r] = ((unsigned long long)(unsigned)home[e] << 21) | (unsigned)e;      // home = 0x7fffffff for ends of no column: last
}
__global__ __launch_bounds__(256) void k_inverse(int nU, const int *uitem, int *urank) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nU) urank[uitem[r]] = r;
}

}  // namespace xmap

using namespace xmap;

extern "C" {

int xmap_nb_index(void *stream, int32_t n_items, const uint8_t *cls, int32_t *nb_list, int32_t *nb_id, int64_t *h_n_nb) {
    XM_ARG(cls && nb_list && nb_id && h_n_nb);
    *h_n_nb = 0;
    if (n_items == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    int *flag = nullptr;
    long long *rank = nullptr;
    XM_HIP(xm_malloc_async((void **)&flag, sizeof(int) * (size_t)n_items, st));
    XM_HIP(xm_malloc_async((void **)&rank, sizeof(long long) * ((size_t)n_items + 1), st));
    const dim3 grid((unsigned)((n_items + 255) / 256)), block(256);
    k_flag_cls<<<grid, block, 0, st>>>(n_items, cls, 2, flag);
    XM_LAUNCH_CHECK();
    int rc = xmap_exclusive_scan_i32_to_i64(stream, flag, (int64_t *)rank, n_items, h_n_nb);
    if (rc) return rc;
    k_nb_fill<<<grid, block, 0, st>>>(n_items, flag, rank, nb_list, nb_id);
    XM_LAUNCH_CHECK();
    XM_HIP(xm_free_async(flag, st));
    XM_HIP(xm_free_async(rank, st));
    return XMAP_OK;
}

int xmap_path_plan(void *stream, int32_t n_items, const int64_t *paths, int32_t start_lo, int32_t start_hi, int64_t chunk,
                   int64_t chunk_div, int64_t max_rows, int64_t cap_units, int32_t *unit_start, int32_t *unit_c, int32_t *unit_G, int32_t *unit_row,
                   int32_t *heavy_unit0, int64_t *h_out /*[5]: units, heavy starts, rows, paths, chunk*/) {
    XM_ARG(paths && h_out && (chunk > 0 || chunk_div > 0) && cap_units >= 0);
    for (int i = 0; i < 5; i++) h_out[i] = 0;
    h_out[4] = chunk;
    if (n_items == 0) return XMAP_OK;
    hipStream_t st = (hipStream_t)stream;
    const int I = n_items;
    const dim3 grid((unsigned)((I + 255) / 256)), block(256);
    int *ibuf = nullptr;
    long long *lbuf = nullptr;
    unsigned long long *keys = nullptr;
    XM_HIP(xm_malloc_async((void **)&ibuf, sizeof(int) * (size_t)I * 7, st));
    int *G = ibuf, *flag = G + I, *vals = flag + I, *vals_t = vals + I, *gs = vals_t + I, *gh = gs + I, *ish = gh + I;
    XM_HIP(xm_malloc_async((void **)&lbuf, sizeof(long long) * ((size_t)4 * (I + 1) + 2), st));
    long long *cnt = lbuf, *rank = cnt + 2, *uoff = rank + (I + 1), *roff = uoff + (I + 1), *hrank = roff + (I + 1);
    XM_HIP(xm_malloc_async((void **)&keys, sizeof(unsigned long long) * (size_t)I * 2, st));
    unsigned long long *keys_t = keys + I;
    long long h_cnt[2];
    if (chunk_div > 0) {    // chunk from the total: max(2^22, paths in the range / chunk_div)
        XM_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(long long), st));
        k_plan_chunks<<<grid, block, 0, st>>>(I, (const long long *)paths, start_lo, start_hi, (long long)1 << 62, G, cnt);
        XM_LAUNCH_CHECK();
        XM_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        chunk = h_cnt[1] / chunk_div;
        if (chunk < (1 << 22)) chunk = 1 << 22;
    }
    for (;;) {      // chunk grows until the dedicated rows of the heavy starts fit the budget
        XM_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(long long), st));
        k_plan_chunks<<<grid, block, 0, st>>>(I, (const long long *)paths, start_lo, start_hi, chunk, G, cnt);
        XM_LAUNCH_CHECK();
        XM_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
        XM_HIP(hipStreamSynchronize(st));
        if (h_cnt[0] <= max_rows || chunk > h_cnt[1]) break;
        chunk *= 2;
    }
    int rc;
    int64_t n_starts = 0, n_units = 0, n_rows = 0, n_heavy = 0;
    // starts with work, heaviest unit first (stable: equal costs in start order)
    k_flag_pos<<<grid, block, 0, st>>>(I, G, flag);
    XM_LAUNCH_CHECK();
    if ((rc = xmap_exclusive_scan_i32_to_i64(stream, flag, (int64_t *)rank, I, &n_starts))) return rc;
    if (n_starts > 0) {
        k_plan_keys<<<grid, block, 0, st>>>(I, (const long long *)paths, G, rank, keys, vals);
        XM_LAUNCH_CHECK();
        if ((rc = radix_sort_pairs(st, keys, vals, keys_t, vals_t, n_starts, 44))) return rc;
        const dim3 gs_grid((unsigned)((n_starts + 255) / 256));
        k_plan_gather<<<gs_grid, block, 0, st>>>((int)n_starts, vals, G, gs, gh, ish);
        XM_LAUNCH_CHECK();
        if ((rc = xmap_exclusive_scan_i32_to_i64(stream, gs, (int64_t *)uoff, n_starts, &n_units))) return rc;
        if ((rc = xmap_exclusive_scan_i32_to_i64(stream, gh, (int64_t *)roff, n_starts, &n_rows))) return rc;
        if ((rc = xmap_exclusive_scan_i32_to_i64(stream, ish, (int64_t *)hrank, n_starts, &n_heavy))) return rc;
        if (n_units <= cap_units) {
            XM_ARG(unit_start && unit_c && unit_G && unit_row && heavy_unit0);
            k_plan_expand<<<gs_grid, block, 0, st>>>((int)n_starts, vals, gs, uoff, roff, hrank, unit_start, unit_c, unit_G, unit_row,
                                                     heavy_unit0);
            XM_LAUNCH_CHECK();
        }
    }
    XM_HIP(xm_free_async(ibuf, st));
    XM_HIP(xm_free_async(lbuf, st));
    XM_HIP(xm_free_async(keys, st));
    h_out[0] = n_units; h_out[1] = n_heavy; h_out[2] = n_rows; h_out[3] = h_cnt[1]; h_out[4] = chunk;
    if (n_units > cap_units) {
        set_error("unit arrays too small: need %lld, have %lld", (long long)n_units, (long long)cap_units);
        return XMAP_ERR_CAPACITY;
    }
    return XMAP_OK;
}

int xmap_end_order(void *stream, int32_t n_items, int top_k, int32_t n_nb, const int32_t *nb_list, const int32_t *kcnt,
                   const int32_t *kcol, int32_t n_ends, int32_t *urank, int32_t *uitem) {
    XM_ARG(urank && uitem && n_ends >= 0 && n_items < (1 << 21));
    if (n_ends == 0 || n_nb == 0) return XMAP_OK;
    XM_ARG(nb_list && kcnt && kcol);
    hipStream_t st = (hipStream_t)stream;
    int *home = nullptr, *vt = nullptr;
    unsigned long long *keys = nullptr;
    XM_HIP(xm_malloc_async((void **)&home, sizeof(int) * ((size_t)n_items + n_ends), st));
    vt = home + n_items;
    XM_HIP(xm_malloc_async((void **)&keys, sizeof(unsigned long long) * (size_t)n_ends * 2, st));
    XM_HIP(hipMemsetAsync(home, 0x7f, sizeof(int) * (size_t)n_items, st));      // 0x7f7f7f7f: after every column
    const long long n = (long long)n_nb * (top_k + 1);
    k_end_home<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n_nb, top_k, nb_list, kcnt, kcol, home);
    XM_LAUNCH_CHECK();
    const dim3 grid((unsigned)((n_ends + 255) / 256)), block(256);
    k_home_keys<<<grid, block, 0, st>>>(n_ends, uitem, home, keys);
    XM_LAUNCH_CHECK();
    int rc = radix_sort_pairs(st, keys, uitem, keys + n_ends, vt, n_ends, 52);
    if (rc) return rc;
    k_inverse<<<grid, block, 0, st>>>(n_ends, uitem, urank);
    XM_LAUNCH_CHECK();
    XM_HIP(xm_free_async(home, st));
    XM_HIP(xm_free_async(keys, st));
    return XMAP_OK;
}
}
