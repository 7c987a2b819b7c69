// api.hip -- the coarse, handle-based C ABI of the hot path: a host in any language drives the three
// pipelines with plain host buffers and never sees a device pointer, a scan or a work-unit plan.
//
//   xmap_ctx_create -> xmap_ctx_upload_ratings -> xmap_ctx_item_sim     (baseliner_calculate_sim_pipeline, assist.py:66-77)
//                                              -> xmap_ctx_extend       (extender_pipeline, assist.py:80-102; lazy lists)
//                                              -> xmap_ctx_generate     (generator_pipeline, assist.py:136-150)
//   xmap_ctx_*_download copy results into caller-allocated host buffers whose sizes the stage call reported.
//
// Everything below is orchestration of the kernels' own entry points (include/xmap_hip.h): buffer sizes, prefix sums,
// overflow retries and unit planning that xmap/engine/device.py does for the Python host.  No torch, no other library.
#include <stdlib.h>
#include <vector>

#include "common.h"

namespace xmap {

struct Pool {
    std::vector<void *> ptrs;
    void release() {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

}  // namespace xmap

using namespace xmap;

struct xmap_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    Pool p_ratings, p_sim, p_ext, p_gen, p_rows;
    // ratings
    xmap_ratings R;
    bool have_ratings = false;
    // stage A
    bool have_sim = false;
    xmap_sim S;
    double *u_avg = nullptr, *u_norm = nullptr, *info = nullptr;
    int64_t n_kept = 0, n_eval = 0, n_contrib = 0;
    int64_t half_contrib = 0;       // sum over the users of d (d - 1) / 2: sizes the pair buffers (known from user_ptr)
    // stage B
    bool have_ext = false;
    int top_k = 0;
    xmap_ext_tables T;
    xmap_path_units Un;
    int32_t *n_cand = nullptr, *top_end = nullptr;
    double *top_val = nullptr;
    int64_t n_out = 0, n_paths = 0;
    // accumulator rows (kernels return them zeroed: kept across passes)
    double *acc = nullptr, *hacc = nullptr;
    int32_t *touched = nullptr, *htouched = nullptr;
    int64_t acc_slots = 0, acc_len = 0, hacc_rows = 0, hacc_len = 0;
    int32_t n_slots = 0;
    int32_t fast_div = 0;
    // stage C
    bool have_gen = false;
    int32_t *g_user = nullptr, *g_item = nullptr;
    double *g_rating = nullptr;
    int64_t *g_time = nullptr;
    int64_t n_rows = 0, n_target_rows = 0;
};

namespace xmap {

template <typename T>
static int dalloc(Pool &pool, T **out, size_t n, hipStream_t st, bool zero = false) {
    void *p = nullptr;
    const size_t bytes = sizeof(T) * (n ? n : 1);
    XM_HIP(hipMalloc(&p, bytes));
    pool.ptrs.push_back(p);
    if (zero) XM_HIP(hipMemsetAsync(p, 0, bytes, st));
    *out = (T *)p;
    return XMAP_OK;
}
#define XM_TRY(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#define XM_ALLOC(pool, ptr, n) XM_TRY(dalloc(pool, &(ptr), (size_t)(n), c->st))
#define XM_ALLOCZ(pool, ptr, n) XM_TRY(dalloc(pool, &(ptr), (size_t)(n), c->st, true))

template <typename T>
static int h2d(Pool &pool, T **out, const T *host, size_t n, hipStream_t st) {
    XM_TRY(dalloc(pool, out, n, st));
    if (n) XM_HIP(hipMemcpyAsync(*out, host, sizeof(T) * n, hipMemcpyHostToDevice, st));
    return XMAP_OK;
}
template <typename T>
static int d2h(T *host, const T *dev, size_t n, hipStream_t st) {
    if (n) XM_HIP(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, st));
    return XMAP_OK;
}

// one reverse adjacency: count -> scan -> fill
static int reverse_list(xmap_ctx *c, int mode, const uint8_t *bb, const int64_t *attach_ptr, const void *thr, int32_t *long_rows,
                        uint8_t *eflag, int32_t *counted /*the list's counts where a fused pass left them, or NULL*/,
                        const int64_t **rptr, const int32_t **ridx, const double **rval, const uint8_t **rflag) {
    const int I = c->R.n_items, k = c->top_k;
    int32_t *rcnt = counted;
    int64_t *ptr;
    XM_ALLOCZ(c->p_ext, ptr, I + 1);
    if (!counted) {
        XM_ALLOCZ(c->p_ext, rcnt, I);
        XM_TRY(xmap_reverse_count(c->st, &c->S, mode, k, bb, c->T.cls, c->T.kcnt, c->T.kcol, c->T.kval, c->R.suffix_cls,
                                  c->R.contains_mask, c->R.flags, attach_ptr, thr, long_rows, eflag, rcnt, 0, I));
    }
    int64_t n = 0;
    XM_TRY(xmap_exclusive_scan_i32_to_i64(c->st, rcnt, ptr, I, &n));
    int32_t *idx;
    double *val;
    uint8_t *flag;
    XM_ALLOC(c->p_ext, idx, n);
    XM_ALLOC(c->p_ext, val, 3 * (size_t)(n ? n : 1));
    XM_ALLOCZ(c->p_ext, flag, n);
    XM_TRY(xmap_reverse_fill(c->st, &c->S, mode, k, bb, c->T.cls, c->T.kcnt, c->T.kcol, c->T.kval, c->R.suffix_cls,
                             c->R.contains_mask, c->R.flags, attach_ptr, thr, long_rows, eflag, ptr, idx, val, flag, 0, I));
    *rptr = ptr; *ridx = idx; *rval = val; *rflag = flag;
    return XMAP_OK;
}

// accumulator rows of the enumeration: zero-filled once, reused while large enough
static int ensure_rows(xmap_ctx *c, int64_t slots, int64_t len, int64_t hrows) {
    if (c->acc == nullptr || c->acc_slots < slots || c->acc_len != len || c->hacc_rows < hrows) {
        c->p_rows.release();
        c->acc = nullptr; c->hacc = nullptr;
        XM_ALLOCZ(c->p_rows, c->acc, (size_t)slots * len * 4);
        XM_ALLOC(c->p_rows, c->touched, (size_t)slots * len);
        if (hrows) {
            XM_ALLOCZ(c->p_rows, c->hacc, (size_t)hrows * len * 4);
            XM_ALLOC(c->p_rows, c->htouched, (size_t)hrows * len);
        }
        c->acc_slots = slots; c->acc_len = len; c->hacc_rows = hrows;
    }
    return XMAP_OK;
}

static int run_enumeration(xmap_ctx *c, int64_t xs_cap, int64_t *xs_off, int32_t *xs_end, double *xs_val) {
    const int I = c->R.n_items;
    xmap_path_rows Rw;
    Rw.n_slots = c->n_slots; Rw.acc = c->acc; Rw.touched = c->touched; Rw.hacc = c->hacc; Rw.htouched = c->htouched;
    xmap_path_out O;
    O.n_cand = c->n_cand; O.top_end = c->top_end; O.top_val = c->top_val;
    O.xs_cap = xs_cap; O.xs_off = xs_off; O.xs_end = xs_end; O.xs_val = xs_val;
    XM_HIP(hipMemsetAsync(c->n_cand, 0, sizeof(int32_t) * (size_t)I, c->st));
    XM_HIP(hipMemsetAsync(c->top_end, 0xff, sizeof(int32_t) * (size_t)I * XMAP_TOPC, c->st));
    XM_HIP(hipMemsetAsync(c->top_val, 0, sizeof(double) * (size_t)I * XMAP_TOPC, c->st));
    int64_t *d_cnt;
    XM_ALLOCZ(c->p_ext, d_cnt, 8);
    int64_t h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc;
    if (c->T.n_nb > 0) {
        rc = xmap_extend_cols(c->st, &c->T, &c->Un, &Rw, &O, c->fast_div, d_cnt, h_cnt);
    } else {    // nothing joint: the per-path kernel over item-indexed rows (the same row buffers, U = I)
        rc = xmap_extend_paths(c->st, I, c->top_k, c->T.cls, c->T.kcnt, c->T.kcol, c->T.kval, c->T.flags, c->T.att_ptr, c->T.att_idx,
                               c->T.att_val, c->T.src_ptr, c->T.src_idx, c->T.src_val, c->T.src_flag, c->T.rnn_ptr, c->T.rnn_idx,
                               c->T.rnn_val, c->Un.n_units, c->Un.unit_start, c->Un.unit_c, c->Un.unit_G, c->Un.unit_row,
                               c->Un.unit_nt, c->Un.n_heavy, c->Un.heavy_unit0, c->n_slots, c->acc, c->touched, c->hacc, c->htouched,
                               c->n_cand, c->top_end, c->top_val, xs_cap, xs_off, xs_end, xs_val, d_cnt, h_cnt);
    }
    if (rc && rc != XMAP_ERR_CAPACITY) {     // a failed pass may leave partial sums in the rows
        c->p_rows.release();
        c->acc = nullptr;
    }
    c->n_out = h_cnt[0]; c->n_paths = h_cnt[1];
    return rc;
}

}  // namespace xmap

extern "C" {

int xmap_ctx_create(int device, xmap_ctx **out) {
    XM_ARG(out);
    *out = nullptr;
    XM_HIP(hipSetDevice(device));
    xmap_ctx *c = new xmap_ctx();
    c->device = device;
    memset(&c->R, 0, sizeof(c->R)); memset(&c->S, 0, sizeof(c->S)); memset(&c->T, 0, sizeof(c->T)); memset(&c->Un, 0, sizeof(c->Un));
    hipError_t e = hipStreamCreate(&c->st);
    if (e != hipSuccess) { delete c; set_error("hipStreamCreate -> %s", hipGetErrorString(e)); return XMAP_ERR_HIP; }
    *out = c;
    return XMAP_OK;
}

void xmap_ctx_destroy(xmap_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    c->p_gen.release(); c->p_ext.release(); c->p_sim.release(); c->p_rows.release(); c->p_ratings.release();
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

int xmap_ctx_upload_ratings(xmap_ctx *c, int64_t n_users, int32_t n_items, const int64_t *user_ptr, const int32_t *item,
                            const float *rating, const int64_t *time, const int32_t *prefix_cls, const int32_t *suffix_cls,
                            const uint32_t *contains_mask, const uint8_t *flags) {
    XM_ARG(c && user_ptr && prefix_cls && suffix_cls && contains_mask && flags && n_users >= 0 && n_items >= 0);
    XM_HIP(hipSetDevice(c->device));
    c->p_gen.release(); c->p_ext.release(); c->p_sim.release(); c->p_ratings.release();
    c->have_sim = c->have_ext = c->have_gen = false;
    const int64_t nnz = user_ptr[n_users];
    XM_ARG(nnz >= 0 && nnz < 2147483647ll && (nnz == 0 || (item && rating && time)));
    for (int64_t e = 0; e < nnz; e++) XM_ARG(item[e] >= 0 && item[e] < n_items);
    xmap_ratings &R = c->R;
    memset(&R, 0, sizeof(R));
    R.n_users = n_users; R.n_items = n_items; R.nnz = nnz;
    c->half_contrib = 0;
    for (int64_t u = 0; u < n_users; u++) { const int64_t d = user_ptr[u + 1] - user_ptr[u]; c->half_contrib += d * (d - 1) / 2; }
    int64_t *d_ptr, *d_time, *d_iptr;
    int32_t *d_item, *d_iuser, *d_pre, *d_suf;
    float *d_rating, *d_irating;
    uint32_t *d_mask;
    uint8_t *d_flags;
    XM_TRY(h2d(c->p_ratings, &d_ptr, user_ptr, (size_t)n_users + 1, c->st));
    XM_TRY(h2d(c->p_ratings, &d_item, item, (size_t)nnz, c->st));
    XM_TRY(h2d(c->p_ratings, &d_rating, rating, (size_t)nnz, c->st));
    XM_TRY(h2d(c->p_ratings, &d_time, time, (size_t)nnz, c->st));
    XM_TRY(h2d(c->p_ratings, &d_pre, prefix_cls, (size_t)n_items, c->st));
    XM_TRY(h2d(c->p_ratings, &d_suf, suffix_cls, (size_t)n_items, c->st));
    XM_TRY(h2d(c->p_ratings, &d_mask, contains_mask, (size_t)n_items, c->st));
    XM_TRY(h2d(c->p_ratings, &d_flags, flags, (size_t)n_items, c->st));
    XM_ALLOCZ(c->p_ratings, d_iptr, n_items + 1);
    XM_ALLOCZ(c->p_ratings, d_iuser, nnz);
    XM_ALLOCZ(c->p_ratings, d_irating, nnz);
    R.user_ptr = d_ptr; R.user_item = d_item; R.user_rating = d_rating; R.user_time = d_time;
    R.item_ptr = d_iptr; R.item_user = d_iuser; R.item_rating = d_irating;
    R.prefix_cls = d_pre; R.suffix_cls = d_suf; R.contains_mask = d_mask; R.flags = d_flags;
    XM_HIP(hipStreamSynchronize(c->st));
    c->have_ratings = true;
    return XMAP_OK;
}

int xmap_ctx_item_sim(xmap_ctx *c, int method, int cap, int64_t *n_kept, int64_t *n_evaluated) {
    XM_ARG(c && c->have_ratings && (method == XMAP_COSINE || method == XMAP_ADJUST_COSINE) && cap > 0);
    XM_HIP(hipSetDevice(c->device));
    c->p_gen.release(); c->p_ext.release(); c->p_sim.release();
    c->have_sim = c->have_ext = c->have_gen = false;
    xmap_ratings &R = c->R;
    const int I = R.n_items;
    const int64_t U = R.n_users, nnz = R.nnz;
    Pool tmp;                       // layout / plan / half COO: released at the end of the stage
    struct Guard { Pool &p; ~Guard() { p.release(); } } guard{tmp};
#define T_ALLOC(ptr, n) XM_TRY(dalloc(tmp, &(ptr), (size_t)(n), c->st))
#define T_ALLOCZ(ptr, n) XM_TRY(dalloc(tmp, &(ptr), (size_t)(n), c->st, true))
    // A2 / A3 + layout of the "tri" formulation: one transposition (xmap_sim3_layout); the CSC arrays stay unbuilt
    int32_t *cnt;
    T_ALLOC(cnt, I);
    double *norms;
    XM_ALLOC(c->p_sim, c->u_avg, U);
    XM_ALLOC(c->p_sim, c->u_norm, U);
    XM_ALLOCZ(c->p_sim, c->info, (size_t)I * 4);
    T_ALLOCZ(norms, (size_t)2 * I);
    int32_t *hist, *ctl, *hid, *hlist;
    int64_t *pre;
    uint64_t *ub_key, *ub, *rcrec, *Wp, *srec, *buf_a, *buf_b;
    const size_t n1 = (size_t)(nnz ? nnz : 1);
    T_ALLOC(hist, U + 2); T_ALLOC(pre, U + 3); T_ALLOC(ctl, 4); T_ALLOC(hid, I); T_ALLOCZ(hlist, 1024);
    T_ALLOC(ub_key, n1); T_ALLOC(ub, n1); T_ALLOC(rcrec, 2 * n1); T_ALLOC(Wp, I);
    T_ALLOC(srec, 2 * n1); T_ALLOC(buf_a, 2 * n1); T_ALLOC(buf_b, 2 * n1);
    const int ch_min = 2048;
    XM_TRY(xmap_sim3_layout(c->st, &R, (int64_t *)R.item_ptr, nullptr, ch_min, 1 | 2 | 4, 0, I, cnt, c->u_avg, c->u_norm, hist, pre, ctl,
                            hid, hlist, ub_key, ub, srec, buf_a, buf_b, rcrec, Wp, c->info, norms, nullptr));
    const int64_t half_contrib = c->half_contrib;
    int32_t *Q, *Cc, *Qcat, *uq_item = nullptr, *uq_q = nullptr, *uc_item = nullptr, *uc_c = nullptr;
    uint8_t *small;
    int64_t *uq_ptr, *uc_ptr;
    T_ALLOCZ(Q, I); T_ALLOCZ(Cc, I); T_ALLOCZ(small, I); T_ALLOC(Qcat, (size_t)5 * (I ? I : 1)); T_ALLOCZ(uq_ptr, (size_t)5 * I + 1);
    T_ALLOCZ(uc_ptr, I + 1);
    int slot_target = 768;          // (75 % load of the 1024-slot tables; device.py: SLOT_TARGET / CH_MIN)
    double coo_slack = 1.0;
    int64_t hc[10];
    int64_t n_light = 0, n_hu = 0;
    int n_heavy = 0;
    auto plan = [&](int target) -> int {
        const int64_t cap_light = half_contrib / target + I + 1, cap_heavy = nnz / ch_min + 1025;
        T_ALLOC(uq_item, cap_light); T_ALLOC(uq_q, 4 * cap_light); T_ALLOC(uc_item, cap_heavy); T_ALLOC(uc_c, cap_heavy);
        XM_TRY(xmap_sim3_plan(c->st, &R, target, pre, hid, ctl, Q, Cc, small, Wp, Qcat, uq_ptr, uc_ptr, 0, uq_item, uq_q, uc_item, uc_c,
                              cap_light, cap_heavy, hc));
        n_light = hc[0]; n_hu = hc[1]; n_heavy = (int)hc[9];
        return XMAP_OK;
    };
    XM_TRY(plan(slot_target));
    int32_t *coo_i = nullptr, *coo_j = nullptr, *coo_mutu = nullptr, *coo_nij = nullptr, *own = nullptr, *mir = nullptr;
    double *coo_sim = nullptr;
    int64_t *d_shards = nullptr;
    int64_t cap_coo = 0, n = 0, n_unordered = 0;
    for (;;) {
        cap_coo = ((int64_t)((double)(half_contrib > 0 ? half_contrib : 1) * coo_slack) / 4096 + 1100) * 4096;
        double *hp_hi, *hp_lo;
        int32_t *hp_cnt, *hp_mut, *rowcnt_h;
        int64_t *d_cnt;
        T_ALLOC(coo_i, cap_coo); T_ALLOC(coo_j, cap_coo); T_ALLOC(coo_sim, cap_coo); T_ALLOC(coo_mutu, cap_coo); T_ALLOC(coo_nij, cap_coo);
        T_ALLOC(own, I); T_ALLOC(mir, I);
        const size_t hp = (size_t)(n_hu ? n_hu : 1) * 1024;
        T_ALLOC(hp_hi, hp); T_ALLOC(hp_lo, hp); T_ALLOC(hp_cnt, hp); T_ALLOC(hp_mut, hp);
        T_ALLOCZ(d_cnt, 6); T_ALLOC(d_shards, 2 * 4096); T_ALLOC(rowcnt_h, 64 * 1024);
        // all phases in one call: the heavy rows run on a side stream next to the class launches of the light rows; own and
        // mirrored row counts apart (mir), kept / evaluated pairs summed on the device (phase 64)
        XM_TRY(xmap_sim2_pairs(c->st, &R, method, cap, c->u_avg, norms, rcrec, ub, Q, small, uq_item, uq_q, hc + 2, 0, n_light, hid, hlist,
                               ctl, Cc, uc_ptr, uc_item, uc_c, (int32_t)n_hu, n_heavy, 8 | 1 | 2 | 4 | 16 | 64 | 128, hp_hi, hp_lo, hp_cnt,
                               hp_mut, cap_coo, coo_i, coo_j, coo_sim, coo_mutu, coo_nij, nullptr, own, rowcnt_h, d_shards, d_cnt, mir));
        int64_t h_cnt[6];
        XM_TRY(d2h(h_cnt, d_cnt, 6, c->st));
        XM_HIP(hipStreamSynchronize(c->st));
        if (h_cnt[2]) {                 // an LDS pair table overflowed: smaller partitions
            if (slot_target <= 32) { set_error("pair-table overflow"); return XMAP_ERR_OVERFLOW; }
            slot_target /= 2;
            XM_TRY(plan(slot_target));
            continue;
        }
        if (h_cnt[3]) {                 // a half-COO shard overflowed: more slack
            if (coo_slack > 64) { set_error("half-COO overflow"); return XMAP_ERR_CAPACITY; }
            coo_slack *= 2;
            continue;
        }
        n = h_cnt[4]; n_unordered = h_cnt[5];
        break;
    }
    // mirror the half COO into the CSR (row = [own | mirrored], tile sort by the heavier item)
    int64_t *row_ptr, *mptr;
    XM_ALLOCZ(c->p_sim, row_ptr, I + 1);
    T_ALLOCZ(mptr, I + 1);
    const int64_t kept = 2 * n;
    int32_t *col, *mutu, *nij, *fill, *tot;
    double *sim;
    uint64_t *mir_a, *mir_b;
    XM_ALLOC(c->p_sim, col, kept); XM_ALLOC(c->p_sim, sim, kept); XM_ALLOC(c->p_sim, mutu, kept); XM_ALLOC(c->p_sim, nij, kept);
    T_ALLOC(fill, I); T_ALLOC(tot, I); T_ALLOC(mir_a, (size_t)3 * (n ? n : 1)); T_ALLOC(mir_b, (size_t)3 * (n ? n : 1));
    XM_TRY(xmap_sim3_mircount(c->st, I, cap_coo, coo_i, coo_j, d_shards, n, 0, mir_a, mir));
    XM_TRY(xmap_sim3_mirror(c->st, I, cap_coo, coo_i, coo_j, coo_sim, coo_mutu, coo_nij, d_shards, n, own, mir, tot, row_ptr, mptr, fill,
                            mir_a, mir_b, col, sim, mutu, nij, nullptr, nullptr, 0, I));
    XM_HIP(hipStreamSynchronize(c->st));
    c->S.n_items = I; c->S.row_ptr = row_ptr; c->S.col = col; c->S.sim = sim; c->S.mutu = mutu; c->S.nij = nij; c->S.info = c->info;
    c->S.frac = nullptr;
    c->n_kept = kept; c->n_eval = 2 * n_unordered; c->n_contrib = 2 * half_contrib;
    c->have_sim = true;
    if (n_kept) *n_kept = kept;
    if (n_evaluated) *n_evaluated = c->n_eval;
    return XMAP_OK;
#undef T_ALLOC
#undef T_ALLOCZ
}

int xmap_ctx_sim_download(xmap_ctx *c, int64_t *row_ptr, int32_t *col, double *sim, int32_t *mutu, int32_t *nij, double *info,
                          double *user_avg) {
    XM_ARG(c && c->have_sim);
    XM_HIP(hipSetDevice(c->device));
    const int I = c->R.n_items;
    if (row_ptr) XM_TRY(d2h(row_ptr, c->S.row_ptr, (size_t)I + 1, c->st));
    if (col) XM_TRY(d2h(col, c->S.col, (size_t)c->n_kept, c->st));
    if (sim) XM_TRY(d2h(sim, c->S.sim, (size_t)c->n_kept, c->st));
    if (mutu) XM_TRY(d2h(mutu, c->S.mutu, (size_t)c->n_kept, c->st));
    if (nij) XM_TRY(d2h(nij, c->S.nij, (size_t)c->n_kept, c->st));
    if (info) XM_TRY(d2h(info, (const double *)c->info, (size_t)I * 4, c->st));
    if (user_avg) XM_TRY(d2h(user_avg, (const double *)c->u_avg, (size_t)c->R.n_users, c->st));
    XM_HIP(hipStreamSynchronize(c->st));
    return XMAP_OK;
}

int xmap_ctx_extend(xmap_ctx *c, int top_k, int64_t *n_out, int64_t *n_paths) {
    XM_ARG(c && c->have_sim && top_k >= 1);
    XM_HIP(hipSetDevice(c->device));
    c->p_gen.release(); c->p_ext.release();
    c->have_ext = c->have_gen = false;
    const int I = c->R.n_items, k = top_k;
    c->top_k = k;
    xmap_ext_tables &T = c->T;
    memset(&T, 0, sizeof(T));
    T.n_items = I; T.top_k = k; T.flags = c->R.flags;
    XM_ALLOCZ(c->p_ext, c->n_cand, I);
    XM_ALLOC(c->p_ext, c->top_end, (size_t)I * XMAP_TOPC);
    XM_ALLOCZ(c->p_ext, c->top_val, (size_t)I * XMAP_TOPC);
    c->n_out = c->n_paths = 0;
    if (I == 0) { c->have_ext = true; if (n_out) *n_out = 0; if (n_paths) *n_paths = 0; return XMAP_OK; }
    // B1-B4: bridge flags, classified top-k lists
    uint8_t *bb, *cls;
    int32_t *kcnt, *kcol;
    double *kval;
    XM_ALLOCZ(c->p_ext, bb, I); XM_ALLOC(c->p_ext, cls, I); XM_ALLOC(c->p_ext, kcnt, (size_t)I * 2);       // (xmap_knn_classify writes
    XM_ALLOC(c->p_ext, kcol, (size_t)I * 2 * k); XM_ALLOC(c->p_ext, kval, (size_t)I * 2 * k * 3);          //  every entry of its rows)
    XM_TRY(xmap_bridge_flags(c->st, &c->S, c->R.prefix_cls, bb));
    XM_TRY(xmap_knn_classify(c->st, &c->S, k, bb, c->R.suffix_cls, c->R.contains_mask, cls, kcnt, kcol, kval, 0, I));
    T.cls = cls; T.kcnt = kcnt; T.kcol = kcol; T.kval = kval;
    XM_TRY(xmap_edge_ranges(c->st, &c->S, &c->fast_div));
    // B5a/b: reverse adjacencies
    double *thr;
    int32_t *long_rows;
    XM_ALLOC(c->p_ext, thr, (size_t)I * 4);
    XM_ALLOC(c->p_ext, long_rows, I + 1);
    uint8_t *eflag;                         // one byte per entry of the matrix: what a count pass found, for its fill pass
    XM_ALLOC(c->p_ext, eflag, (size_t)(c->n_kept > 0 ? c->n_kept : 1));
    XM_TRY(xmap_knn_thresholds(c->st, I, k, kcnt, kcol, kval, thr));
    const uint8_t *dummy;
    // attach and rnn lists: ONE count pass over the matrix for both, then their fill passes; then the src lists (whose
    // predicate reads the attach offsets)
    int32_t *cnt_att, *cnt_rnn;
    XM_ALLOCZ(c->p_ext, cnt_att, I);
    XM_ALLOCZ(c->p_ext, cnt_rnn, I);
    XM_TRY(xmap_reverse_count_att_rnn(c->st, &c->S, k, bb, cls, kcnt, kcol, kval, c->R.suffix_cls, c->R.contains_mask, c->R.flags,
                                      thr, long_rows, eflag, cnt_att, cnt_rnn, 0, I));
    XM_TRY(reverse_list(c, 0, bb, nullptr, thr, long_rows, eflag, cnt_att, &T.att_ptr, &T.att_idx, &T.att_val, &dummy));
    XM_TRY(reverse_list(c, 2, bb, nullptr, thr, long_rows, eflag, cnt_rnn, &T.rnn_ptr, &T.rnn_idx, &T.rnn_val, &dummy));
    XM_TRY(reverse_list(c, 1, bb, T.att_ptr, thr, long_rows, eflag, nullptr, &T.src_ptr, &T.src_idx, &T.src_val, &T.src_flag));
    // exact per-start path counts -> work units
    int64_t *wtmp, *P;
    XM_ALLOCZ(c->p_ext, wtmp, (size_t)4 * I); XM_ALLOCZ(c->p_ext, P, I);
    XM_TRY(xmap_path_weights(c->st, I, k, cls, kcnt, kcol, c->R.flags, T.att_ptr, T.att_idx, T.src_ptr, T.src_idx, T.src_flag, T.rnn_ptr,
                             T.rnn_idx, wtmp, P));
    // middle lists of the joint paths (row-wise construction)
    int32_t *nb_list, *nb_id;
    int64_t n_nb = 0;
    XM_ALLOC(c->p_ext, nb_list, I); XM_ALLOC(c->p_ext, nb_id, I);
    XM_TRY(xmap_nb_index(c->st, I, cls, nb_list, nb_id, &n_nb));
    T.n_nb = (int32_t)n_nb; T.nb_list = nb_list; T.nb_id = nb_id;
    if (n_nb > 0) {
        int32_t *ng;
        int64_t *nrec, *dir_ptr, *rec_ptr, n_tiles = 0, n_records = 0;
        void *dir, *midX;
        XM_ALLOC(c->p_ext, ng, n_nb); XM_ALLOC(c->p_ext, nrec, n_nb); XM_ALLOCZ(c->p_ext, dir_ptr, n_nb + 1); XM_ALLOCZ(c->p_ext, rec_ptr, n_nb + 1);
        XM_TRY(xmap_mid_rows_count(c->st, I, k, cls, kcnt, kcol, kval, c->R.flags, T.att_ptr, T.att_idx, T.att_val, T.src_ptr, T.src_idx,
                                   T.src_val, T.src_flag, (int32_t)n_nb, nb_list, nb_id, ng, nrec));
        XM_TRY(xmap_exclusive_scan_i32_to_i64(c->st, ng, dir_ptr, n_nb, &n_tiles));
        XM_TRY(xmap_exclusive_scan_i64(c->st, nrec, rec_ptr, n_nb, &n_records));
        char *dir_c, *mid_c;
        XM_ALLOC(c->p_ext, dir_c, (size_t)(n_tiles ? n_tiles : 1) * 24);
        XM_ALLOC(c->p_ext, mid_c, (size_t)(n_records ? n_records : 1) * 64);
        dir = dir_c; midX = mid_c;
        XM_TRY(xmap_mid_rows_place(c->st, I, k, cls, kcnt, kcol, kval, c->R.flags, T.att_ptr, T.att_idx, T.att_val, T.src_ptr, T.src_idx,
                                   T.src_val, T.src_flag, (int32_t)n_nb, nb_list, nb_id, dir_ptr, rec_ptr, dir, midX));
        T.dir = dir; T.midX = midX; T.dir_ptr = dir_ptr;
    }
    // end universe (rows are indexed by end rank, in column order)
    int64_t len = I;
    if (n_nb > 0) {
        int32_t *mark, *urank, *uitem;
        int64_t *rank, n_ends = 0;
        XM_ALLOC(c->p_ext, mark, I); XM_ALLOC(c->p_ext, rank, I + 1); XM_ALLOC(c->p_ext, urank, I); XM_ALLOC(c->p_ext, uitem, I);
        XM_TRY(xmap_end_universe(c->st, &T, mark, rank, urank, uitem, &n_ends));
        XM_TRY(xmap_end_order(c->st, I, k, (int32_t)n_nb, nb_list, kcnt, kcol, (int32_t)n_ends, urank, uitem));
        T.n_ends = (int32_t)n_ends; T.urank = urank; T.uitem = uitem;
        len = n_ends > 0 ? n_ends : 1;
    }
    // work units; accumulator rows: one per resident wavefront (xmap_extend_cols_slots), capped by 24 GB
    int32_t resident = 0;
    XM_TRY(xmap_extend_cols_slots(&resident));
    int64_t n_slots = resident;
    const int64_t slot_cap = ((int64_t)24 << 30) / (36 * len);
    if (n_slots > slot_cap) n_slots = slot_cap > 4 ? slot_cap : 4;
    const int64_t max_rows = ((int64_t)16 << 30) / (36 * len) > 2 ? ((int64_t)16 << 30) / (36 * len) : 2;
    const int64_t cap_units = (int64_t)I + max_rows;
    int32_t *unit_start, *unit_c, *unit_G, *unit_row, *unit_nt, *heavy_unit0;
    XM_ALLOC(c->p_ext, unit_start, cap_units); XM_ALLOC(c->p_ext, unit_c, cap_units); XM_ALLOC(c->p_ext, unit_G, cap_units);
    XM_ALLOC(c->p_ext, unit_row, cap_units); XM_ALLOC(c->p_ext, heavy_unit0, I);
    int64_t hp[5];
    XM_TRY(xmap_path_plan(c->st, I, P, 0, I, 0, 8192, max_rows, cap_units, unit_start, unit_c, unit_G, unit_row, heavy_unit0, hp));
    XM_ALLOCZ(c->p_ext, unit_nt, hp[0]);
    xmap_path_units &Un = c->Un;
    Un.n_units = (int32_t)hp[0]; Un.unit_start = unit_start; Un.unit_c = unit_c; Un.unit_G = unit_G; Un.unit_row = unit_row;
    Un.unit_nt = unit_nt; Un.n_heavy = (int32_t)hp[1]; Un.heavy_unit0 = heavy_unit0;
    if (n_slots > hp[0]) n_slots = hp[0] > 4 ? hp[0] : 4;
    c->n_slots = (int32_t)n_slots;
    XM_TRY(ensure_rows(c, n_slots, len, hp[2]));
    XM_TRY(run_enumeration(c, 0, nullptr, nullptr, nullptr));
    c->have_ext = true;
    if (n_out) *n_out = c->n_out;
    if (n_paths) *n_paths = c->n_paths;
    return XMAP_OK;
}

int xmap_ctx_ext_download(xmap_ctx *c, int32_t *n_cand, int32_t *top_end, double *top_val) {
    XM_ARG(c && c->have_ext);
    XM_HIP(hipSetDevice(c->device));
    const size_t I = (size_t)c->R.n_items;
    if (n_cand) XM_TRY(d2h(n_cand, (const int32_t *)c->n_cand, I, c->st));
    if (top_end) XM_TRY(d2h(top_end, (const int32_t *)c->top_end, I * XMAP_TOPC, c->st));
    if (top_val) XM_TRY(d2h(top_val, (const double *)c->top_val, I * XMAP_TOPC, c->st));
    XM_HIP(hipStreamSynchronize(c->st));
    return XMAP_OK;
}

int xmap_ctx_ext_lists(xmap_ctx *c, int64_t *xs_off, int32_t *xs_end, double *xs_val) {
    XM_ARG(c && c->have_ext && xs_off && (c->n_out == 0 || (xs_end && xs_val)));
    XM_HIP(hipSetDevice(c->device));
    const int I = c->R.n_items;
    if (I == 0) return XMAP_OK;
    const int64_t cap = c->n_out > 0 ? c->n_out : 1;      // exact: the candidate counts of the first pass
    Pool tmp;
    struct Guard { Pool &p; ~Guard() { p.release(); } } guard{tmp};
    int64_t *d_off;
    int32_t *d_end;
    double *d_val;
    XM_TRY(dalloc(tmp, &d_off, (size_t)I, c->st, true));
    XM_TRY(dalloc(tmp, &d_end, (size_t)cap, c->st));
    XM_TRY(dalloc(tmp, &d_val, (size_t)cap, c->st));
    XM_TRY(run_enumeration(c, cap, d_off, d_end, d_val));
    XM_TRY(d2h(xs_off, (const int64_t *)d_off, (size_t)I, c->st));
    XM_TRY(d2h(xs_end, (const int32_t *)d_end, (size_t)c->n_out, c->st));
    XM_TRY(d2h(xs_val, (const double *)d_val, (size_t)c->n_out, c->st));
    XM_HIP(hipStreamSynchronize(c->st));
    return XMAP_OK;
}

int xmap_ctx_candidates(xmap_ctx *c, int32_t *n_top) {
    XM_ARG(c && c->have_ext && n_top);
    XM_HIP(hipSetDevice(c->device));
    const int I = c->R.n_items;
    if (I == 0) return XMAP_OK;
    Pool tmp;
    struct Guard { Pool &p; ~Guard() { p.release(); } } guard{tmp};
    int32_t *d_top, *d_choice, *d_map;
    XM_TRY(dalloc(tmp, &d_top, (size_t)I, c->st, true));
    XM_TRY(dalloc(tmp, &d_choice, (size_t)I, c->st, true));
    XM_TRY(dalloc(tmp, &d_map, (size_t)I, c->st, true));
    XM_TRY(xmap_select_map(c->st, I, 0, c->n_cand, c->top_end, nullptr, d_top, d_choice, d_map));
    XM_TRY(d2h(n_top, (const int32_t *)d_top, (size_t)I, c->st));
    XM_HIP(hipStreamSynchronize(c->st));
    return XMAP_OK;
}

int xmap_ctx_generate(xmap_ctx *c, int private_flag, const int32_t *picks, int32_t *choice, int64_t *n_rows, int64_t *n_target_rows) {
    XM_ARG(c && c->have_ext);
    XM_HIP(hipSetDevice(c->device));
    c->p_gen.release();
    c->have_gen = false;
    const int I = c->R.n_items;
    const int64_t U = c->R.n_users;
    int32_t *d_top, *d_choice, *d_map, *d_picks = nullptr;
    XM_ALLOCZ(c->p_gen, d_top, I); XM_ALLOCZ(c->p_gen, d_choice, I); XM_ALLOCZ(c->p_gen, d_map, I);
    if (picks) XM_TRY(h2d(c->p_gen, &d_picks, picks, (size_t)I, c->st));
    XM_TRY(xmap_select_map(c->st, I, private_flag ? 1 : 0, c->n_cand, c->top_end, d_picks, d_top, d_choice, d_map));
    if (choice) XM_TRY(d2h(choice, (const int32_t *)d_choice, (size_t)I, c->st));
    int32_t *cnt_t, *cnt_m;
    int64_t *off_t, *off_m, nt = 0, nm = 0;
    XM_ALLOCZ(c->p_gen, cnt_t, U); XM_ALLOCZ(c->p_gen, cnt_m, U); XM_ALLOCZ(c->p_gen, off_t, U + 1); XM_ALLOCZ(c->p_gen, off_m, U + 1);
    XM_TRY(xmap_alterego_count(c->st, &c->R, d_map, cnt_t, cnt_m, nullptr));
    XM_TRY(xmap_exclusive_scan_i32_to_i64(c->st, cnt_t, off_t, U, &nt));
    XM_TRY(xmap_exclusive_scan_i32_to_i64(c->st, cnt_m, off_m, U, &nm));
    const int64_t n = nt + nm;
    XM_ALLOC(c->p_gen, c->g_user, n); XM_ALLOC(c->p_gen, c->g_item, n); XM_ALLOC(c->p_gen, c->g_rating, n); XM_ALLOC(c->p_gen, c->g_time, n);
    XM_TRY(xmap_alterego_fill(c->st, &c->R, d_map, off_t, off_m, nt, c->g_user, c->g_item, c->g_rating, c->g_time));
    XM_HIP(hipStreamSynchronize(c->st));
    c->n_rows = n; c->n_target_rows = nt;
    c->have_gen = true;
    if (n_rows) *n_rows = n;
    if (n_target_rows) *n_target_rows = nt;
    return XMAP_OK;
}

int xmap_ctx_gen_download(xmap_ctx *c, int32_t *user, int32_t *item, double *rating, int64_t *time) {
    XM_ARG(c && c->have_gen);
    XM_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)c->n_rows;
    if (user) XM_TRY(d2h(user, (const int32_t *)c->g_user, n, c->st));
    if (item) XM_TRY(d2h(item, (const int32_t *)c->g_item, n, c->st));
    if (rating) XM_TRY(d2h(rating, (const double *)c->g_rating, n, c->st));
    if (time) XM_TRY(d2h(time, (const int64_t *)c->g_time, n, c->st));
    XM_HIP(hipStreamSynchronize(c->st));
    return XMAP_OK;
}
}
