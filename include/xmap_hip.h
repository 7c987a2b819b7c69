/*
 * xmap_hip.h -- C ABI of libxmap_hip.so, the MI355X (gfx950) engine behind X-MAP's hot path.
 *
 * The reference (LPD-EPFL-ML/X-MAP) is pure Python on Spark and has no FFI; the interface this
 * library replaces is the set of L2 "tool" methods that the three pipeline functions of
 * code/xmap/utils/assist.py call.  Each entry point below cites the
 * reference method(s) whose work it performs.  The host-side mirror of the Python API
 * (xmap.utils.assist / xmap.core.*) binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer into HBM unless its name starts with h_;
 *     buffers are allocated by the caller (the Python host uses torch tensors as containers);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - items are int32 indices in lexicographic order of the reference's id strings, users are
 *     int32 indices in trainRDD order; string predicates arrive as small per-item arrays
 *     (prefix_cls = class of iid[:2], suffix_cls = class of iid[-2:], contains_mask bit c =
 *     class-c suffix string occurs in iid, flags bit0 = "S:" in iid, bit1 = "T:" in iid);
 *   - return value 0 = ok, negative = error (xmap_last_error() gives the thread-local text);
 *     functions that hand a count back to the host synchronise `stream` before returning.
 */
#ifndef XMAP_HIP_H
#define XMAP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMAP_OK 0
#define XMAP_ERR_HIP -1       /* a HIP runtime call or kernel launch failed */
#define XMAP_ERR_ARG -2       /* bad argument */
#define XMAP_ERR_OVERFLOW -3  /* an on-chip accumulator table overflowed; retry with a smaller slot target */
#define XMAP_ERR_CAPACITY -4  /* caller-provided output buffer too small; needed size is reported */

#define XMAP_COSINE 0         /* BaselinerSim.method == "cosine"        (core/baselinerSim.py:213) */
#define XMAP_ADJUST_COSINE 1  /* BaselinerSim.method == "adjust_cosine" (core/baselinerSim.py:215) */

#define XMAP_TOPC 10          /* candidates kept per start item: generator.py:85 keeps 10, :109 keeps 4 */

/* Ratings resident in HBM: CSR by user (trainRDD order, profile order kept) + CSC by item
 * (raters in ascending user index = the order reduceByKey concatenates co-raters in). */
typedef struct xmap_ratings {
    int64_t n_users;
    int32_t n_items;
    int64_t nnz;
    const int64_t *user_ptr;    /* [n_users+1] */
    const int32_t *user_item;   /* [nnz] */
    const float *user_rating;   /* [nnz] */
    const int64_t *user_time;   /* [nnz] unix seconds (stage C only) */
    const int64_t *item_ptr;    /* [n_items+1] */
    const int32_t *item_user;   /* [nnz] */
    const float *item_rating;   /* [nnz] */
    const int32_t *prefix_cls;  /* [n_items] */
    const int32_t *suffix_cls;  /* [n_items] */
    const uint32_t *contains_mask; /* [n_items] */
    const uint8_t *flags;       /* [n_items] */
} xmap_ratings;

const char *xmap_last_error(void);
int xmap_version(void);

/* The library keeps its own temporaries (per thread, device and stream; recycled when a call ends, trimmed to 256 MiB when
 * idle).  xmap_trim hands everything the calling thread's idle arenas still hold back to the driver (synchronises the
 * device).  xmap_debug_arena / xmap_debug_arena_call are test hooks: the arena's live temporaries and reserved bytes, and a
 * call that takes two temporaries and leaves through an error path when fail != 0. */
int xmap_trim(void);
int xmap_debug_arena(void *stream, int64_t *live, int64_t *reserved);
int xmap_debug_arena_call(void *stream, int64_t bytes, int fail);

/* exclusive prefix sum of n int64 values; out[n] receives the total; *h_total (may be NULL) too (syncs). */
int xmap_exclusive_scan_i64(void *stream, const int64_t *in, int64_t *out, int64_t n, int64_t *h_total);
int xmap_exclusive_scan_i32_to_i64(void *stream, const int32_t *in, int64_t *out, int64_t n, int64_t *h_total);

/* ---- stage A: baseliner_calculate_sim_pipeline (utils/assist.py:66-77) ------------------- */

/* CSC (item -> raters) of the ratings from the CSR by user: the device-side counterpart of the flatMap + combineByKey
 * shuffle of get_universal_item_info (core/baselinerSim.py:65-82).  Rater order within an item is unspecified (every
 * consumer sums exactly).  Fills item_ptr / item_user / item_rating, which xmap_ratings then points at. */
int xmap_build_csc(void *stream, int64_t n_users, int32_t n_items, int64_t nnz, const int64_t *user_ptr,
                   const int32_t *user_item, const float *user_rating, int32_t *cnt /*[I] scratch*/,
                   int64_t *item_ptr /*[I+1]*/, int32_t *item_user /*[nnz]*/, float *item_rating /*[nnz]*/);

/* BaselinerSim.get_universal_user_info (core/baselinerSim.py:17-38): avg[u], norm2[u] (fp64). */
int xmap_user_stats(void *stream, const xmap_ratings *R, double *u_avg, double *u_norm2);

/* BaselinerSim.get_universal_item_info (core/baselinerSim.py:40-82): info[i] = (avg, norm2, adjnorm2, n).
 * Also emits the stage-A private copies of the index arrays with bit 31 = (rating >= item avg), which is
 * all retrieve_path_info (core/baselinerSim.py:97-113) needs per co-rating:
 *   ua_item[e] = user_item[e] | ge<<31,  ia_user[p] = item_user[p] | ge<<31.
 * Only the complete-rows formulation (xmap_sim_count / xmap_sim_fill) reads them: pass both NULL to skip them. */
int xmap_item_stats(void *stream, const xmap_ratings *R, const double *u_avg, double *info /*[I][4]*/,
                    double *norms /*[2][I] dense copies of norm2 / adjnorm2, may be NULL*/,
                    int32_t *ua_item /*[nnz] or NULL*/, int32_t *ia_user /*[nnz] or NULL*/,
                    int32_t item_lo, int32_t item_hi /*items [lo, hi): a rank's share when items are sharded (0, n_items: all)*/);

#ifdef XMAP_CROSSCHECK   /* test formulation: exported by libxmap_hip_xcheck.so only (csrc/Makefile), never by the product library */
/* Work decomposition for the pair kernel: unit = (item i, hash partition q of its partner space),
 * Q[i] = ceil(min(W_i, I-1) / slot_target), W_i = sum over raters of (profile length - 1).
 * Writes Q[I], W[I], unit_ptr[I+1] (exclusive scan of Q); *h_n_units, *h_contrib (= sum W_i = P). Syncs. */
int xmap_sim_plan(void *stream, const xmap_ratings *R, int32_t slot_target, int32_t *Q, int64_t *W, int64_t *unit_ptr,
                  int64_t *h_n_units, int64_t *h_contrib);
int xmap_sim_units(void *stream, int32_t n_items, const int32_t *Q, const int64_t *unit_ptr,
                   int32_t *unit_item /*[n_units]*/, int32_t *unit_q /*[n_units]*/);

/* BaselinerSim.calculate_item2item_sim (core/baselinerSim.py:176-216) restricted to the units
 * [unit_lo, unit_hi) (item shards for multi-GPU): produce_pairwise_items + reduceByKey + cosine /
 * adjusted-cosine + significance weighting + mutuality + zero filter.
 * Pass 1 (count): unit_cnt[u] = kept pairs of unit u; h_counters[0] += kept, [1] += evaluated (D),
 *                 [2] = overflow flag.  Pass 2 (fill) writes the kept pairs of unit u at
 *                 unit_off[u] + [0, unit_cnt[u]) as (col, sim fp64, mutu, n_ij).
 * sim, mutu are symmetric bit for bit; frac_mutu = mutu / (n_i + n_j - n_ij) is derived by consumers. */
int xmap_sim_count(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg,
                   const double *info, const int32_t *ua_item, const int32_t *ia_user, const int32_t *Q,
                   const int32_t *unit_item, const int32_t *unit_q, int64_t unit_lo, int64_t unit_hi,
                   int32_t *unit_cnt /*[n_units]*/, int64_t *d_counters /*[4] device*/, int64_t *h_counters /*[4]*/);
int xmap_sim_fill(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg,
                  const double *info, const int32_t *ua_item, const int32_t *ia_user, const int32_t *Q,
                  const int32_t *unit_item, const int32_t *unit_q, int64_t unit_lo, int64_t unit_hi,
                  const int64_t *unit_off /*[n_units+1]*/, int32_t *col, double *sim, int32_t *mutu, int32_t *nij);

/* row_ptr[i] = unit_off[unit_ptr[i]], i in [0, I]: CSR row pointers of the kept pairs (units of one
 * item are contiguous, so its partitions concatenate into its row). */
int xmap_sim_row_ptr(void *stream, int32_t n_items, const int64_t *unit_ptr, const int64_t *unit_off /*[n_units+1]*/,
                     int64_t *row_ptr /*[I+1]*/);
#endif /* XMAP_CROSSCHECK */

/* ---- stage A, second formulation (stage_a2.hip): each unordered pair is computed once, in the row of its lighter
 * item (weight = (rater count, index)), appended to a half COO and mirrored into the CSR.  Same results as
 * xmap_sim_count/fill (sums are exact, hence order-independent); about 40x fewer rater visits on skewed data.
 *   layout : per-user private profile copies sorted heaviest first (ub: item | flag, rating interleaved), one 16-byte
 *            rater record per CSC entry (profile offset, prefix length | flag, rating, user; no atomics, raters stay in
 *            ascending user order), the heavy set H = items with more than CH raters (|H| <= 1024; ctl[0] = CH >=
 *            ch_min, ctl[1] = |H|), pre[v] = #items with fewer than v raters.
 *   plan   : W+[i] = contributions of row i (sum of its raters' prefix lengths); Q[i] hash partitions for light rows,
 *            small[i] = LDS table class of the row (1: 128 slots, 3: 256, 2: 512, 0: 1024, 4: 1024 shared by 16 waves
 *            for light rows with >= 2048 raters), C[i] rater chunks for
 *            rows of H.  The light units are listed class-major (largest tables first): Qcat[rank][i] = Q[i] in the
 *            row's class rank, uq_ptr = exclusive scan over Qcat; h_counts = {light units, heavy units, first unit
 *            of class rank 0..4, light units} (cls_ptr of xmap_sim2_pairs = h_counts + 2).
 *   pairs  : phases bit 8 = reset counters/rowcnt, 1 = k_pair_heavy (chunk partials of the rows of H), 2 = k_pair_tri
 *            (light units [unit_lo, unit_hi), one launch per table class; 16 / 4 / 2 waves share a 1024 / 1024 / 512-slot table),
 *            4 = k_heavy_merge, 16 = (last; mircnt == NULL only) the mirrored row counts rowcnt[j]++ from the COO's partner
 *            column, 128 = with 8: do not mark the unused COO entries (a caller that reads the COO through the shard
 *            cursors only saves a 4 B x coo_cap fill); with 1 | 2 | 4 in one call the
 *            heavy rows (partials, then merge) run on a side stream next to the class launches of the light rows;
 *            kept pairs (i lighter, j heavier) ->
 *            half COO: coo_cap entries cut into 4096 shards with a cursor each (d_shards[0][s]; unused entries keep
 *            coo_i = -1), rowcnt[i]++ / rowcnt[j]++; d_shards[1][s] sums to the unordered pairs evaluated;
 *            d_counters[2] = table overflow, [3] = COO shard overflow.
 *            bits 16-23 / 8-15 of phases = m / r (m > 1): only the rows of H whose item index % m == r are computed
 *            (item-sharded ranks deal the heavy rows round-robin; partials and merge of a row stay on one rank).
 *   scatter: after an exclusive scan of rowcnt -> row_ptr, both directions of every valid COO entry (n_coo = coo_cap
 *            entries are scanned) into the CSR. */
int xmap_sim2_layout(void *stream, const xmap_ratings *R, const double *info, int32_t ch_min, int32_t *hist /*[U+2]*/,
                     int64_t *pre /*[U+3]*/, int32_t *ctl /*[4]*/, int32_t *hid /*[I]*/, int32_t *hlist /*[1024]*/,
                     uint64_t *ub_key /*[nnz] scratch*/, void *ub /*[nnz] x 8 B: item|flag, rating*/,
                     void *rc /*[nnz] x 16 B rater records in CSC order*/,
                     uint64_t *Wp /*[I] out: contributions per row (sum of its raters' prefix lengths)*/,
                     int32_t dups /*1: a profile may hold an item more than once (AlterEgo rows)*/, int32_t *h_ctl /*[2]*/);
int xmap_sim2_plan(void *stream, const xmap_ratings *R, int32_t slot_target, const void *rc, const int64_t *pre,
                   const int32_t *hid, const int32_t *ctl, int32_t *Q, int32_t *C, uint8_t *small /*[I]*/,
                   uint64_t *Wp /*[I] as left by xmap_sim2_layout*/, int32_t *Qcat /*[5 I]*/, int64_t *uq_ptr /*[5 I + 1]*/,
                   int64_t *uc_ptr /*[I + 1]*/, int32_t dups, int64_t *h_counts /*[8], host*/);
/* light unit u: uq_item[u] and the 16-byte record uq_q[4 u ..] = (hash partition, first rater, end of raters, partitions of
 * the row): what a pair kernel needs to start, in one round trip */
int xmap_sim2_units(void *stream, int32_t n_items, const int64_t *item_ptr, const int32_t *Qcat, const int64_t *uq_ptr,
                    int32_t *uq_item, int32_t *uq_q /*[4 light units]*/, const int32_t *C, const int64_t *uc_ptr, int32_t *uc_item,
                    int32_t *uc_c);
int xmap_sim2_pairs(void *stream, const xmap_ratings *R, int method, int cap, const double *u_avg, const double *norms,
                    const void *rc, const void *ub, const int32_t *Q,
                    const uint8_t *small, const int32_t *uq_item, const int32_t *uq_q, const int64_t *cls_ptr /*[6], host*/,
                    int64_t unit_lo, int64_t unit_hi, const int32_t *hid,
                    const int32_t *hlist, const int32_t *ctl, const int32_t *C, const int64_t *uc_ptr,
                    const int32_t *uc_item, const int32_t *uc_c, int32_t n_heavy_units, int32_t n_heavy, int phases,
                    double *hp_hi, double *hp_lo, int32_t *hp_cnt, int32_t *hp_mut, int64_t coo_cap, int32_t *coo_i,
                    int32_t *coo_j, double *coo_sim, int32_t *coo_mutu, int32_t *coo_nij,
                    double *coo_ls /*NULL, or the RecommenderSim variant (below)*/, int32_t *rowcnt,
                    int32_t *rowcnt_h /*[64][1024] scratch*/, int64_t *d_shards /*[2][4096]*/,
                    int64_t *d_counters /*[4]; [6] with phases bit 64: [4] / [5] = kept / evaluated unordered pairs, the sums
                                          of d_shards[0] / [1]*/,
                    int32_t *mircnt /*[I] or NULL.  NULL: rowcnt[i]++ / rowcnt[j]++ as described above.  Else rowcnt counts
                                      the pairs a row computed itself only and mircnt is cleared: the caller gets the mirrored
                                      counts from xmap_sim3_mircount (the pair kernels issue no atomic per kept pair)*/);
int xmap_sim2_scatter(void *stream, int32_t n_items, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j,
                      const double *coo_sim, const int32_t *coo_mutu, const int32_t *coo_nij, const double *coo_ls /*or NULL*/,
                      const int64_t *row_ptr, int32_t *fill /*[I] scratch*/, const int32_t *hid, const int32_t *hlist,
                      int32_t *col, double *sim, int32_t *mutu, int32_t *nij, double *ls /*or NULL*/);
/* ---- round 3: one transposition per pass -----------------------------------------------------------------------------
 * xmap_sim3_layout replaces xmap_build_csc + xmap_user_stats + xmap_item_stats + xmap_sim2_layout for the "tri" formulation:
 *   k_count3          raters per item in one pass over the CSR (LDS-cached atomics) -> item_ptr (exclusive scan)
 *   k_user_stats      u_avg / u_norm (get_universal_user_info, core/baselinerSim.py:17-38)        [float ratings only]
 *   k_hist .. k_mark_heavy   as xmap_sim2_layout
 *   k_sort_profiles3  profiles sorted heaviest first: ub (8 B per entry; 16 B with rating64) and one sort record per entry
 *                     {item, position, rating, user} (16 B; 24 B with rating64)
 *   tile sort         (csrc/tilesort.h) sort records -> rater records rc in item order (any order inside an item), W+ summed
 *   k_item_stats3 ..  get_universal_item_info (:40-82) from the rater records -> info, norms; items with more than 512
 *                     raters in chunks of 2048 on a wave each, merged in chunk order
 *   flags             `rating >= item average` (retrieve_path_info, :97-113) into the rater records and the profile copy
 * The CSC arrays (R->item_user / item_rating) are neither read nor written; R->item_ptr (= item_ptr) is written.
 * rating64 != NULL: the ratings are fp64 (RecommenderSim over AlterEgo means, core/recommenderSim.py:64-133; R->user_rating is
 * ignored), the user averages are zero by construction (u_avg must be zero-filled, u_norm may be NULL), there is no
 * mutuality, and ub / rc use the 16-byte wide forms xmap_sim2_pairs reads when coo_ls != NULL without phases bit 32.
 * phases: 1 = everything up to the rater records and W+; 2 = item statistics of the items [stats_lo, stats_hi) (and the flags
 * of THEIR rater records); 4 = flags of the profile copy, 8 = flags of all rater records -- both from the complete info.
 * One GPU: 1 | 2 | 4 with all items.  Item-sharded ranks: 1 | 2 with the rank's share, an all-gather of info / norms (the
 * all-gather of per-item norms), then 4 | 8.
 * h_ctl (host, [2], may be NULL) = {CH, |H|}; synchronises if given. */
int xmap_sim3_layout(void *stream, const xmap_ratings *R, int64_t *item_ptr /*[I+1] = R->item_ptr*/,
                     const double *rating64 /*[nnz] or NULL*/, int32_t ch_min, int32_t phases, int32_t stats_lo, int32_t stats_hi,
                     int32_t *cnt /*[I] scratch*/,
                     double *u_avg /*[U]*/, double *u_norm /*[U] or NULL with rating64*/, int32_t *hist /*[U+2]*/,
                     int64_t *pre /*[U+3]*/, int32_t *ctl /*[4]*/, int32_t *hid /*[I]*/, int32_t *hlist /*[1024]*/,
                     uint64_t *ub_key /*[nnz] scratch*/, void *ub /*[nnz] x 8 B (16 B)*/, void *srec /*[nnz] x 16 B (24 B) scratch*/,
                     void *bufA /*as srec, scratch*/, void *bufB /*as srec, scratch*/, void *rc /*[nnz] x 16 B*/,
                     uint64_t *Wp /*[I]*/, double *info /*[I][4]*/, double *norms /*[2][I]*/, int32_t *h_ctl /*[2], host*/);
/* xmap_sim2_plan + xmap_sim2_units with one synchronisation instead of four: the caller sizes the unit arrays from bounds it
 * knows without asking the device (light units <= contributions / slot_target + n_items with contributions = sum over the
 * users of d (d - 1) / 2; heavy units <= nnz / ch_min + 1024).  h_out [10], host = {light units, heavy units, first unit of
 * table class rank 0..4, light units, CH, |H|} (h_out + 2 is the cls_ptr of xmap_sim2_pairs). */
int xmap_sim3_plan(void *stream, const xmap_ratings *R, int32_t slot_target, const int64_t *pre, const int32_t *hid,
                   const int32_t *ctl, int32_t *Q, int32_t *C, uint8_t *small, uint64_t *Wp, int32_t *Qcat /*[5 I]*/,
                   int64_t *uq_ptr /*[5 I + 1]*/, int64_t *uc_ptr /*[I + 1]*/, int32_t dups, int32_t *uq_item, int32_t *uq_q /*[4 cap_light]*/,
                   int32_t *uc_item, int32_t *uc_c, int64_t cap_light, int64_t cap_heavy, int64_t *h_out /*[10], host*/);
/* mir[j] = entries of a half COO whose second index is j, i.e. the mirrored entries row j gets (skip_self: an entry pairing a
 * row with itself has none).  Three coalesced passes over the partner column (bucket histogram, scatter, LDS windows) instead
 * of the device atomic per kept pair the pair kernels used to issue: those 2.65e7 atomics per pass were what the class
 * launches waited for.  scratch: n_pairs ints.  d_shards as in xmap_sim3_mirror. */
int xmap_sim3_mircount(void *stream, int32_t n_items, int64_t coo_cap, const int32_t *coo_i, const int32_t *coo_j,
                       const int64_t *d_shards /*or NULL*/, int64_t n_pairs, int32_t skip_self, void *scratch, int32_t *mir /*[I] out*/);
/* The mirror of round 3.  own[i] = pairs row i computed (xmap_sim2_pairs' rowcnt), mir[j] = pairs computed in lighter rows
 * (xmap_sim3_mircount).  Row i of the CSR = [own | mirrored]: row_ptr = exclusive scan of own + mir; the own halves are written
 * in runs straight from the COO, the mirrored halves go through the tile sort keyed by the heavier item (positions mptr =
 * exclusive scan of mir).  n_pairs = valid COO entries.  coo_aux / aux (both or neither; RecommenderSim): a sixth column --
 * the pair's local sensitivity -- travels along (32-byte records), and a row may pair with itself: such an entry is an own
 * entry only (xmap_sim2_pairs counted it that way), so the CSR has row_ptr[I] <= 2 n_pairs entries. */
int xmap_sim3_mirror(void *stream, int32_t n_items, int64_t coo_cap, const int32_t *coo_i, const int32_t *coo_j,
                     const double *coo_sim, const int32_t *coo_mutu, const int32_t *coo_nij,
                     const int64_t *d_shards /*[4096] fill of the COO's shards as left by xmap_sim2_pairs (coo_cap / 4096
                                               slots each), or NULL: the COO is one range of n_pairs records*/,
                     int64_t n_pairs, const int32_t *own /*[I]*/, const int32_t *mir /*[I]*/, int32_t *tot /*[I] scratch*/,
                     int64_t *row_ptr /*[I+1] out*/, int64_t *mptr /*[I+1] scratch*/, int32_t *fill /*[I] scratch*/,
                     void *bufA /*[n_pairs] x 24 B (32 B with aux) scratch*/, void *bufB /*as bufA*/, int32_t *col,
                     double *sim, int32_t *mutu, int32_t *nij, const double *coo_aux /*or NULL*/, double *aux /*or NULL*/,
                     int32_t row_lo, int32_t row_hi /*the rows to build: 0, n_items = all.  An item-sharded rank builds its share of
                     the rows from the complete COO: own / mir must be zero outside [row_lo, row_hi), entries of other rows are skipped*/);

/* User-sharded input (BASELINE configs[2]: "reduce-scatter of cross-shard partial similarities"): a rank
 * holds the complete profiles of a share of the USERS.  Per item its share of get_universal_item_info's sums
 * (core/baselinerSim.py:56-82) is xmap_item_partials -> [I][5] = (sum r, sum r^2, sum (r - avg_u)^2 as an exact (value, error)
 * pair, raters); the shares of all ranks, gathered as [n_parts][I][5], are added up and finished by xmap_item_merge
 * (rank order; exact for the adjusted norm and for integer-valued ratings).  xmap_sim2_pairs with phases bit 32 ("raw": no
 * heavy set, coo_ls != NULL) then emits, for every pair two of the rank's users co-rated, the partial sums of
 * calculate_cosine_sim / calculate_adjusted_cosine_sim (:115-174) and retrieve_path_info (:97-113) unfinished and unfiltered:
 * coo_sim / coo_ls = the dot product as an exact (value, error) pair, coo_nij, coo_mutu.  xmap_sim2_pack_partials turns
 * them into 32-byte records (key = lower index << 32 | higher index, value, error, n_ij | mutu << 32; *h_count of them),
 * xmap_sim2_sort_partials groups them by the rank that owns the lower item (stable radix sort), and -- after the exchange,
 * sorted once more by pair key (stable: the shares of a pair stay in rank order), so that every rank holds ALL records of the
 * pairs it owns -- xmap_sim2_merge_partials adds
 * the shares of a pair up (the dot product exactly), applies cosine, significance weighting and the zero filter
 * (:84-95,:198,:207) with the merged item norms and appends the kept pairs (i < j) to a half COO + row counts, which
 * xmap_sim2_scatter mirrors as usual.  h_counts = {kept, evaluated} unordered pairs. */
/* The exchange of a sharded step's kept pairs before stage B (the reference broadcasts its knn tables, utils/assist.py:88-101):
 * valid entries of a half COO (coo_i >= 0) -> 24-byte records (i | j << 32, sim bits, mutu | n_ij << 32), *h_count of them,
 * in any order; and back into COO columns after the all-gather. */
int xmap_sim2_pack_pairs(void *stream, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j, const double *coo_sim,
                         const int32_t *coo_mutu, const int32_t *coo_nij, int64_t *rec /*[n_coo][3]*/, int64_t *h_count);
int xmap_sim2_unpack_pairs(void *stream, int64_t n, const int64_t *rec /*[n][3]*/, int32_t *coo_i, int32_t *coo_j, double *coo_sim,
                           int32_t *coo_mutu, int32_t *coo_nij);
int xmap_item_partials(void *stream, const xmap_ratings *R, const double *u_avg, double *partial /*[I][5]*/);
int xmap_item_merge(void *stream, int32_t n_items, int32_t n_parts, const double *parts /*[n_parts][I][5]*/, double *info /*[I][4]*/,
                    double *norms /*[2][I]*/);
int xmap_sim2_pack_partials(void *stream, int64_t n_coo, const int32_t *coo_i, const int32_t *coo_j, const double *coo_hi,
                            const double *coo_lo, const int32_t *coo_mutu, const int32_t *coo_nij, int64_t *rec /*[n_coo][4]*/,
                            int64_t *h_count);
int xmap_sim2_sort_partials(void *stream, int64_t n, const int64_t *rec /*[n][4]*/, int64_t *rec_sorted /*[n][4]*/, int32_t n_items,
                            int32_t n_owners /*> 0: group by the rank owning the lower item (before the exchange); 0: by pair key*/);
int xmap_sim2_merge_partials(void *stream, int method, int cap, int32_t n_items, int64_t n, const int64_t *rec_sorted,
                             const double *norms /*[2][I]*/, int32_t *coo_i, int32_t *coo_j, double *coo_sim, int32_t *coo_mutu,
                             int32_t *coo_nij, int32_t *rowcnt /*[I]*/, int64_t *h_counts /*[2]*/);

/* RecommenderSim.calculate_sim (core/recommenderSim.py:65-133,188-195; both method names take the cosine branch, :190)
 * is the same pair machinery over the AlterEgo rows: call xmap_item_stats with u_avg = 0 (adjnorm is then the exact
 * norm), xmap_sim2_layout with dups = 1 and ch_min > n_users (no heavy set), xmap_sim2_plan with dups = 1, and
 * xmap_sim2_pairs with method XMAP_ADJUST_COSINE (exact double-double sums), u_avg = 0 and coo_ls != NULL: nothing is
 * filtered, an item held twice by a user pairs with itself (one entry, both orders counted), and coo_ls receives the
 * leave-one-out local sensitivity of every pair (second walk over the raters with the final inner product; NaN
 * propagates like np.max).  xmap_sim2_scatter then mirrors (col, sim, n_ij, ls); mutu is unused (0). */

/* ---- stage B: extender_pipeline (utils/assist.py:80-133) ---------------------------------- */

/* Similarity matrix of stage A, CSR by first item (get_item_sim, core/baselinerSim.py:218-233). */
typedef struct xmap_sim {
    int32_t n_items;
    const int64_t *row_ptr; /* [I+1] */
    const int32_t *col;
    const double *sim;
    const int32_t *mutu;
    const int32_t *nij;
    const double *info;     /* [I][4] item info of stage A (n_i is info[i][3]) */
    const double *frac;     /* optional [nnz] frac_mutu per pair; NULL = derive mutu / (n_i + n_j - n_ij) */
} xmap_sim;

/* build_sim_DF + "SELECT DISTINCT id1 ... WHERE label = 1" (core/baselinerSim.py:235-244,
 * utils/assist.py:82-87): bb[i] = 1 iff item i has a kept pair whose 2-char prefixes differ. */
int xmap_bridge_flags(void *stream, const xmap_sim *S, const int32_t *prefix_cls, uint8_t *bb);

/* ExtendSim.find_knn_items (core/extender.py:16-44) + extract_siminfo (utils/assist.py:105-133):
 * per item the two top-k lists by (|sim| desc, col asc): list 0 = BB_BB | NB_BB, list 1 = BB_NB | NB_NN.
 * cls[i] = 0 none, 1 bridge record, 2 non-bridge record. kval = (sim, mutu, frac_mutu) fp64.  Every entry of the rows
 * [row_lo, row_hi) is written (the unused tail of a list with zeros): the tables may come uninitialised. */
int xmap_knn_classify(void *stream, const xmap_sim *S, int top_k, const uint8_t *bb, const int32_t *suffix_cls,
                      const uint32_t *contains_mask, uint8_t *cls, int32_t *kcnt /*[I][2]*/,
                      int32_t *kcol /*[I][2][k]*/, double *kval /*[I][2][k][3]*/,
                      int32_t row_lo, int32_t row_hi /*rows [lo, hi): a rank's share (0, n_items: all)*/);

/* Reverse adjacencies of the knn tables, built deterministically in row order:
 *   mode 0 ATTACH: attach(b) = [x : x non-bridge record, b in NB_BB(x)]   (core/extender.py:48-59,171-173)
 *   mode 1 SRC   : src(t)    = [s : s bridge, "S:" in s, attach(s) != [], t in keys(knn_BB[s])], "T:" in t
 *                               (core/extender.py:61-70,174,176); rflag bit0 = (t,s) also joins TGT (:72-81,175-178)
 *   mode 2 RNN   : rnn(y)    = [x : x non-bridge record, y in NB_NN(x)]    (core/extender.py:142-169 longest_path)
 * count pass: rcnt[I]; fill pass: ridx / rval (sim, mutu, frac) / rflag at rptr[a] + ... */
/* thr[i][l] = {|sim| (double), column (int32), length (int32)} of the LAST entry of list l of item i (16 B each): the
 * membership tests of xmap_reverse_* then cost one gather into a 32 B x n_items table instead of three into the lists. */
int xmap_knn_thresholds(void *stream, int32_t n_items, int top_k, const int32_t *kcnt, const int32_t *kcol, const double *kval,
                        void *thr /*[n_items][2] x 16 B*/);
int xmap_reverse_count(void *stream, const xmap_sim *S, int mode, int top_k, const uint8_t *bb, const uint8_t *cls,
                       const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                       const uint32_t *contains_mask, const uint8_t *flags, const int64_t *attach_ptr,
                       const void *thr /*xmap_knn_thresholds or NULL*/,
                       int32_t *long_rows /*[I+1] scratch or NULL: written here, rows of > 4096 entries get 16 waves*/,
                       uint8_t *eflag /*one byte per entry of the rows [row_lo, row_hi) or NULL: written here (bit 0 = listed,
                       bit 1 = joint); xmap_reverse_fill given the same buffer reads it instead of testing every entry again*/,
                       int32_t *rcnt /*[I]*/, int32_t row_lo, int32_t row_hi /*the rows (= targets of the lists) of this
                       call: the list of a row is built from that row alone, so a rank's share of the rows gives a
                       contiguous share of the lists; [0, I) = all*/);
/* attach (mode 0) and rnn (mode 2) lists counted in ONE pass over the rows -- both ask the same non-bridge neighbours b of an
 * entry, about their two lists: one class gather and one read of the matrix instead of two.  eflag (required) then serves
 * xmap_reverse_fill of mode 0 and of mode 2 (bit 0 / bit 2). */
int xmap_reverse_count_att_rnn(void *stream, const xmap_sim *S, int top_k, const uint8_t *bb, const uint8_t *cls,
                               const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                               const uint32_t *contains_mask, const uint8_t *flags, const void *thr, int32_t *long_rows,
                               uint8_t *eflag, int32_t *rcnt_att /*[I]*/, int32_t *rcnt_rnn /*[I]*/, int32_t row_lo, int32_t row_hi);
int xmap_reverse_fill(void *stream, const xmap_sim *S, int mode, int top_k, const uint8_t *bb, const uint8_t *cls,
                      const int32_t *kcnt, const int32_t *kcol, const double *kval, const int32_t *suffix_cls,
                      const uint32_t *contains_mask, const uint8_t *flags, const int64_t *attach_ptr,
                      const void *thr /*xmap_knn_thresholds or NULL*/,
                      int32_t *long_rows /*as left by xmap_reverse_count, or NULL*/,
                      uint8_t *eflag /*as left by xmap_reverse_count of the same mode and rows, or NULL*/, const int64_t *rptr /*[I+1]*/,
                      int32_t *ridx, double *rval /*[n][3]*/, uint8_t *rflag, int32_t row_lo, int32_t row_hi);

/* Scheduling weights of the path enumeration: paths[start] = number of paths that start at `start`
 * (exact; tails(s) summed over src(t), heads over NB_BB / rnn).  tmp: 4*n_items int64 of scratch. */
int xmap_path_weights(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                      const int64_t *src_ptr, const int32_t *src_idx, const uint8_t *src_flag, const int64_t *rnn_ptr,
                      const int32_t *rnn_idx, int64_t *tmp /*[4][I]*/, int64_t *paths /*[I]*/);

/* ExtendSim.sim_extend + get_final_extension (core/extender.py:46-217), start-sharded and streamed:
 * every path of final_nonjoint_extend / final_joint_extend is enumerated in registers with its
 * s_p, c_p (calculate_path_confidence, :83-89) and accumulated into (sum s_p c_p, sum c_p) per
 * (start, end); xsim = ratio (:198-201).  The sums are kept as double-double (error-free two-sum), so
 * they do not depend on the enumeration order.
 * Work units (built by the caller from xmap_path_weights): unit u = (unit_start, chunk unit_c of unit_G).
 *   unit_G == 1, unit_row == -1: one wave owns the start, uses its slot row and finalises it;
 *   unit_G  > 1: the start's (head, t) entries are dealt round-robin to unit_G consecutive units with
 *   dedicated rows unit_row .. unit_row+G-1 of hacc/htouched; heavy_unit0[h] = first unit of heavy start h;
 *   the rows are merged and finalised after the enumeration.
 * For every start that has a unit: n_cand[start] = number of distinct ends, top_end/top_val[start][XMAP_TOPC]
 * = the candidates a Generator reads (stable sort by -|xsim|, generator.py:85,109; ties by ascending end).
 * If xs_cap > 0 the full candidate lists are also written: xs_off[start], entries (xs_end, xs_val);
 * needs xs_cap >= total (reported in h_counters[0]; XMAP_ERR_CAPACITY otherwise).  h_counters[1] = paths.
 * scratch (zero-filled by the caller, left zero): acc[n_slots][I][4], hacc[n_rows][I][4] doubles;
 * touched[n_slots][I], htouched[n_rows][I] int32. */
int xmap_extend_paths(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt,
                      const int32_t *kcol, const double *kval, const uint8_t *flags, const int64_t *att_ptr,
                      const int32_t *att_idx, const double *att_val, const int64_t *src_ptr, const int32_t *src_idx,
                      const double *src_val, const uint8_t *src_flag, const int64_t *rnn_ptr, const int32_t *rnn_idx,
                      const double *rnn_val, int32_t n_units, const int32_t *unit_start, const int32_t *unit_c,
                      const int32_t *unit_G, const int32_t *unit_row, int32_t *unit_nt, int32_t n_heavy,
                      const int32_t *heavy_unit0, int32_t n_slots, double *acc, int32_t *touched, double *hacc,
                      int32_t *htouched, int32_t *n_cand, int32_t *top_end, double *top_val, int64_t xs_cap,
                      int64_t *xs_off, int32_t *xs_end, double *xs_val, int64_t *d_counters /*[4] device*/,
                      int64_t *h_counters /*[4]*/);

/* Second formulation of the enumeration ("middle lists", stage_b.hip): per non-bridge record x' (nb_list, n_nb of
 * them; nb_id[i] = position of i in nb_list or -1) the middles (t,s,x) of all joint paths through x' are materialised
 * once as 64-byte records grouped per tile (x', x).  A dense n_nb x n_nb table gives the tile sizes:
 *   xmap_mid_tally : tile_cnt[x'][x] and ng[x'] = number of non-empty tiles of x';
 *   (caller: exclusive scans tile_cnt -> tile_off [n_nb*n_nb+1], ng -> dir_ptr [n_nb+1]; allocates dir, midX)
 *   xmap_mid_place : the tile directory of every x' (24 B per tile: x, 1+|NN(x)|, count, offset) and the records.
 * xmap_extend_paths2 is xmap_extend_paths with the joint paths streamed from these lists, tile-major: the (up to 64)
 * heads of a start are merged by x, a lane keeps its end's double-double sums in registers across all tiles (x', x)
 * of the start's heads, so a start's row is touched once per (start, x) instead of once per path.  Same results
 * (the sums are exact). */
#ifdef XMAP_CROSSCHECK   /* test formulation: exported by libxmap_hip_xcheck.so only (csrc/Makefile), never by the product library */
int xmap_mid_tally(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                   const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                   const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                   const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                   int32_t *tile_cnt /*[n_nb*n_nb]*/, int32_t *ng /*[n_nb]*/);
int xmap_mid_place(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                   const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                   const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                   const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                   int32_t *tile_cnt, const int64_t *tile_off /*[n_nb*n_nb+1]*/, const int64_t *dir_ptr /*[n_nb+1]*/,
                   void *dir /*24 B per tile*/, void *midX /*64 B per record*/);
#endif /* XMAP_CROSSCHECK */
/* Row-wise construction of the same lists (default, any n_nb): one block per x' keeps the tile sizes of its row in LDS --
 * XMAP_MID_ROWS_SPAN columns at a time; a row with more non-bridge items is built in column ranges, one after the
 * other -- so there is no n_nb x n_nb table and no global atomic:
 *   xmap_mid_rows_count : ng[x'] = non-empty tiles, nrec[x'] = records of x';
 *   (caller: exclusive scans ng -> dir_ptr [n_nb+1], nrec -> rec_ptr [n_nb+1]; allocates dir, midX)
 *   xmap_mid_rows_place : the tile directory (in x order) and the records of every x'.
 * Output identical to xmap_mid_tally / xmap_mid_place up to the order of the records inside a tile. */
#define XMAP_MID_ROWS_SPAN 36864   /* columns of a row whose counters fit the LDS of a block (4 B each; 160 KB per CU on gfx950, 7 KB of it the walk's tables) */
int xmap_mid_rows_count(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                        const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                        const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                        const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                        int32_t *ng /*[n_nb]*/, int64_t *nrec /*[n_nb]*/);
int xmap_mid_rows_place(void *stream, int32_t n_items, int top_k, const uint8_t *cls, const int32_t *kcnt, const int32_t *kcol,
                        const double *kval, const uint8_t *flags, const int64_t *att_ptr, const int32_t *att_idx,
                        const double *att_val, const int64_t *src_ptr, const int32_t *src_idx, const double *src_val,
                        const uint8_t *src_flag, int32_t n_nb, const int32_t *nb_list, const int32_t *nb_id,
                        const int64_t *dir_ptr /*[n_nb+1]*/, const int64_t *rec_ptr /*[n_nb+1]*/, void *dir /*24 B per tile*/,
                        void *midX /*64 B per record*/);
#ifdef XMAP_CROSSCHECK   /* test formulation: exported by libxmap_hip_xcheck.so only (csrc/Makefile), never by the product library */
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
                       const int64_t *dir_ptr, const int32_t *ng);
#endif /* XMAP_CROSSCHECK */

/* ---- extension, column form (default) ----------------------------------------------------------------------------
 * Same work units and results as xmap_extend_paths2 (units = starts, heavy starts cut into unit_G chunks with dedicated
 * rows), rebuilt around what bounds it: a column (start, x) is ONE set of lanes -- W ends x S record
 * slices, S = 4 / 2 / 1 by the column's width -- and one row update; rows are indexed by the rank of the end among
 * the n_ends items that can end a path at all (xmap_end_universe: urank[item] / uitem[rank]) instead of by item; the
 * ends of a column come from one table of 32-byte records; sums are (value, error) pairs folded at the end.
 * Scratch (zero-filled by the caller once, returned zeroed): acc [n_slots][n_ends][4] doubles, touched
 * [n_slots][n_ends], hacc [rows][n_ends][4], htouched [rows][n_ends].  fast_div: every edge has a positive mutuality
 * and |sim * mutu| within 2^+-400 (what stage A produces), so the division of a path is the bare
 * reciprocal-refinement sequence.  Results as xmap_extend_paths (same exact sums). */
typedef struct xmap_ext_tables {
    int32_t n_items, top_k;
    const uint8_t *cls; const int32_t *kcnt; const int32_t *kcol; const double *kval; const uint8_t *flags;
    const int64_t *att_ptr; const int32_t *att_idx; const double *att_val;
    const int64_t *src_ptr; const int32_t *src_idx; const double *src_val; const uint8_t *src_flag;
    const int64_t *rnn_ptr; const int32_t *rnn_idx; const double *rnn_val;
    int32_t n_nb; const int32_t *nb_id; const int32_t *nb_list; const void *midX; const void *dir; const int64_t *dir_ptr;
    int32_t n_ends; const int32_t *urank; const int32_t *uitem;
} xmap_ext_tables;
typedef struct xmap_path_units {
    int32_t n_units; const int32_t *unit_start; const int32_t *unit_c; const int32_t *unit_G; const int32_t *unit_row;
    int32_t *unit_nt; int32_t n_heavy; const int32_t *heavy_unit0;
} xmap_path_units;
typedef struct xmap_path_rows { int32_t n_slots; double *acc; int32_t *touched; double *hacc; int32_t *htouched; } xmap_path_rows;
typedef struct xmap_path_out {
    int32_t *n_cand; int32_t *top_end; double *top_val; int64_t xs_cap; int64_t *xs_off; int32_t *xs_end; double *xs_val;
} xmap_path_out;
/* fast_div precondition of xmap_extend_cols over a similarity matrix: *h_fast_ok = 1 iff every kept pair has mutu >= 1
 * and sim * mutu is zero or within 2^+-400 (always true for what stage A produces; 0 for a matrix that carries its own
 * frac column, i.e. generic records).  One pass over the pairs; synchronises. */
int xmap_edge_ranges(void *stream, const xmap_sim *S, int32_t *h_fast_ok);
int xmap_end_universe(void *stream, const xmap_ext_tables *T, int32_t *mark /*[I] scratch*/, int64_t *rank /*[I+1] scratch*/,
                      int32_t *urank /*[I]*/, int32_t *uitem /*[I]*/, int64_t *h_n_ends);
int xmap_extend_cols(void *stream, const xmap_ext_tables *T, const xmap_path_units *U, const xmap_path_rows *R,
                      const xmap_path_out *O, int fast_div, int64_t *d_counters /*[8] device*/,
                      int64_t *h_counters /*[8]: candidates, paths, -, -, row updates (read-modify-writes of row entries)*/);
/* Size limits of the extension live in the HOST layer, not here: the Python engine refuses an extension of more than
 * XMAP_MAX_PATHS paths (5e12) before calling xmap_extend_cols and enumerates path by path (xmap_extend_paths) when the middle
 * lists would exceed XMAP_MID_BUDGET_GB (100 GB); xmap_ctx_extend refuses both.  INTEGRATION.md, "Limits a caller can hit".
 * accumulator rows xmap_extend_cols can keep busy: one per wavefront resident on the device (compute units x SIMDs x the
 * kernel's waves per SIMD); what a caller sizes xmap_path_rows.n_slots with (more rows are never touched). */
int xmap_extend_cols_slots(int32_t *h_n_slots);

/* ---- planning steps of stage B (round 1 did these with torch ops on the device) ----------------------------------
 * xmap_nb_index : nb_list = the non-bridge records (cls == 2) in item order, nb_id[item] = position in it or -1.
 * xmap_path_plan: work units of the enumeration from the exact per-start path counts (xmap_path_weights): starts in
 *   [start_lo, start_hi) with more than `chunk` paths (chunk_div > 0: chunk = max(2^22, paths in the range / chunk_div))
 *   are cut into ceil(paths / chunk) chunks with dedicated rows
 *   (chunk doubles until their rows fit max_rows); units heaviest first (own stable radix sort), the chunks of a start
 *   consecutive.  h_out = {units, heavy starts, rows, paths in the range, chunk used}; XMAP_ERR_CAPACITY with the
 *   counts filled in when the unit arrays (cap_units entries each; heavy_unit0: one per heavy start) are too small
 *   (call with cap_units = 0 and NULL arrays to size them).
 * xmap_end_order: reorders the end universe (urank / uitem of xmap_end_universe, in place) so that every end sits
 *   with the first column x whose end list {x} + NN(x) holds it: the ends of a column are then neighbours in a row. */
int xmap_nb_index(void *stream, int32_t n_items, const uint8_t *cls, int32_t *nb_list, int32_t *nb_id, int64_t *h_n_nb);
int xmap_path_plan(void *stream, int32_t n_items, const int64_t *paths, int32_t start_lo, int32_t start_hi, int64_t chunk,
                   int64_t chunk_div, int64_t max_rows, int64_t cap_units, int32_t *unit_start, int32_t *unit_c, int32_t *unit_G,
                   int32_t *unit_row, int32_t *heavy_unit0, int64_t *h_out);
int xmap_end_order(void *stream, int32_t n_items, int top_k, int32_t n_nb, const int32_t *nb_list, const int32_t *kcnt,
                   const int32_t *kcol, int32_t n_ends, int32_t *urank, int32_t *uitem);

/* Candidate arrays from explicit X-Sim lists (an extended_simRDD that did not come from this engine,
 * e.g. a canonically re-fed one): CSR (xs_ptr, xs_end, xs_val) -> n_cand, top_end, top_val as above. */
int xmap_topc_from_lists(void *stream, int32_t n_items, const int64_t *xs_ptr, const int32_t *xs_end,
                         const double *xs_val, int32_t *n_cand, int32_t *top_end, double *top_val);

/* ---- dense item-factor variant (BASELINE.json configs[4]; no counterpart in the reference) --------------------
 * xsim(t, s) = cosine of K-dimensional item factors: a dense (n_t x K) x (K x n_s) contraction on the fp32 matrix
 * cores (v_mfma_f32_32x32x2_f32; the accumulation is the k-ordered fmaf chain, bit for bit) with the per-row top-k
 * by (|sim| desc, source index asc) fused behind it.  normalize: Fn = F / ||F|| (norm in fp64).  top_k <= 64,
 * dim in {64, 128}.  out_idx/out_val: [n_t][top_k], unused entries -1 / 0.  The (row block x source tile) work grid
 * is cut into one equal share per resident workgroup; a row block whose tiles fall into several shares is ranked in
 * pieces that a merge kernel folds.  xmap_dense_layout (host only) returns the pieces per row the scratch
 * part_idx/part_val [n_t][n_pieces][top_k] must hold (1: scratch unused, may be NULL). */
int xmap_dense_normalize(void *stream, int32_t n, int32_t dim, const float *F, float *Fn);
int xmap_dense_layout(int32_t n_t, int32_t n_s, int32_t *n_pieces);
int xmap_dense_topk(void *stream, int32_t n_t, int32_t n_s, int32_t dim, const float *Ft, const float *Fs, int32_t top_k,
                    int32_t n_pieces, int32_t *part_idx, float *part_val, int32_t *out_idx, float *out_val);

/* ---- RecommenderPrivacy.nonprivate_neighbor_selection (core/recommenderPrivacy.py:22-35,141-152)
 * over the RecommenderSim rows (CSR of xmap_sim2_scatter with ls): per item the `keep` (= mapping_range, <= 64)
 * neighbours by (|sim| desc, neighbour index asc) -- the reference's stable sort keeps the arrival order of equal
 * similarities, which Spark does not define.  out_col/out_sim/out_ls: [I][keep], unused entries -1 / 0; out_cnt [I]. */
int xmap_rec_select(void *stream, int32_t n_items, const int64_t *row_ptr, const int32_t *col, const double *sim,
                    const double *ls, int32_t keep, int32_t *out_cnt, int32_t *out_col, double *out_sim, double *out_ls);

/* ---- RecommenderPrediction.item_based_prediction (core/recommenderPrediction.py:26-105): one test pair
 * (user, item) per thread.  test_item = -1: the item has no neighbour list (the reference emits ()), status 1.  Neighbour
 * lists nb_* in the order of the similarity broadcast; the ratings of an item rt_* sorted by user index, stable, so that a
 * user's ratings of an item keep their list order (test_user = -1: a user without ratings); times as numbers whose order
 * and ties are those of the reference's time objects.  wtab[d] = exp(-alpha d) for d = 0 .. n_w - 1, made by the host with
 * the reference's np.exp.  out_plain / out_decay: bound_rating(prediction without / with temporal decay) (:17-23, :86-97);
 * status 2: more than 64 evidence entries (or more ranks than wtab holds) -- the caller decides that pair on the host.
 * The reference tests `uid in rater_id` (substring); the host maps that to user indices (equality when all ids have one
 * length). */
int xmap_predict(void *stream, int64_t n_test, const int32_t *test_user, const int32_t *test_item, const int64_t *nb_ptr,
                 const int32_t *nb_item, const double *nb_sim, const int64_t *rt_ptr, const int32_t *rt_user, const double *rt_rating,
                 const double *rt_time, const double *item_avg, const double *wtab, int32_t n_w, double *out_plain,
                 double *out_decay, int32_t *status);

/* ---- stage C: generator_pipeline (utils/assist.py:136-150) ---------------------------------- */

/* Generator.cross_private_mapping / cross_nonprivate_mapping (core/generator.py:27-111) + map_to_dict
 * (utils/assist.py:210-215).  private: choice = candidate 0 (arg-max |xsim|).  non-private: choice =
 * top4[picks[start]], picks drawn on the host with np.random.randint in ascending start order.
 * n_top[start] = min(private ? 10 : 4, n_cand); map_src2tgt[choice] = largest start choosing it, else -1. */
int xmap_select_map(void *stream, int32_t n_items, int private_flag, const int32_t *n_cand, const int32_t *top_end,
                    const int32_t *picks /* may be NULL */, int32_t *n_top, int32_t *choice, int32_t *map_src2tgt);

/* Generator.build_alterEgo (core/generator.py:113-157).  count pass: cnt_t[u] pass-through rows
 * ("T:" in iid), cnt_m[u] AlterEgo rows (distinct mapped targets, first-seen order).  fill pass writes
 * rows [off_t[u]..) and [n_t_total + off_m[u]..): (user, item, rating = mean fp32, time of first row). */
int xmap_alterego_count(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, int32_t *cnt_t, int32_t *cnt_m,
                        int64_t *d_profiles /* [64] device, zeroed by the caller: the users with at least one output row are
                                               added to these 64 counters (their sum is the number of profiles); or NULL */);
int xmap_alterego_fill(void *stream, const xmap_ratings *R, const int32_t *map_src2tgt, const int64_t *off_t,
                       const int64_t *off_m, int64_t n_t_total, int32_t *out_user, int32_t *out_item,
                       double *out_rating, int64_t *out_time);

/* ==== coarse, handle-based entry points =============================================================
 * What a host in any language binds to replace the three pipelines: plain host buffers in, plain host buffers out,
 * sizes reported by the stage call; the library owns every device buffer, prefix sum, overflow retry and work-unit
 * plan (csrc/api.hip).  All ids are int32 indices into the caller's lexicographically sorted id tables; the four
 * per-item predicate arrays are the string tests of the reference evaluated once per item (xmap/engine/ids.py):
 * prefix_cls (iid[:2] class, baselinerSim.py:191), suffix_cls (iid[-2:] class), contains_mask (bit c: the suffix of
 * class c occurs in the id, extender.py:29-35), flags (bit 0 "S:" in iid, bit 1 "T:" in iid).
 *   xmap_ctx_upload_ratings : trainRDD in index space, CSR by user in trainRDD / profile order      (assist.py:66)
 *   xmap_ctx_item_sim       : baseliner_calculate_sim_pipeline (assist.py:66-77) -> n_kept directed pairs kept
 *   xmap_ctx_sim_download   : CSR by first item (row_ptr [I+1], col/sim/mutu/n_ij [n_kept]; rows not sorted), item info
 *                             [I][4], user averages [U]; any pointer may be NULL
 *   xmap_ctx_extend         : extender_pipeline (assist.py:80-102), lazy: per start item the number of candidates and
 *                             the XMAP_TOPC best by (|xsim| desc, end asc); n_out = sum of the candidate counts
 *   xmap_ctx_ext_lists      : the (start, [(end, xsim)*]) lists themselves (xs_off [I], xs_end/xs_val [n_out]): the
 *                             enumeration runs once more into buffers of exactly n_out entries
 *   xmap_ctx_candidates     : n_top[start] = min(candidates, 4): what cross_nonprivate_mapping draws from
 *                             (generator.py:109-110); the caller draws picks[start] in [0, n_top - 1) itself
 *   xmap_ctx_generate       : generator_pipeline (assist.py:136-150): private: arg-max |xsim|; else picks [I];
 *                             choice [I] (or NULL) receives the chosen source item per start (-1: none)
 *   xmap_ctx_gen_download   : AlterEgo rows (user, item, rating fp64, time), pass-through target rows first
 * Errors: negative return code, text in xmap_last_error(). */
typedef struct xmap_ctx xmap_ctx;

/* ---- native feeder (csrc/feeder.hip; host code, no GPU): raw lines `uid iid rating unix_ts` (reference README.md:41-42) ->
 * id tables + CSR + predicate arrays, with the reference's clean stage in between (core/baselinerClean.py:40-101: fields =
 * re.split(r"\s+"), local-time year in [year_from, year_to], item id = field + label, the latest rating of an item wins in
 * place, users with fewer than min_ratings ratings dropped; users in first-seen order, items in lexicographic id order).
 *   xmap_feed_text   : one domain's text                       xmap_feed_merge : source + target feed of one problem
 *   xmap_feed_texts  : the domains of one problem in one call (= the merge of their feeds, without building them)
 *   xmap_feed_sizes  : {users, items, ratings, bytes of the user ids, bytes of the item ids, lines read, lines in the period}
 *   xmap_feed_arrays : copies into caller buffers (any but user_ptr may be NULL); when = the timestamps as doubles
 *   xmap_feed_ids    : the id strings back to back + offsets [n + 1] (which = 0 users, 1 items; bytes may be NULL); which | 2:
 *                      a newline behind every id (bytes + n in all; offsets may then be NULL)
 *   xmap_ctx_upload_feed : the coarse ABI's upload straight from a feed (xmap_ctx_upload_ratings on its arrays)
 *   xmap_feed_format : test / bench utility, the inverse for one domain (items [item_lo, item_hi) of a CSR -> text) */
typedef struct xmap_feed xmap_feed;
int xmap_feed_text(const char *text, int64_t len, int32_t year_from, int32_t year_to, const char *label, int32_t min_ratings,
                   xmap_feed **out);
int xmap_feed_texts(int32_t n_parts, const char *const *texts, const int64_t *lens, const char *const *labels, int32_t year_from,
                    int32_t year_to, int32_t min_ratings, xmap_feed **out);
int xmap_feed_merge(const xmap_feed *a, const xmap_feed *b, xmap_feed **out);
int xmap_feed_sizes(const xmap_feed *f, int64_t *sizes /*[7]*/);
int xmap_feed_arrays(const xmap_feed *f, int64_t *user_ptr, int32_t *item, double *rating, double *when, int32_t *prefix_cls,
                     int32_t *suffix_cls, uint32_t *contains_mask, uint8_t *flags);
int xmap_feed_ids(const xmap_feed *f, int32_t which, char *bytes, int64_t *offsets);
void xmap_feed_free(xmap_feed *f);
int xmap_ctx_upload_feed(xmap_ctx *ctx, const xmap_feed *f);
int xmap_feed_format(int64_t n_users, const int64_t *user_ptr, const int32_t *item, const float *rating, const int64_t *when,
                     const char *uid_fmt, const char *iid_fmt, const int64_t *item_number, int32_t item_lo, int32_t item_hi,
                     char *out, int64_t cap, int64_t *written);

int xmap_ctx_create(int device, xmap_ctx **out);
void xmap_ctx_destroy(xmap_ctx *ctx);
int xmap_ctx_upload_ratings(xmap_ctx *ctx, int64_t n_users, int32_t n_items, const int64_t *user_ptr, const int32_t *item,
                            const float *rating, const int64_t *time, const int32_t *prefix_cls, const int32_t *suffix_cls,
                            const uint32_t *contains_mask, const uint8_t *flags);
int xmap_ctx_item_sim(xmap_ctx *ctx, int method, int cap, int64_t *n_kept, int64_t *n_evaluated);
int xmap_ctx_sim_download(xmap_ctx *ctx, int64_t *row_ptr, int32_t *col, double *sim, int32_t *mutu, int32_t *nij, double *info,
                          double *user_avg);
int xmap_ctx_extend(xmap_ctx *ctx, int top_k, int64_t *n_out, int64_t *n_paths);
int xmap_ctx_ext_download(xmap_ctx *ctx, int32_t *n_cand, int32_t *top_end, double *top_val);
int xmap_ctx_ext_lists(xmap_ctx *ctx, int64_t *xs_off, int32_t *xs_end, double *xs_val);
int xmap_ctx_candidates(xmap_ctx *ctx, int32_t *n_top);
int xmap_ctx_generate(xmap_ctx *ctx, int private_flag, const int32_t *picks, int32_t *choice, int64_t *n_rows,
                      int64_t *n_target_rows);
int xmap_ctx_gen_download(xmap_ctx *ctx, int32_t *user, int32_t *item, double *rating, int64_t *time);

#ifdef __cplusplus
}
#endif
#endif
