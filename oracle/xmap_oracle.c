/*
 * xmap_oracle.c -- CPU restatement (plain C, index space) of X-MAP's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared with
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The product
 * path (x-map_amd/) never includes, links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is checked against golden vectors
 * captured by importing the reference's own Python modules in the build
 * container (oracle/ref_harness/make_golden.py -> tests/golden/*.npz); see
 * tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/code/xmap).  Items are int32 indices in lexicographic order
 * of the reference's id strings; users are indices in trainRDD order.  String
 * predicates of the reference are precomputed per item by the caller:
 *   prefix_cls[i]    class id of iid[:2]                (core/baselinerSim.py:191)
 *   suffix_cls[i]    class id c of iid[-2:]             (core/extender.py:29)
 *   contains_mask[i] bit c set iff class-c suffix string occurs in iid (extender.py:32,34)
 *   flags[i]         bit0: "S:" in iid, bit1: "T:" in iid  (extender.py:68,79,174,175; generator.py:157)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <stdio.h>

#define XO_COSINE 0
#define XO_ADJUST 1

static double xo_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ utils */
/* numpy's float add-reduce kernel (pairwise summation, block 128, unroll 8);
 * np.sum / np.mean on a contiguous float64 array return 0.0 + this. */
static double np_pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}
static double np_sum(const double *a, int64_t n) { return 0.0 + np_pairwise_sum(a, n); }
static void dd_add(double *hi, double *lo, double x);

/* --------------------------------------------------------------- stage A */

/* A2: core/baselinerSim.py:17-38  get_universal_user_info
 *   avg = float(sum(r)/len), norm2 = float(sqrt(sum(r*r)))  (python sum: left to right) */
void xo_user_info(int64_t U, const int64_t *ptr, const float *rating, double *avg, double *norm2) {
    for (int64_t u = 0; u < U; u++) {
        double s = 0.0, q = 0.0;
        for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) {
            double r = (double)rating[e];
            s += r;
            q += r * r;
        }
        int64_t d = ptr[u + 1] - ptr[u];
        avg[u] = d ? s / (double)d : 0.0;
        norm2[u] = sqrt(q);
    }
}

/* A3: core/baselinerSim.py:40-82  get_universal_item_info
 *   combineByKey in trainRDD order (one partition): (sum r, sum r**2, sum (r-avg_u)**2, n)
 *   -> (1.0*sum/n, sqrt(sum r**2), sqrt(sum (r-avg_u)**2), 1.0*n).  `**2` is C pow(); the third sum
 *   is accumulated error-free (order-independent canonical value). */
void xo_item_info(int64_t U, int32_t I, const int64_t *ptr, const int32_t *item, const float *rating,
                  const double *uavg, double *info /* [I][4] */) {
    double *acc = (double *)calloc((size_t)I * 4, sizeof(double));
    double *lo = (double *)calloc((size_t)I, sizeof(double));
    for (int64_t u = 0; u < U; u++)
        for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) {
            double r = (double)rating[e];
            double *x = acc + (size_t)item[e] * 4;
            double d = r - uavg[u];
            x[0] += r;
            x[1] += pow(r, 2.0);
            /* canonical value: exact sum of the fp64 squares (see dd_add); the reference's
             * left-to-right sum of pow(d, 2) differs from it by a few ulp at most */
            dd_add(&x[2], &lo[item[e]], d * d);
            x[3] += 1.0;
        }
    free(lo);
    for (int32_t i = 0; i < I; i++) {
        double *x = acc + (size_t)i * 4, *o = info + (size_t)i * 4;
        o[0] = x[3] > 0 ? 1.0 * x[0] / x[3] : 0.0;
        o[1] = sqrt(x[1]);
        o[2] = sqrt(x[2]);
        o[3] = 1.0 * x[3];
    }
    free(acc);
}

typedef struct {
    int32_t I;
    int64_t n_eval; /* D: directed pairs with >=1 co-rater that were evaluated */
    int64_t n_contrib; /* P: directed co-rating contributions */
    int64_t *row_ptr; /* [I+1] kept pairs, CSR by first item, cols ascending */
    int32_t *col;
    double *sim;
    int32_t *mutu;
    int32_t *nij;
    double seconds[3]; /* wall clock of the call's phases: item-major copy of the ratings, the rows (all pair work), output concatenation */
} XoSim;

typedef struct {
    int32_t *col;
    double *sim;
    int32_t *mutu;
    int32_t *nij;
    int64_t n, cap;
} Arena;
static void arena_push(Arena *a, int32_t c, double s, int32_t m, int32_t n) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4096;
        a->col = (int32_t *)realloc(a->col, a->cap * sizeof(int32_t));
        a->sim = (double *)realloc(a->sim, a->cap * sizeof(double));
        a->mutu = (int32_t *)realloc(a->mutu, a->cap * sizeof(int32_t));
        a->nij = (int32_t *)realloc(a->nij, a->cap * sizeof(int32_t));
    }
    a->col[a->n] = c; a->sim[a->n] = s; a->mutu[a->n] = m; a->nij[a->n] = n; a->n++;
}
static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* A4-A9: core/baselinerSim.py:176-216 (+ :84-95, :97-174).
 * Row-wise restatement: for item i, walk its raters in trainRDD order (that is the
 * order reduceByKey concatenates the co-rater triples in, one partition), collect the
 * per-pair term lists, then apply exactly the reference's reductions:
 *   cosine : python sum() left to right of 1.0*r_i*r_j                        (:126-131)
 *   adjust : np.sum((rx-avg)*(ry-avg))  -> exact sum of the same fp64 terms, rounded once (:156-164)
 *   sim    = (cos * min(n,cap)) / cap, cos = dot/(norm_i*norm_j) if (norm_i*norm_j) else 0.0   (:84-95)
 *   mutu   = #{(r_i>=avg_i & r_j>=avg_j) | (r_i<avg_i & r_j<avg_j)}            (:97-113)
 *   frac   = 1.0*mutu/(n_i+n_j-n)                                              (:139-141,:171-173)
 *   keep   iff 0.0 not in (sim, mutu, frac)                                    (:198,:207)
 * Only users with >= 2 ratings form pairs (:184-185).
 * Rows in [row_lo,row_hi) only (bounded samples for the CPU baseline); others stay empty. */
XoSim *xo_item_sim(int method, int cap, int64_t U, int32_t I, const int64_t *ptr, const int32_t *item,
                   const float *rating, const double *uavg, const double *info, int nthreads,
                   int32_t row_lo, int32_t row_hi) {
    /* CSC (item -> raters ascending user index) */
    const double t_begin = xo_now();
    int64_t nnz = ptr[U];
    int64_t *iptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    for (int64_t e = 0; e < nnz; e++) iptr[item[e] + 1]++;
    for (int32_t i = 0; i < I; i++) iptr[i + 1] += iptr[i];
    int32_t *iuser = (int32_t *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    double *irat = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));   /* np.float64: AlterEgo ratings are means (generator.py:134) */
    {
        int64_t *cur = (int64_t *)malloc((size_t)I * sizeof(int64_t));
        memcpy(cur, iptr, (size_t)I * sizeof(int64_t));
        for (int64_t u = 0; u < U; u++)
            for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) {
                int64_t p = cur[item[e]]++;
                iuser[p] = (int32_t)u;
                irat[p] = rating[e];
            }
        free(cur);
    }
    if (nthreads < 1) nthreads = 1;
    const double t_csc = xo_now();
    Arena *arenas = (Arena *)calloc((size_t)nthreads, sizeof(Arena));
    int32_t *row_thr = (int32_t *)calloc((size_t)I, sizeof(int32_t));
    int64_t *row_off = (int64_t *)calloc((size_t)I, sizeof(int64_t));
    int64_t *row_cnt = (int64_t *)calloc((size_t)I, sizeof(int64_t));
    int64_t n_eval = 0, n_contrib = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+ : n_eval, n_contrib)
#endif
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        Arena *A = &arenas[tid];
        int32_t *cnt = (int32_t *)calloc((size_t)I, sizeof(int32_t));   /* n_ij per partner */
        int32_t *pos = (int32_t *)calloc((size_t)I, sizeof(int32_t));   /* fill cursor / group start */
        int32_t *mut = (int32_t *)calloc((size_t)I, sizeof(int32_t));
        int32_t *touched = (int32_t *)malloc((size_t)I * sizeof(int32_t));
        int64_t bufcap = 1024;
        int32_t *bj = (int32_t *)malloc(bufcap * sizeof(int32_t));
        double *bt = (double *)malloc(bufcap * sizeof(double));
        double *grouped = (double *)malloc(bufcap * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int32_t i = row_lo; i < row_hi; i++) {
            double avg_i = info[(size_t)i * 4 + 0];
            int64_t w = 0, nt = 0;
            for (int64_t p = iptr[i]; p < iptr[i + 1]; p++) {
                int32_t u = iuser[p];
                int64_t d = ptr[u + 1] - ptr[u];
                if (d < 2) continue;
                double ri = (double)irat[p];
                int ge_i = ri >= avg_i;
                for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) {
                    int32_t j = item[e];
                    if (j == i) continue;
                    double rj = (double)rating[e];
                    if (w == bufcap) {
                        bufcap *= 2;
                        bj = (int32_t *)realloc(bj, bufcap * sizeof(int32_t));
                        bt = (double *)realloc(bt, bufcap * sizeof(double));
                        grouped = (double *)realloc(grouped, bufcap * sizeof(double));
                    }
                    bj[w] = j;
                    bt[w] = (method == XO_COSINE) ? (1.0 * ri * rj) : ((ri - uavg[u]) * (rj - uavg[u]));
                    w++;
                    if (cnt[j]++ == 0) touched[nt++] = j;
                    int ge_j = rj >= info[(size_t)j * 4 + 0];
                    mut[j] += (ge_i == ge_j);
                }
            }
            n_contrib += w;
            n_eval += nt;
            qsort(touched, (size_t)nt, sizeof(int32_t), cmp_i32);
            int64_t off = 0;
            for (int64_t t = 0; t < nt; t++) { pos[touched[t]] = (int32_t)off; off += cnt[touched[t]]; }
            for (int64_t k = 0; k < w; k++) grouped[pos[bj[k]]++] = bt[k];   /* stable: keeps rater order */
            row_thr[i] = tid;
            row_off[i] = A->n;
            off = 0;
            for (int64_t t = 0; t < nt; t++) {
                int32_t j = touched[t];
                int32_t n = cnt[j];
                const double *terms = grouped + off;
                off += n;
                double dot;
                if (method == XO_COSINE) {
                    dot = 0.0;
                    for (int32_t k = 0; k < n; k++) dot += terms[k];
                } else {
                    /* canonical value: the correctly rounded EXACT sum of the fp64 terms (error-free
                     * double-double accumulation).  The reference's np.sum rounds in pairwise order of a
                     * co-rater list whose order depends on Spark partitioning; the exact sum is the
                     * order-independent value both oracle and HIP path adopt (differs from np.sum by
                     * <= 1e-13 relative on the golden vectors, every discrete output identical). */
                    double hi = 0.0, lo = 0.0;
                    for (int32_t k = 0; k < n; k++) dd_add(&hi, &lo, terms[k]);
                    dot = hi;
                }
                int c1 = (method == XO_COSINE) ? 1 : 2;
                double np_ = info[(size_t)i * 4 + c1] * info[(size_t)j * 4 + c1];
                double cs = np_ ? 1.0 * dot / np_ : 0.0;
                int32_t mn = n < cap ? n : cap;
                double sim = 1.0 * cs * (double)mn / (double)cap;
                double mutu = (double)mut[j];
                double frac = 1.0 * mutu / (info[(size_t)i * 4 + 3] + info[(size_t)j * 4 + 3] - (double)n);
                if (sim != 0.0 && mutu != 0.0 && frac != 0.0) arena_push(A, j, sim, mut[j], n);
                cnt[j] = 0; mut[j] = 0; pos[j] = 0;
            }
            row_cnt[i] = A->n - row_off[i];
        }
        free(cnt); free(pos); free(mut); free(touched); free(bj); free(bt); free(grouped);
    }
    const double t_rows = xo_now();
    XoSim *S = (XoSim *)calloc(1, sizeof(XoSim));
    S->I = I;
    S->n_eval = n_eval;
    S->n_contrib = n_contrib;
    S->row_ptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    for (int32_t i = 0; i < I; i++) S->row_ptr[i + 1] = S->row_ptr[i] + row_cnt[i];
    int64_t D = S->row_ptr[I];
    S->col = (int32_t *)malloc((size_t)(D ? D : 1) * sizeof(int32_t));
    S->sim = (double *)malloc((size_t)(D ? D : 1) * sizeof(double));
    S->mutu = (int32_t *)malloc((size_t)(D ? D : 1) * sizeof(int32_t));
    S->nij = (int32_t *)malloc((size_t)(D ? D : 1) * sizeof(int32_t));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int32_t i = 0; i < I; i++) {
        Arena *A = &arenas[row_thr[i]];
        int64_t o = S->row_ptr[i], c = row_cnt[i], s = row_off[i];
        if (!c) continue;
        memcpy(S->col + o, A->col + s, (size_t)c * sizeof(int32_t));
        memcpy(S->sim + o, A->sim + s, (size_t)c * sizeof(double));
        memcpy(S->mutu + o, A->mutu + s, (size_t)c * sizeof(int32_t));
        memcpy(S->nij + o, A->nij + s, (size_t)c * sizeof(int32_t));
    }
    for (int t = 0; t < nthreads; t++) { free(arenas[t].col); free(arenas[t].sim); free(arenas[t].mutu); free(arenas[t].nij); }
    free(arenas); free(row_thr); free(row_off); free(row_cnt); free(iptr); free(iuser); free(irat);
    S->seconds[0] = t_csc - t_begin; S->seconds[1] = t_rows - t_csc; S->seconds[2] = xo_now() - t_rows;
    if (getenv("XO_TIMING"))
        fprintf(stderr, "xo_item_sim: %d threads, csc %.3f s, rows %.3f s, concat %.3f s\n", nthreads, t_csc - t_begin,
                t_rows - t_csc, xo_now() - t_rows);
    return S;
}
void xo_sim_free(XoSim *S) {
    if (!S) return;
    free(S->row_ptr); free(S->col); free(S->sim); free(S->mutu); free(S->nij); free(S);
}

/* --------------------------------------------------------------- stage B */

typedef struct {
    int32_t I, k;
    uint8_t *bb;      /* [I] bridge flag            utils/assist.py:82-87 */
    uint8_t *cls;     /* [I] 0 dropped/absent, 1 BB record, 2 NB record   core/extender.py:16-44 */
    int32_t *cnt;     /* [I][2] list lengths: list 0 = BB_BB | NB_BB, list 1 = BB_NB | NB_NN */
    int32_t *col;     /* [I][2][k] */
    double *val;      /* [I][2][k][3] (sim, mutu, frac_mutu) */
    /* X-Sim output (CSR by start item, ends ascending)   core/extender.py:184-217 */
    int64_t n_paths;
    int64_t *xs_ptr;  /* [I+1] */
    int32_t *xs_end;
    double *xs_val;
    double path_seconds; /* wall time of the path enumeration + X-Sim loop alone (CPU baseline of stage B) */
} XoExt;

typedef struct { double a; int32_t j; int64_t pos; } SortEnt;
static int cmp_abs_desc(const void *x, const void *y) {
    const SortEnt *a = (const SortEnt *)x, *b = (const SortEnt *)y;
    if (a->a > b->a) return -1;
    if (a->a < b->a) return 1;
    return (a->j > b->j) - (a->j < b->j); /* stable on ascending-col input == ascending col */
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
/* (start,end) -> (sum s_p*c_p, sum c_p) open-addressing map */
/* The per-(start,end) sums are accumulated error-free (double-double, Knuth two-sum), i.e. to
 * ~2^-104: the reference's own result (np.dot through BLAS, np.sum pairwise; extender.py:198-201)
 * depends on an unknowable summation order at the 1e-16 level, so the canonical value adopted by
 * oracle and HIP path alike is the correctly rounded exact sum -- order-independent, and items with
 * identical path multisets tie exactly (tie-break: ascending end index). */
typedef struct { uint64_t *key; double *sc; double *scl; double *c; double *cl; uint64_t cap, n; } PMap;
static void dd_add(double *hi, double *lo, double x) {
    double s = *hi + x;
    double bb = s - *hi;
    double e = (*hi - (s - bb)) + (x - bb);
    e += *lo;
    double h2 = s + e;
    *lo = e - (h2 - s);
    *hi = h2;
}
static void pmap_init(PMap *m, uint64_t cap) {
    m->cap = cap; m->n = 0;
    m->key = (uint64_t *)malloc(cap * sizeof(uint64_t));
    memset(m->key, 0xff, cap * sizeof(uint64_t));
    m->sc = (double *)calloc(cap, sizeof(double));
    m->scl = (double *)calloc(cap, sizeof(double));
    m->c = (double *)calloc(cap, sizeof(double));
    m->cl = (double *)calloc(cap, sizeof(double));
}
static uint64_t pmap_hash(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
static void pmap_add(PMap *m, uint64_t k, double sc, double c);
static void pmap_grow(PMap *m) {
    PMap n; pmap_init(&n, m->cap * 2);
    for (uint64_t i = 0; i < m->cap; i++) if (m->key[i] != UINT64_MAX) {
        uint64_t h = pmap_hash(m->key[i]) & (n.cap - 1);
        while (n.key[h] != UINT64_MAX) h = (h + 1) & (n.cap - 1);
        n.key[h] = m->key[i]; n.sc[h] = m->sc[i]; n.scl[h] = m->scl[i]; n.c[h] = m->c[i]; n.cl[h] = m->cl[i]; n.n++;
    }
    free(m->key); free(m->sc); free(m->scl); free(m->c); free(m->cl); *m = n;
}
static void pmap_add(PMap *m, uint64_t k, double sc, double c) {
    if (m->n * 2 >= m->cap) pmap_grow(m);
    uint64_t h = pmap_hash(k) & (m->cap - 1);
    while (m->key[h] != UINT64_MAX && m->key[h] != k) h = (h + 1) & (m->cap - 1);
    if (m->key[h] == UINT64_MAX) { m->key[h] = k; m->n++; }
    dd_add(&m->sc[h], &m->scl[h], sc);
    dd_add(&m->c[h], &m->cl[h], c);
}

/* edge lookup with the reference's 4-way fallback   core/extender.py:100-113 */
static const double *knn_find(const XoExt *X, int32_t a, int32_t b) {
    for (int l = 0; l < 2; l++) {
        int32_t n = X->cnt[(size_t)a * 2 + l];
        const int32_t *c = X->col + ((size_t)a * 2 + l) * X->k;
        for (int32_t t = 0; t < n; t++) if (c[t] == b) return X->val + (((size_t)a * 2 + l) * X->k + t) * 3;
    }
    return NULL;
}
static const double *edge(const XoExt *X, int32_t a, int32_t b) {
    const double *v;
    if (X->cls[a] == 1 && (v = knn_find(X, a, b))) return v;
    if (X->cls[b] == 1 && (v = knn_find(X, b, a))) return v;
    if (X->cls[a] == 2 && (v = knn_find(X, a, b))) return v;
    if (X->cls[b] == 2 && (v = knn_find(X, b, a))) return v;
    return NULL; /* unreachable by construction */
}
/* s_p, c_p of one path   core/extender.py:83-89, :91-122 */
static void path_value(const XoExt *X, const int32_t *path, int len, double *sp, double *cp) {
    double den = 0.0, num = 0.0, c = 1.0;
    for (int e = 0; e + 1 < len; e++) {
        const double *v = edge(X, path[e], path[e + 1]);
        double sim = v ? v[0] : 0.0, mutu = v ? v[1] : 0.0, frac = v ? v[2] : 0.0;
        den = (e == 0) ? sim * mutu : den + sim * mutu;  /* python sum(): 0 + x == x */
        num = (e == 0) ? mutu : num + mutu;
        c = (e == 0) ? frac : c * frac;
    }
    *sp = num ? 1.0 * den / num : 0.0;
    *cp = c;
}
typedef struct { int32_t n[4]; int len; } Path;

/* B1-B6: utils/assist.py:80-133 + core/extender.py:16-217, from the kept pairs of stage A.
 * do_paths = 0 stops after the knn classification (B1-B4). */
XoExt *xo_extend(const XoSim *S, int top_k, const double *info, const int32_t *prefix_cls,
                 const int32_t *suffix_cls, const uint32_t *contains_mask, const uint8_t *flags,
                 int do_paths, int32_t s_lo, int32_t s_hi /* source records [s_lo, s_hi): a bounded sample for the CPU
                 baseline timing; (0, I) = everything */, double max_seconds /* > 0: stop the sample after this long */,
                 const uint8_t *want /* NULL, or [I]: only the paths that START at an item with want[item] != 0 are
                 enumerated -- the complete X-Sim lists of a sample of starts at a size where all paths are out of reach */) {
    int32_t I = S->I, k = top_k;
    XoExt *X = (XoExt *)calloc(1, sizeof(XoExt));
    X->I = I; X->k = k;
    X->bb = (uint8_t *)calloc((size_t)I, 1);
    X->cls = (uint8_t *)calloc((size_t)I, 1);
    X->cnt = (int32_t *)calloc((size_t)I * 2, sizeof(int32_t));
    X->col = (int32_t *)malloc((size_t)I * 2 * k * sizeof(int32_t));
    X->val = (double *)malloc((size_t)I * 2 * k * 3 * sizeof(double));
    X->xs_ptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    /* B1  assist.py:82-87 + baselinerSim.py:189-191: BB = {id1 : exists kept (id1,id2) with prefix differing} */
    for (int32_t i = 0; i < I; i++)
        for (int64_t p = S->row_ptr[i]; p < S->row_ptr[i + 1]; p++)
            if (prefix_cls[S->col[p]] != prefix_cls[i]) { X->bb[i] = 1; break; }
    /* B2-B3  baselinerSim.py:218-233 + extender.py:16-44 */
    int64_t maxlen = 1;
    for (int32_t i = 0; i < I; i++) { int64_t l = S->row_ptr[i + 1] - S->row_ptr[i]; if (l > maxlen) maxlen = l; }
    SortEnt *se = (SortEnt *)malloc((size_t)maxlen * sizeof(SortEnt));
    for (int32_t i = 0; i < I; i++) {
        int64_t lo = S->row_ptr[i], n = S->row_ptr[i + 1] - lo;
        if (!n) continue;
        for (int64_t t = 0; t < n; t++) { se[t].a = fabs(S->sim[lo + t]); se[t].j = S->col[lo + t]; se[t].pos = lo + t; }
        qsort(se, (size_t)n, sizeof(SortEnt), cmp_abs_desc);
        int32_t *cA = X->col + ((size_t)i * 2 + 0) * k, *cB = X->col + ((size_t)i * 2 + 1) * k;
        double *vA = X->val + ((size_t)i * 2 + 0) * k * 3, *vB = X->val + ((size_t)i * 2 + 1) * k * 3;
        int32_t nA = 0, nB = 0;
        for (int64_t t = 0; t < n; t++) {
            int32_t j = se[t].j; int64_t p = se[t].pos;
            double mutu = (double)S->mutu[p];
            double frac = 1.0 * mutu / (info[(size_t)i * 4 + 3] + info[(size_t)j * 4 + 3] - (double)S->nij[p]);
            int inA, inB;
            if (X->bb[i]) {
                int has = (contains_mask[j] >> suffix_cls[i]) & 1u; /* domain_label in pair[0] */
                inA = !has; inB = has;
            } else {
                inA = X->bb[j]; inB = 1; /* NB_NN keeps every neighbour (quirk A.5-3) */
            }
            if (inA && nA < k) { cA[nA] = j; vA[nA * 3] = S->sim[p]; vA[nA * 3 + 1] = mutu; vA[nA * 3 + 2] = frac; nA++; }
            if (inB && nB < k) { cB[nB] = j; vB[nB * 3] = S->sim[p]; vB[nB * 3 + 1] = mutu; vB[nB * 3 + 2] = frac; nB++; }
        }
        if (X->bb[i]) { X->cls[i] = 1; X->cnt[(size_t)i * 2] = nA; X->cnt[(size_t)i * 2 + 1] = nB; }
        else if (nA > 0) { X->cls[i] = 2; X->cnt[(size_t)i * 2] = nA; X->cnt[(size_t)i * 2 + 1] = nB; }
        /* else: no bridge neighbour -> dropped (extender.py:39) */
    }
    free(se);
    if (!do_paths) return X;

    /* B5a  extender.py:48-59,171-173: attach(b) = [(x, NN ids(x)) : x NB record, b in NB_BB(x)], x ascending */
    int64_t *aptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    for (int32_t x = 0; x < I; x++) if (X->cls[x] == 2)
        for (int32_t t = 0; t < X->cnt[(size_t)x * 2]; t++) aptr[X->col[((size_t)x * 2) * k + t] + 1]++;
    for (int32_t i = 0; i < I; i++) aptr[i + 1] += aptr[i];
    int32_t *ax = (int32_t *)malloc((size_t)(aptr[I] ? aptr[I] : 1) * sizeof(int32_t));
    {
        int64_t *cur = (int64_t *)malloc((size_t)I * sizeof(int64_t));
        memcpy(cur, aptr, (size_t)I * sizeof(int64_t));
        for (int32_t x = 0; x < I; x++) if (X->cls[x] == 2)
            for (int32_t t = 0; t < X->cnt[(size_t)x * 2]; t++) ax[cur[X->col[((size_t)x * 2) * k + t]]++] = x;
        free(cur);
    }
    PMap M; pmap_init(&M, 1 << 16);
    int64_t n_paths = 0;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    size_t p0cap = 1024;
    Path *P0 = (Path *)malloc(p0cap * sizeof(Path));
    uint8_t *wanty = NULL, *wantt = NULL;      /* wanty[x'] = some y' of NB_NN(x') is a wanted start */
    if (want) {
        wanty = (uint8_t *)calloc((size_t)(I ? I : 1), 1);
        for (int32_t x = 0; x < I; x++) if (X->cls[x] == 2)
            for (int32_t y = 0; y < X->cnt[(size_t)x * 2 + 1]; y++)
                if (want[X->col[((size_t)x * 2 + 1) * k + y]]) { wanty[x] = 1; break; }
        /* wantt[t] = a (t, s) record can produce a wanted start: t itself, an x' of attach(t), or a y' of such an x' */
        wantt = (uint8_t *)calloc((size_t)(I ? I : 1), 1);
        for (int32_t t = 0; t < I; t++) {
            if (want[t]) { wantt[t] = 1; continue; }
            for (int64_t a = aptr[t]; a < aptr[t + 1]; a++)
                if (want[ax[a]] || wanty[ax[a]]) { wantt[t] = 1; break; }
        }
    }
    /* B5b-B5e: every SRC record ((t,s), attach(s))   extender.py:61-70,174,176 */
    if (s_lo < 0) s_lo = 0;
    if (s_hi > I || s_hi < 0) s_hi = I;
    for (int32_t s = s_lo; s < s_hi; s++) {
        if (max_seconds > 0.0 && (s & 15) == 0) {
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            if ((double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec) > max_seconds) break;
        }
        if (!(flags[s] & 1) || aptr[s + 1] == aptr[s]) continue; /* "S:" in s, attach(s) non-empty (s is BB by construction) */
        for (int l = 0; l < 2; l++) for (int32_t q = 0; q < X->cnt[(size_t)s * 2 + l]; q++) {
            int32_t t = X->col[((size_t)s * 2 + l) * k + q];       /* v in knn_BB[s].keys() */
            if (!(flags[t] & 2)) continue;                         /* "T:" in v */
            if (want && !wantt[t]) continue;                       /* (no wanted start behind this record) */
            /* P0(t,s)  extender.py:134-138 / :154-158 */
            size_t np0 = 0;
            size_t need = 1;
            for (int64_t a = aptr[s]; a < aptr[s + 1]; a++) need += 1 + (size_t)X->cnt[(size_t)ax[a] * 2 + 1];
            if (need > p0cap) { p0cap = need * 2; P0 = (Path *)realloc(P0, p0cap * sizeof(Path)); }
            P0[np0].n[0] = t; P0[np0].n[1] = s; P0[np0].len = 2; np0++;
            for (int64_t a = aptr[s]; a < aptr[s + 1]; a++) {
                int32_t x = ax[a];
                P0[np0].n[0] = t; P0[np0].n[1] = s; P0[np0].n[2] = x; P0[np0].len = 3; np0++;
                for (int32_t y = 0; y < X->cnt[(size_t)x * 2 + 1]; y++) {
                    P0[np0].n[0] = t; P0[np0].n[1] = s; P0[np0].n[2] = x;
                    P0[np0].n[3] = X->col[((size_t)x * 2 + 1) * k + y]; P0[np0].len = 4; np0++;
                }
            }
            /* joined with TGT?  extender.py:72-81,175,177-178: t has attach, "T:" in t, s in knn_BB[t].keys(), "S:" in s */
            int joint = (aptr[t + 1] > aptr[t]) && X->cls[t] == 1 && knn_find(X, t, s) != NULL;
            int32_t path[6];
            double sp, cp;
            if (joint) { /* final_joint_extend  extender.py:142-169 */
                for (int64_t a = aptr[t]; a < aptr[t + 1]; a++) {
                    int32_t xp = ax[a];
                    if (want && !want[xp] && !wanty[xp]) continue;
                    if (!want || want[xp])
                    for (size_t p = 0; p < np0; p++) { /* target_path */
                        path[0] = xp; memcpy(path + 1, P0[p].n, (size_t)P0[p].len * sizeof(int32_t));
                        path_value(X, path, P0[p].len + 1, &sp, &cp);
                        pmap_add(&M, ((uint64_t)(uint32_t)xp << 32) | (uint32_t)P0[p].n[P0[p].len - 1], sp * cp, cp);
                        n_paths++;
                    }
                    for (size_t p = 0; p < np0; p++) /* longest_path */
                        for (int32_t y = 0; y < X->cnt[(size_t)xp * 2 + 1]; y++) {
                            int32_t yp = X->col[((size_t)xp * 2 + 1) * k + y];
                            if (want && !want[yp]) continue;
                            path[0] = yp; path[1] = xp; memcpy(path + 2, P0[p].n, (size_t)P0[p].len * sizeof(int32_t));
                            path_value(X, path, P0[p].len + 2, &sp, &cp);
                            pmap_add(&M, ((uint64_t)(uint32_t)yp << 32) | (uint32_t)P0[p].n[P0[p].len - 1], sp * cp, cp);
                            n_paths++;
                        }
                }
            }
            /* final_nonjoint_extend on EVERY source record (quirk A.5-5)  extender.py:124-140,:180 */
            if (!want || want[t])
            for (size_t p = 0; p < np0; p++) {
                path_value(X, P0[p].n, P0[p].len, &sp, &cp);
                pmap_add(&M, ((uint64_t)(uint32_t)t << 32) | (uint32_t)P0[p].n[P0[p].len - 1], sp * cp, cp);
                n_paths++;
            }
        }
    }
    free(P0); free(wanty); free(wantt);
    X->n_paths = n_paths;
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    X->path_seconds = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
    /* B6  extender.py:184-217: xsim = sum(s_p c_p)/sum(c_p), grouped by start; ends ascending (canonical) */
    uint64_t *keys = (uint64_t *)malloc((size_t)(M.n ? M.n : 1) * sizeof(uint64_t));
    uint64_t nk = 0;
    for (uint64_t i = 0; i < M.cap; i++) if (M.key[i] != UINT64_MAX) keys[nk++] = M.key[i];
    /* sort keys ascending (start major, end minor) */
    qsort(keys, (size_t)nk, sizeof(uint64_t), cmp_u64);
    X->xs_end = (int32_t *)malloc((size_t)(nk ? nk : 1) * sizeof(int32_t));
    X->xs_val = (double *)malloc((size_t)(nk ? nk : 1) * sizeof(double));
    for (uint64_t q = 0; q < nk; q++) {
        uint64_t key = keys[q];
        uint64_t h = pmap_hash(key) & (M.cap - 1);
        while (M.key[h] != key) h = (h + 1) & (M.cap - 1);
        int32_t st = (int32_t)(key >> 32);
        X->xs_ptr[st + 1]++;
        X->xs_end[q] = (int32_t)(key & 0xffffffffu);
        X->xs_val[q] = 1.0 * M.sc[h] / M.c[h];
    }
    for (int32_t i = 0; i < I; i++) X->xs_ptr[i + 1] += X->xs_ptr[i];
    free(keys); free(M.key); free(M.sc); free(M.scl); free(M.c); free(M.cl); free(aptr); free(ax);
    return X;
}
void xo_ext_free(XoExt *X) {
    if (!X) return;
    free(X->bb); free(X->cls); free(X->cnt); free(X->col); free(X->val);
    free(X->xs_ptr); free(X->xs_end); free(X->xs_val); free(X);
}

/* --------------------------------------------------------------- stage C */

/* C2/C3 candidate ranking: stable sort by -abs(xsim) of the (end ascending) candidate list,
 * keep `keep` (10 private generator.py:85, 4 non-private :109).  Writes min(keep, n) ends. */
static int rank_cands(const XoExt *X, int32_t start, int keep, int32_t *out) {
    int64_t lo = X->xs_ptr[start], n = X->xs_ptr[start + 1] - lo;
    int m = 0;
    for (int r = 0; r < keep && r < n; r++) {
        int64_t best = -1;
        for (int64_t t = 0; t < n; t++) {
            int taken = 0;
            for (int q = 0; q < m; q++) if (out[q] == X->xs_end[lo + t]) { taken = 1; break; }
            if (taken) continue;
            if (best < 0 || fabs(X->xs_val[lo + t]) > fabs(X->xs_val[lo + best])) best = t;
        }
        out[m++] = X->xs_end[lo + best];
    }
    return m;
}

/* C2-C4: generator.py:27-111 + utils/assist.py:210-215.
 *   private    : choice = argmax |xsim| (weighted pick degenerates under py3)
 *   nonprivate : choice = top4[picks[start]] where the caller drew picks[start] =
 *                np.random.randint(0, len(top)-1) in ascending start order (global RNG)
 *   M[choice] = start, last writer (largest start index) wins.
 * n_top[start] = min(4 or 10, #candidates) (0 if start has no candidates); choice[start] = -1 if none. */
void xo_select(const XoExt *X, int private_flag, const int32_t *picks, int32_t *n_top, int32_t *choice,
               int32_t *map_src2tgt /* [I] */) {
    int32_t I = X->I;
    int32_t top[10];
    for (int32_t i = 0; i < I; i++) map_src2tgt[i] = -1;
    for (int32_t st = 0; st < I; st++) {
        choice[st] = -1;
        int m = rank_cands(X, st, private_flag ? 10 : 4, top);
        n_top[st] = m;
        if (!m) continue;
        int idx = private_flag ? 0 : (picks ? picks[st] : 0);
        if (idx < 0 || idx >= m) idx = 0;
        choice[st] = top[idx];
        map_src2tgt[top[idx]] = st;
    }
}

typedef struct { int64_t n_rows; int32_t *user; int32_t *item; double *rating; int64_t *time; int64_t n_target_rows; int64_t n_profiles; } XoAlter;

/* C5: generator.py:113-157 build_alterEgo.
 * rows with "T:" in iid pass through (:156-157); mapped rows are grouped per user by new iid in
 * first-seen order with rating = np.mean, time = first row's time (:123-138).  Output order = the
 * reference's: all pass-through rows (trainRDD order), then all AlterEgo rows (user order). */
XoAlter *xo_alterego(int64_t U, const int64_t *ptr, const int32_t *item, const float *rating, const int64_t *time,
                     const uint8_t *flags, const int32_t *map_src2tgt) {
    int64_t nnz = ptr[U];
    XoAlter *A = (XoAlter *)calloc(1, sizeof(XoAlter));
    A->user = (int32_t *)malloc((size_t)(2 * nnz + 1) * sizeof(int32_t));
    A->item = (int32_t *)malloc((size_t)(2 * nnz + 1) * sizeof(int32_t));
    A->rating = (double *)malloc((size_t)(2 * nnz + 1) * sizeof(double));
    A->time = (int64_t *)malloc((size_t)(2 * nnz + 1) * sizeof(int64_t));
    int64_t n = 0;
    for (int64_t u = 0; u < U; u++)
        for (int64_t e = ptr[u]; e < ptr[u + 1]; e++)
            if (flags[item[e]] & 2) { A->user[n] = (int32_t)u; A->item[n] = item[e]; A->rating[n] = (double)rating[e]; A->time[n] = time[e]; n++; }
    A->n_target_rows = n;
    int64_t maxd = 1;
    for (int64_t u = 0; u < U; u++) if (ptr[u + 1] - ptr[u] > maxd) maxd = ptr[u + 1] - ptr[u];
    int32_t *gi = (int32_t *)malloc((size_t)maxd * sizeof(int32_t));
    double *gr = (double *)malloc((size_t)maxd * sizeof(double));
    uint8_t *used = (uint8_t *)malloc((size_t)maxd);
    uint8_t *seen_user = (uint8_t *)calloc((size_t)U, 1);
    for (int64_t i = 0; i < n; i++) seen_user[A->user[i]] = 1;
    for (int64_t u = 0; u < U; u++) {
        int64_t lo = ptr[u], d = ptr[u + 1] - lo;
        memset(used, 0, (size_t)d);
        for (int64_t a = 0; a < d; a++) {
            int32_t m = map_src2tgt[item[lo + a]];
            if (m < 0 || used[a]) continue;
            int64_t g = 0;
            for (int64_t b = a; b < d; b++)
                if (!used[b] && map_src2tgt[item[lo + b]] == m) { used[b] = 1; gr[g++] = (double)rating[lo + b]; }
            (void)gi;
            A->user[n] = (int32_t)u; A->item[n] = m; A->rating[n] = np_sum(gr, g) / (double)g; A->time[n] = time[lo + a]; n++;
            seen_user[u] = 1;
        }
    }
    A->n_rows = n;
    for (int64_t u = 0; u < U; u++) A->n_profiles += seen_user[u];
    free(gi); free(gr); free(used); free(seen_user);
    return A;
}
void xo_alter_free(XoAlter *A) {
    if (!A) return;
    free(A->user); free(A->item); free(A->rating); free(A->time); free(A);
}

/* ------------------------------------------------------------------ dense item-factor variant
 * BASELINE.json configs[4] ("no reference counterpart; oracle = the defining arithmetic").
 * PARITY UNPINNED against the reference by construction: the reference has no dense path, so there is nothing to
 * capture vectors from.  The arithmetic is defined here and the HIP path must reproduce it bit for bit:
 *   Fn[i][k] = (float)((double)F[i][k] / sqrt(sum_k (double)F[i][k]^2))   (sum in k order; all-zero row -> zeros)
 *   sim(t,s) = fmaf chain over k ascending in fp32, starting from +0
 *   top-k per target row by (|sim| desc, source index asc); unused entries idx -1, val 0. */
void xo_dense_normalize(int32_t n, int32_t K, const float *F, float *Fn) {
    for (int32_t i = 0; i < n; i++) {
        double q = 0.0;
        for (int32_t k = 0; k < K; k++) { double x = (double)F[(size_t)i * K + k]; q += x * x; }
        double nrm = sqrt(q);
        for (int32_t k = 0; k < K; k++)
            Fn[(size_t)i * K + k] = (nrm > 0.0) ? (float)((double)F[(size_t)i * K + k] / nrm) : 0.f;
    }
}

static int dense_better(float va, int32_t ia, float vb, int32_t ib) {
    float aa = fabsf(va), ab = fabsf(vb);
    return (aa > ab) || (aa == ab && ia < ib);
}

void xo_dense_topk(int32_t n_t, int32_t n_s, int32_t K, const float *Ft, const float *Fs, int32_t top_k,
                   int32_t *out_idx, float *out_val, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int32_t i = 0; i < n_t; i++) {
        int32_t *oi = out_idx + (size_t)i * top_k;
        float *ov = out_val + (size_t)i * top_k;
        int cnt = 0;
        for (int32_t j = 0; j < n_s; j++) {
            float acc = 0.f;
            for (int32_t k = 0; k < K; k++) acc = fmaf(Ft[(size_t)i * K + k], Fs[(size_t)j * K + k], acc);
            if (cnt == top_k && !dense_better(acc, j, ov[cnt - 1], oi[cnt - 1])) continue;
            int p = (cnt < top_k) ? cnt++ : top_k - 1;     /* sorted insertion */
            while (p > 0 && dense_better(acc, j, ov[p - 1], oi[p - 1])) { ov[p] = ov[p - 1]; oi[p] = oi[p - 1]; p--; }
            ov[p] = acc; oi[p] = j;
        }
        for (int p = cnt; p < top_k; p++) { oi[p] = -1; ov[p] = 0.f; }
    }
}

/* ------------------------------------------------------------------ RecommenderSim
 * core/recommenderSim.py:65-133,188-195: item-item cosine over the AlterEgo profile with significance weighting and
 * the per-pair leave-one-out LOCAL SENSITIVITY; both method names take the cosine branch (:190: "cosine_item" is a
 * substring of "adjust_cosine_item").  For every directed pair (i, j) with co-raters C:
 *   inner = sum_{u in C} r_ui r_uj            python sum() in co-rater order   -> canonical: exact sum, rounded once
 *   norm  = np.sqrt(np.sum(r^2)) per item     (:41, pairwise)                  -> canonical: exact sum, rounded once
 *   sim   = (cos(inner, nx ny) * min(n, cap)) / cap                            (:78-88,:124-126)
 *   ls    = max over u in C of |w(cos(inner - r_ui r_uj, sqrt((nx^2 - r_ui^2) ny^2)), n-1) - sim| and the same with
 *           sqrt(nx^2 (ny^2 - r_uj^2))                                         (:98-116); NaN propagates like np.max
 * Only users with >= 2 ratings form pairs (:73-74).  Nothing is filtered.  Canonical sums differ from the
 * reference's roundings by <= 1e-12 relative (tests assert rtol 1e-9 and the same NaN pattern). */
typedef struct { int32_t I; int64_t *row_ptr; int32_t *col; double *sim; double *ls; int32_t *nij; double *norm; } XoRec;

static double rec_weight(double cs, int32_t n, int cap) {
    int32_t mn = n < cap ? n : cap;
    return 1.0 * cs * (double)mn / (double)cap;
}

XoRec *xo_rec_sim(int cap, int64_t U, int32_t I, const int64_t *ptr, const int32_t *item, const double *rating) {
    int64_t nnz = ptr[U];
    XoRec *S = (XoRec *)calloc(1, sizeof(XoRec));
    S->I = I;
    S->norm = (double *)calloc((size_t)(I ? I : 1), sizeof(double));
    {   /* exact sum of the squares per item */
        double *hi = (double *)calloc((size_t)(I ? I : 1), sizeof(double)), *lo = (double *)calloc((size_t)(I ? I : 1), sizeof(double));
        for (int64_t e = 0; e < nnz; e++) { double r = rating[e]; dd_add(&hi[item[e]], &lo[item[e]], r * r); }
        for (int32_t i = 0; i < I; i++) S->norm[i] = sqrt(hi[i]);
        free(hi); free(lo);
    }
    int64_t *iptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    for (int64_t e = 0; e < nnz; e++) iptr[item[e] + 1]++;
    for (int32_t i = 0; i < I; i++) iptr[i + 1] += iptr[i];
    int32_t *iuser = (int32_t *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    int64_t *ient = (int64_t *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int64_t));   /* the rating's own profile entry */
    double *irat = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));   /* np.float64: AlterEgo ratings are means (generator.py:134) */
    {
        int64_t *cur = (int64_t *)malloc((size_t)(I ? I : 1) * sizeof(int64_t));
        memcpy(cur, iptr, (size_t)I * sizeof(int64_t));
        for (int64_t u = 0; u < U; u++)
            for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) { int64_t p = cur[item[e]]++; iuser[p] = (int32_t)u; irat[p] = rating[e]; ient[p] = e; }
        free(cur);
    }
    Arena A; memset(&A, 0, sizeof(A));
    double *als = NULL; int64_t als_cap = 0;
    S->row_ptr = (int64_t *)calloc((size_t)I + 1, sizeof(int64_t));
    int32_t *cnt = (int32_t *)calloc((size_t)(I ? I : 1), sizeof(int32_t));
    int32_t *pos = (int32_t *)calloc((size_t)(I ? I : 1), sizeof(int32_t));
    int32_t *touched = (int32_t *)malloc((size_t)(I ? I : 1) * sizeof(int32_t));
    int64_t bufcap = 1024;
    int32_t *bj = (int32_t *)malloc(bufcap * sizeof(int32_t));
    double *b0 = (double *)malloc(bufcap * sizeof(double)), *b1 = (double *)malloc(bufcap * sizeof(double));
    double *g0 = (double *)malloc(bufcap * sizeof(double)), *g1 = (double *)malloc(bufcap * sizeof(double));
    for (int32_t i = 0; i < I; i++) {
        int64_t w = 0, nt = 0;
        for (int64_t p = iptr[i]; p < iptr[i + 1]; p++) {
            int32_t u = iuser[p];
            if (ptr[u + 1] - ptr[u] < 2) continue;
            for (int64_t e = ptr[u]; e < ptr[u + 1]; e++) {
                int32_t j = item[e];
                /* combinations(ratings, 2) pairs every two ENTRIES of a profile (:71): an AlterEgo profile can hold
                 * an item twice (a pass-through rating and a mapped one), which yields (X, X) keys and two co-rating
                 * entries per such user; only the entry itself is skipped */
                if (e == ient[p]) continue;
                if (w == bufcap) {
                    bufcap *= 2;
                    bj = (int32_t *)realloc(bj, bufcap * sizeof(int32_t));
                    b0 = (double *)realloc(b0, bufcap * sizeof(double)); b1 = (double *)realloc(b1, bufcap * sizeof(double));
                    g0 = (double *)realloc(g0, bufcap * sizeof(double)); g1 = (double *)realloc(g1, bufcap * sizeof(double));
                }
                bj[w] = j; b0[w] = irat[p]; b1[w] = rating[e]; w++;
                if (cnt[j]++ == 0) touched[nt++] = j;
            }
        }
        qsort(touched, (size_t)nt, sizeof(int32_t), cmp_i32);
        int64_t off = 0;
        for (int64_t t = 0; t < nt; t++) { pos[touched[t]] = (int32_t)off; off += cnt[touched[t]]; }
        for (int64_t k = 0; k < w; k++) { int32_t q = pos[bj[k]]++; g0[q] = b0[k]; g1[q] = b1[k]; }
        off = 0;
        const double nx = S->norm[i];
        for (int64_t t = 0; t < nt; t++) {
            int32_t j = touched[t], n = cnt[j];
            const double *r0 = g0 + off, *r1 = g1 + off;
            off += n;
            double hi = 0.0, lo = 0.0;
            for (int32_t k = 0; k < n; k++) dd_add(&hi, &lo, r0[k] * r1[k]);
            const double inner = hi, ny = S->norm[j];
            const double np_ = nx * ny;
            const double sim = rec_weight(np_ ? 1.0 * inner / np_ : 0.0, n, cap);
            double ls = 0.0;
            int isnan_ls = 0;
            for (int32_t k = 0; k < n; k++) {
                const double rest = inner - r0[k] * r1[k];
                const double m1 = sqrt((nx * nx - r0[k] * r0[k]) * (ny * ny));
                const double m2 = sqrt((nx * nx) * (ny * ny - r1[k] * r1[k]));
                const double d1 = fabs(rec_weight(m1 != 0.0 ? 1.0 * rest / m1 : 0.0, n - 1, cap) - sim);   /* NaN != 0 */
                const double d2 = fabs(rec_weight(m2 != 0.0 ? 1.0 * rest / m2 : 0.0, n - 1, cap) - sim);
                if (d1 != d1 || d2 != d2) isnan_ls = 1;
                if (d1 > ls) ls = d1;
                if (d2 > ls) ls = d2;
            }
            if (isnan_ls) ls = NAN;
            arena_push(&A, j, sim, 0, n);
            if (A.n > als_cap) { als_cap = A.n * 2 + 1024; als = (double *)realloc(als, (size_t)als_cap * sizeof(double)); }
            als[A.n - 1] = ls;
            cnt[j] = 0; pos[j] = 0;
        }
        S->row_ptr[i + 1] = A.n;
    }
    S->col = A.col; S->sim = A.sim; S->nij = A.nij; S->ls = als;
    free(A.mutu);
    free(cnt); free(pos); free(touched); free(bj); free(b0); free(b1); free(g0); free(g1); free(iptr); free(iuser); free(irat); free(ient);
    return S;
}
void xo_rec_free(XoRec *S) {
    if (!S) return;
    free(S->row_ptr); free(S->col); free(S->sim); free(S->ls); free(S->nij); free(S->norm); free(S);
}

/* RecommenderPrivacy.nonprivate_neighbor_selection (core/recommenderPrivacy.py:22-35,141-152) on the rows of
 * xo_rec_sim: per item the `keep` neighbours with the largest |sim|; equal similarities in ascending neighbour index
 * (the reference's stable sort keeps their arrival order, which Spark does not define). */
typedef struct { double a; int32_t c; int64_t p; } RecEnt;
static int cmp_rec(const void *x, const void *y) {
    const RecEnt *a = (const RecEnt *)x, *b = (const RecEnt *)y;
    if (a->a > b->a) return -1;
    if (a->a < b->a) return 1;
    return (a->c > b->c) - (a->c < b->c);
}
void xo_rec_select(const XoRec *S, int32_t keep, int32_t *out_cnt, int32_t *out_col, double *out_sim, double *out_ls) {
    int64_t maxlen = 1;
    for (int32_t i = 0; i < S->I; i++) { int64_t l = S->row_ptr[i + 1] - S->row_ptr[i]; if (l > maxlen) maxlen = l; }
    RecEnt *e = (RecEnt *)malloc((size_t)maxlen * sizeof(RecEnt));
    for (int32_t i = 0; i < S->I; i++) {
        int64_t lo = S->row_ptr[i], n = S->row_ptr[i + 1] - lo;
        for (int64_t t = 0; t < n; t++) { e[t].a = fabs(S->sim[lo + t]); e[t].c = S->col[lo + t]; e[t].p = lo + t; }
        qsort(e, (size_t)n, sizeof(RecEnt), cmp_rec);
        int32_t c = (int32_t)(n < keep ? n : keep);
        out_cnt[i] = c;
        for (int32_t t = 0; t < keep; t++) {
            size_t o = (size_t)i * keep + t;
            out_col[o] = t < c ? e[t].c : -1;
            out_sim[o] = t < c ? S->sim[e[t].p] : 0.0;
            out_ls[o] = t < c ? S->ls[e[t].p] : 0.0;
        }
    }
    free(e);
}
