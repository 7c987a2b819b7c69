"""Seeded synthetic Amazon-format two-domain ratings (build-owned).

The reference ships no data (code/data/README.md:1-9); its input format is
`uid \\t iid \\t rating \\t unix_ts` (README.md:41-42) and, after the clean stage,
`trainRDD[(uid, [(iid + label, rating, datetime)*])]` (baselinerClean.py:49-52).

Shape of the generator (parameters from SURVEY.md §8d / BASELINE.md §3):
log-normal profile lengths per domain (d_min + floor(LogNormal(mu, sigma))),
Zipf(0.8) item popularity, ratings {1..5} with p = (.05,.05,.10,.25,.55),
timestamps uniform in 2012-2013, a fraction `overlap` of users active in both
domains.  Source raw ids are 10 digits starting with "00" (ISBN-like), target raw
ids start with "B0" (ASIN-like) so that the reference's prefix-based cross-domain
label (baselinerSim.py:189-191) coincides with the domain.

Items are indexed in lexicographic order of their id strings (source block first,
then target block; zero-padded numbers keep numeric = lexicographic order), which
is the canonical tie-break order of SURVEY.md Appendix B.
"""
import numpy as np

T0 = 1325376000  # 2012-01-01 00:00:00 UTC
T1 = 1388534399  # 2013-12-31 23:59:59 UTC
RATING_P = np.array([.05, .05, .10, .25, .55])


class Ratings(object):
    """Index-space ratings: CSR by user in trainRDD order."""

    def __init__(self, user_ptr, item, rating, time, n_items, n_src_items,
                 src_numbers, tgt_numbers):
        self.user_ptr = user_ptr          # int64 [U+1]
        self.item = item                  # int32 [nnz]   lexicographic item index
        self.rating = rating              # float32 [nnz]
        self.time = time                  # int64 [nnz]   unix seconds
        self.n_items = int(n_items)
        self.n_src_items = int(n_src_items)
        self.src_numbers = src_numbers    # raw id numbers of the source items kept
        self.tgt_numbers = tgt_numbers

    @property
    def n_users(self):
        return len(self.user_ptr) - 1

    @property
    def nnz(self):
        return len(self.item)

    # --- per-item attribute arrays the engine needs (see xmap.engine.ids) ---
    def item_attrs(self):
        I, Is = self.n_items, self.n_src_items
        prefix_cls = np.zeros(I, np.int32)
        prefix_cls[Is:] = 1
        suffix_cls = prefix_cls.copy()
        contains_mask = np.where(prefix_cls == 0, 1, 2).astype(np.uint32)
        flags = np.where(prefix_cls == 0, 1, 2).astype(np.uint8)
        return prefix_cls, suffix_cls, contains_mask, flags

    # --- string views (small cases only) ---
    def item_ids(self):
        return (["%010dS:" % n for n in self.src_numbers] +
                ["B0%08dT:" % n for n in self.tgt_numbers])

    def user_ids(self):
        return ["A%013d" % u for u in range(self.n_users)]

    def train_records(self):
        """[(uid, [(iid, rating, unix_ts)*])*] in trainRDD order."""
        iids = self.item_ids()
        uids = self.user_ids()
        out = []
        for u in range(self.n_users):
            a, b = int(self.user_ptr[u]), int(self.user_ptr[u + 1])
            out.append((uids[u], [(iids[self.item[e]], float(self.rating[e]),
                                   int(self.time[e])) for e in range(a, b)]))
        return out


def _domain_draws(rng, users, n_items, d_min, mu, sigma, zipf):
    """(user, item-number) draws for one domain, de-duplicated per user keeping
    first occurrences in generated order."""
    n = len(users)
    d = d_min + np.floor(rng.lognormal(mu, sigma, n)).astype(np.int64)
    d = np.minimum(d, n_items)
    w = np.arange(1, n_items + 1, dtype=np.float64) ** (-zipf)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    tot = int(d.sum())
    ranks = np.searchsorted(cdf, rng.random(tot), side="right")
    ranks = np.minimum(ranks, n_items - 1)
    perm = rng.permutation(n_items)            # popularity rank -> item number
    items = perm[ranks].astype(np.int64)
    u = np.repeat(users.astype(np.int64), d)
    key = u * n_items + items
    _, first = np.unique(key, return_index=True)
    first.sort()
    return u[first], items[first]


def make_two_domain(seed, n_users, n_src_items, n_tgt_items, overlap=0.25,
                    d_min=5, mu=0.7, sigma=1.0, zipf=0.8, src_share=0.5):
    """src_share: fraction of the single-domain users that are source-only (0.5: as many as target-only)"""
    rng = np.random.default_rng(seed)
    r = rng.random(n_users)
    both = r < overlap
    src_only = (~both) & (r < overlap + (1.0 - overlap) * src_share)
    tgt_only = ~(both | src_only)
    all_u = np.arange(n_users)
    us, its = _domain_draws(rng, all_u[both | src_only], n_src_items,
                            d_min, mu, sigma, zipf)
    ut, itt = _domain_draws(rng, all_u[both | tgt_only], n_tgt_items,
                            d_min, mu, sigma, zipf)
    # compact item numbers to the ones present; lexicographic index = source
    # block (ascending number) then target block (ascending number)
    src_numbers, its_c = np.unique(its, return_inverse=True)
    tgt_numbers, itt_c = np.unique(itt, return_inverse=True)
    n_src = len(src_numbers)
    u = np.concatenate([us, ut])
    it = np.concatenate([its_c, itt_c + n_src]).astype(np.int32)
    order = np.argsort(u, kind="stable")       # source entries first per user
    u, it = u[order], it[order]
    nnz = len(u)
    rating = (rng.choice(5, size=nnz, p=RATING_P) + 1).astype(np.float32)
    time = rng.integers(T0, T1 + 1, size=nnz, dtype=np.int64)
    cnt = np.bincount(u, minlength=n_users)
    ptr = np.zeros(n_users + 1, np.int64)
    np.cumsum(cnt, out=ptr[1:])
    return Ratings(ptr, it, rating, time, n_src + len(tgt_numbers), n_src,
                   src_numbers, tgt_numbers)


def make_multi_domain(seed, n_users, n_src_items, n_tgt_items, n_sources, overlap=0.25,
                      d_min=5, mu=0.7, sigma=1.0, zipf=0.8):
    """N source domains against ONE target domain (BASELINE configs[3] shape; reference multidomain_demo.py:101-128 runs
    every source as an independent two-domain problem against the same target).  Returns one Ratings per source: its
    source entries + the shared target entries; users and target item numbers are common to all of them."""
    rng = np.random.default_rng(seed)
    all_u = np.arange(n_users)
    in_tgt = rng.random(n_users) < 0.5 + overlap / 2
    ut, itt = _domain_draws(rng, all_u[in_tgt], n_tgt_items, d_min, mu, sigma, zipf)
    tgt_numbers, itt_c = np.unique(itt, return_inverse=True)
    rt = (rng.choice(5, size=len(ut), p=RATING_P) + 1).astype(np.float32)
    tt = rng.integers(T0, T1 + 1, size=len(ut), dtype=np.int64)
    out = []
    for d in range(n_sources):
        r = rng.random(n_users)
        in_src = np.where(in_tgt, r < 2 * overlap, r < 0.6)         # a share of the target's users, plus source-only ones
        us, its = _domain_draws(rng, all_u[in_src], n_src_items, d_min, mu, sigma, zipf)
        src_numbers, its_c = np.unique(its, return_inverse=True)
        n_src = len(src_numbers)
        rs = (rng.choice(5, size=len(us), p=RATING_P) + 1).astype(np.float32)
        ts = rng.integers(T0, T1 + 1, size=len(us), dtype=np.int64)
        u = np.concatenate([us, ut])
        it = np.concatenate([its_c, itt_c + n_src]).astype(np.int32)
        order = np.argsort(u, kind="stable")       # source entries first per user
        cnt = np.bincount(u, minlength=n_users)
        ptr = np.zeros(n_users + 1, np.int64)
        np.cumsum(cnt, out=ptr[1:])
        out.append(Ratings(ptr, it[order], np.concatenate([rs, rt])[order], np.concatenate([ts, tt])[order],
                           n_src + len(tgt_numbers), n_src, src_numbers, tgt_numbers))
    return out


# named workloads (BASELINE.json configs)
def config_c1(seed=1):
    """10k users / 2x5k items (BASELINE configs[0])."""
    return make_two_domain(seed, 10000, 5000, 5000)


def config_c4(seed=4, n_sources=4):
    """4 source domains -> 1 target, ~5M users in total (BASELINE configs[3]): every source problem has 1.25 M users."""
    return make_multi_domain(seed, 1250000, 200000, 200000, n_sources)


def config_s1(seed=3):
    """The shape of the reference's own large scenario S1 (TechReport_XMap.pdf, Tables 3 and 5): source = movies, 128 402
    items / 473 764 users; target = books, 403 234 items / 725 846 users; the two catalogues barely share users (rating
    density of the union 0.0147 %).  Same generator as the other configs (log-normal profile lengths, Zipf popularity), 3 %
    of the users active in both domains: the bridge set is thin and most items are non-bridge records -- the regime the
    stage-B middle lists have to survive (n_nb far above one LDS span of columns)."""
    n_src_users, n_tgt_users, overlap = 473764, 725846, 0.03
    n_users = int(round((n_src_users + n_tgt_users) / (1.0 + overlap)))
    both = overlap * n_users
    src_share = (n_src_users - both) / max(n_users - both, 1.0)
    return make_two_domain(seed, n_users, 128402, 403234, overlap=overlap, src_share=src_share)


def config_c2(seed=2):
    """~1M users / 200k+200k items (BASELINE configs[1])."""
    return make_two_domain(seed, 1000000, 200000, 200000)
