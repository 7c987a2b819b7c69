// feeder.hip -- native feeder of the hot path (host code only; SURVEY.md 8f-1: "text parse -> CSR is also the H2D feeder").
//
// Raw Amazon-format lines `uid iid rating unix_ts` (reference README.md:41-42) -> what the device engine uploads: user and
// item id tables, CSR by user (item index, rating, time), the four per-item predicate arrays -- with the semantics of the
// reference's clean stage in between (core/baselinerClean.py):
//   parse_line    :40-54   fields = re.split(r"\s+", line); kept when the LOCAL-time year of the timestamp lies in
//                          [year_from, year_to]; item id = field 1 + domain label; rating = float(field 2)
//   remove_invalid:64-87   per user ONE rating per item: a strictly later one replaces an earlier one in place
//                          (first-seen item order is kept); users in first-seen order (aggregateByKey on one partition)
//   clean_data    :98-101  users with fewer than num_atleast_rating ratings are dropped
// then the engine's own conventions (xmap/engine/ids.py): items indexed in lexicographic id order, predicates
// prefix_cls = class of iid[:2], suffix_cls = class of iid[-2:] (classes numbered in first-seen order over the sorted
// ids), contains_mask bit c = "label c is a substring of the id", flags = ("S:" in id) | ("T:" in id) << 1.
// Python did this with dict / list operations per rating: 15 s for the 10^7 ratings of BASELINE configs[1]; this is
// one pass over the text with two hash tables.  ASCII whitespace only (Python's \s also knows the Unicode separators).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace xmap {
namespace {

struct Entry { int32_t item; double rating; double when; };

struct Feed {
    std::vector<std::string> uids, iids;
    std::vector<int64_t> ptr;
    std::vector<int32_t> item;
    std::vector<double> rating, when;
    std::vector<int32_t> prefix_cls, suffix_cls;
    std::vector<uint32_t> contains;
    std::vector<uint8_t> flags;
    int64_t n_lines = 0, n_in_period = 0;
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

struct Rec { std::string_view uid, iid; double rating, when; uint64_t uh, ih; };     // one line inside the period

inline uint64_t hash_bytes(std::string_view s) {          // FNV-1a, finished with a multiply-shift mix
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    return h;
}

// open-addressing table string -> dense index (first-seen order); the keys are views into the text being parsed
struct StrMap {
    struct Slot { uint64_t h; const char *p; uint32_t len; int32_t idx; };
    std::vector<Slot> t;
    size_t n = 0, mask;
    explicit StrMap(size_t cap) : t(cap, Slot{0, nullptr, 0, -1}), mask(cap - 1) {}
    void grow() {
        std::vector<Slot> old;
        old.swap(t);
        t.assign(old.size() * 2, Slot{0, nullptr, 0, -1});
        mask = t.size() - 1;
        for (const Slot &s : old)
            if (s.idx >= 0) {
                size_t i = s.h & mask;
                while (t[i].idx >= 0) i = (i + 1) & mask;
                t[i] = s;
            }
    }
    int32_t find(std::string_view key, uint64_t h) const {
        size_t i = h & mask;
        for (;;) {
            const Slot &s = t[i];
            if (s.idx < 0) return -1;
            if (s.h == h && s.len == key.size() && memcmp(s.p, key.data(), key.size()) == 0) return s.idx;
            i = (i + 1) & mask;
        }
    }
    int32_t get(std::string_view key, uint64_t h, int32_t next, bool &fresh) {
        if ((n + 1) * 10 > t.size() * 7) grow();
        size_t i = h & mask;
        for (;;) {
            Slot &s = t[i];
            if (s.idx < 0) { s = Slot{h, key.data(), (uint32_t)key.size(), next}; n++; fresh = true; return next; }
            if (s.h == h && s.len == key.size() && memcmp(s.p, key.data(), key.size()) == 0) { fresh = false; return s.idx; }
            i = (i + 1) & mask;
        }
    }
};

// float(token): the whole token must parse.  Plain decimals (what rating and timestamp columns hold) are converted here --
// digits accumulate exactly in 64 bits and one division by a power of ten that is itself exact rounds correctly, the same
// value strtod gives --; everything else (exponents, inf / nan, more than 18 digits) goes to strtod
bool to_double(std::string_view t, double &v) {
    static const double P10[19] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18};
    {
        size_t i = 0;
        const bool neg = !t.empty() && t[0] == '-';
        if (neg || (!t.empty() && t[0] == '+')) i = 1;
        uint64_t m = 0;
        int nd = 0, frac = 0;
        bool dot = false, plain = i < t.size();
        for (; i < t.size(); i++) {
            const char c = t[i];
            if (c >= '0' && c <= '9') { m = m * 10 + (uint64_t)(c - '0'); nd++; if (dot) frac++; }
            else if (c == '.' && !dot) dot = true;
            else { plain = false; break; }
        }
        if (plain && nd >= 1 && nd <= 15 && frac <= 15) {        // m < 2^53 and 10^frac exact: one correctly rounded division
            const double x = (double)m / P10[frac];
            v = neg ? -x : x;
            return true;
        }
    }
    // the rest as Python's float() reads it: decimal digits with single underscores BETWEEN digits ("1_0" is 10.0), one
    // '.', an exponent, a sign -- or inf / infinity / nan in any case; no hex floats, no "nan(...)" (strtod takes both)
    if (t.empty() || t.size() > 63) return false;
    char buf[64];
    size_t n = 0, i = 0;
    if (t[0] == '+' || t[0] == '-') buf[n++] = t[i++];
    auto word = [&](const char *w) {
        const size_t L = strlen(w);
        if (t.size() - i != L) return false;
        for (size_t k = 0; k < L; k++) if ((t[i + k] | 0x20) != w[k]) return false;
        return true;
    };
    if (word("inf") || word("infinity") || word("nan")) {
        memcpy(buf + n, t.data() + i, t.size() - i);
        n += t.size() - i;
    } else {
        for (; i < t.size(); i++) {
            const char c = t[i];
            const bool dig = c >= '0' && c <= '9';
            if (c == '_') {
                const bool before = i > 0 && t[i - 1] >= '0' && t[i - 1] <= '9';
                const bool after = i + 1 < t.size() && t[i + 1] >= '0' && t[i + 1] <= '9';
                if (!before || !after) return false;
                continue;
            }
            if (!(dig || c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-')) return false;
            buf[n++] = c;
        }
    }
    buf[n] = 0;
    char *end = nullptr;
    v = strtod(buf, &end);
    return n > 0 && end == buf + n;
}

void predicates(Feed &F) {
    const size_t n = F.iids.size();
    F.prefix_cls.resize(n); F.suffix_cls.resize(n); F.contains.assign(n, 0u); F.flags.assign(n, 0);
    std::unordered_map<std::string, int> pre, suf;
    std::vector<std::string> labels;
    for (size_t k = 0; k < n; k++) {
        const std::string &s = F.iids[k];
        const std::string p = s.substr(0, 2), q = s.size() >= 2 ? s.substr(s.size() - 2) : s;
        F.prefix_cls[k] = pre.emplace(p, (int)pre.size()).first->second;
        auto it = suf.emplace(q, (int)suf.size());
        if (it.second) labels.push_back(q);
        F.suffix_cls[k] = it.first->second;
    }
    for (size_t k = 0; k < n; k++) {
        const std::string &s = F.iids[k];
        uint32_t m = 0;
        for (size_t c = 0; c < labels.size() && c < 32; c++)
            if (s.find(labels[c]) != std::string::npos) m |= 1u << c;
        F.contains[k] = m;
        F.flags[k] = (uint8_t)((s.find("S:") != std::string::npos ? 1 : 0) | (s.find("T:") != std::string::npos ? 2 : 0));
    }
}

}  // namespace
}  // namespace xmap

using namespace xmap;

extern "C" {

struct xmap_feed { Feed F; };

/* n_parts texts (the domains of one problem, each with its label) -> ONE feed: users in first-seen order over the parts in
 * the order given, a user's entries of an earlier part in front of those of a later one (= xmap_feed_merge of the parts'
 * feeds, without building them) */
static int feed_texts_impl(int32_t n_parts, const char *const *texts, const int64_t *lens, const char *const *labels,
                           int32_t year_from, int32_t year_to, int32_t min_ratings, xmap_feed **out) {
    std::unique_ptr<xmap_feed> H(new xmap_feed());
    Feed &F = H->F;
    // the period as a range of seconds: local-time year in [year_from, year_to]  <=>  t0 <= floor(t) < t1 (one mktime per
    // bound instead of one localtime per line)
    auto year_start = [](int y) {
        struct tm b;
        memset(&b, 0, sizeof(b));
        b.tm_year = y - 1900; b.tm_mon = 0; b.tm_mday = 1; b.tm_isdst = -1;
        return (double)mktime(&b);
    };
    const double t0 = year_start(year_from), t1 = year_start(year_to + 1);
    const bool verbose = getenv("XMAP_FEED_VERBOSE") != nullptr;
    struct timespec ta, tb;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    auto lap = [&](const char *what) {
        clock_gettime(CLOCK_MONOTONIC, &tb);
        if (verbose) fprintf(stderr, "feeder: %s %.3f s\n", what, (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec));
        ta = tb;
    };
    // phase 1 (threads): lines -> records (views into the text, numbers, hashes of the two ids); a work item = a piece of a
    // part cut at line ends
    int nth = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("XMAP_FEED_THREADS")) nth = atoi(e);
    if (nth < 1) nth = 1;
    if (nth > 32) nth = 32;
    struct Piece { int part; int64_t lo, hi; };
    std::vector<Piece> pieces;
    for (int q = 0; q < n_parts; q++) {
        const char *text = texts[q];
        const int64_t len = lens[q];
        const int np = len < (1 << 20) ? 1 : nth;
        int64_t prev = 0;
        for (int k = 1; k <= np; k++) {
            int64_t c = (k == np) ? len : len / np * k;
            if (c < prev) c = prev;
            while (c < len && text[c] != '\n') c++;
            if (c < len) c++;
            if (c > prev) pieces.push_back(Piece{q, prev, c});
            prev = c;
        }
    }
    const int npc = (int)pieces.size();
    std::vector<std::vector<Rec>> parts(npc);
    std::vector<int64_t> n_lines(npc, 0), bad_line(npc, -1);
    std::vector<int> bad_kind(npc, 0);
    auto parse_piece = [&](int k) {
        const char *text = texts[pieces[k].part];
        const int64_t len = lens[pieces[k].part];
        std::vector<Rec> &out_ = parts[k];
        out_.reserve((size_t)((pieces[k].hi - pieces[k].lo) / 32 + 16));
        int64_t pos = pieces[k].lo, ln = 0;
        while (pos < pieces[k].hi) {
            int64_t eol = pos;
            while (eol < len && text[eol] != '\n') eol++;
            ln++;
            // re.split(r"\s+", line): the fields between runs of whitespace; a leading run leaves an empty first field
            std::string_view tok[4];
            int nt = 0;
            int64_t p = pos;
            if (p < eol && is_ws(text[p])) { tok[nt++] = std::string_view(); while (p < eol && is_ws(text[p])) p++; }
            while (nt < 4) {
                const int64_t b0 = p;
                while (p < eol && !is_ws(text[p])) p++;
                tok[nt++] = std::string_view(text + b0, (size_t)(p - b0));      // (a trailing run leaves an empty last field)
                if (p >= eol) break;
                while (p < eol && is_ws(text[p])) p++;
            }
            if (nt < 4 || eol == pos) { bad_line[k] = ln; bad_kind[k] = 1; return; }      // [''] / short line: IndexError
            // parse_line: the timestamp of every line is converted, the rating only of the lines inside the period
            // (baselinerClean.py:46-52: float(rating) sits behind the year test -- a bad rating outside the period is skipped)
            double t, r = 0.0;
            if (!to_double(tok[3], t)) { bad_line[k] = ln; bad_kind[k] = 2; return; }
            n_lines[k]++;
            const double fl = floor(t);
            if (fl >= t0 && fl < t1) {
                if (!to_double(tok[2], r)) { bad_line[k] = ln; bad_kind[k] = 2; return; }
                Rec q;
                q.uid = tok[0]; q.iid = tok[1]; q.rating = r; q.when = t; q.uh = hash_bytes(tok[0]); q.ih = hash_bytes(tok[1]);
                out_.push_back(q);
            }
            pos = eol + 1;
        }
    };
    {
        std::atomic<int> next(0);
        auto worker = [&]() {
            for (int k = next++; k < npc; k = next++) {
                try { parse_piece(k); }
                catch (...) { bad_line[k] = 0; bad_kind[k] = 3; }      // (out of memory inside a piece: reported below)
            }
        };
        std::vector<std::thread> th;
        for (int k = 1; k < nth && k < npc; k++) th.emplace_back(worker);
        worker();
        for (std::thread &x : th) x.join();
    }
    {
        int64_t before = 0;
        int part = -1;
        for (int k = 0; k < npc; k++) {
            if (pieces[k].part != part) { part = pieces[k].part; before = 0; }
            if (bad_line[k] >= 0) {
                const long long ln = (long long)(before + bad_line[k]);
                if (bad_kind[k] == 3) { set_error("feeder: out of host memory while parsing text %d", part); return XMAP_ERR_CAPACITY; }
                if (bad_kind[k] == 1) set_error("text %d, line %lld has fewer than 4 fields", part, ln);
                else set_error("text %d, line %lld: rating / timestamp is not a number", part, ln);
                return XMAP_ERR_ARG;
            }
            before += n_lines[k];
            F.n_lines += n_lines[k];
        }
    }
    lap("phase 1 (parse)");
    // phase 2 (one thread: first-seen orders are sequential by nature), part by part: dense user and item indices, one
    // rating per (user, item) -- the latest wins in place (strictly later), first-seen item order --, the part's runs
    // (user, entries) in first-seen user order.  Files are usually grouped by user: then a user's records are one run and
    // nothing is sorted; otherwise the part's records are grouped by a stable counting sort first.
    size_t n_rec = 0;
    for (const std::vector<Rec> &v : parts) n_rec += v.size();
    F.n_in_period = (int64_t)n_rec;
    StrMap umap(1 << 16);
    std::vector<std::string_view> unames, inames;              // first-seen order
    std::vector<int8_t> ipart;                                  // part (label) of an item
    struct Run { int32_t user; int64_t lo, hi; };
    std::vector<std::vector<Run>> runs(n_parts);
    std::vector<Entry> ent;
    ent.reserve(n_rec);
    struct Tmp { int32_t u, it; double rating, when; };
    for (int q = 0; q < n_parts; q++) {
        StrMap imap(1 << 16);
        std::vector<Tmp> tmp;
        std::vector<int32_t> first_in_part;                     // global users in the part's first-seen order
        std::vector<int32_t> local;                             // global user -> index in first_in_part, -1
        local.assign(unames.size(), -1);
        std::string_view last_uid;
        int32_t last_u = -1;
        bool grouped = true;
        size_t cnt_part = 0;
        for (int k = 0; k < npc; k++) if (pieces[k].part == q) cnt_part += parts[k].size();
        tmp.reserve(cnt_part);
        for (int k = 0; k < npc; k++) {
            if (pieces[k].part != q) continue;
            for (const Rec &r : parts[k]) {
                int32_t u;
                if (last_u >= 0 && r.uid == last_uid) u = last_u;
                else {
                    bool fresh;
                    u = umap.get(r.uid, r.uh, (int32_t)unames.size(), fresh);
                    if (fresh) { unames.push_back(r.uid); local.push_back(-1); }
                    if (local[u] < 0) { local[u] = (int32_t)first_in_part.size(); first_in_part.push_back(u); }
                    else grouped = false;                       // the user was met before in this part: not one run
                    last_uid = r.uid; last_u = u;
                }
                bool fresh;
                const int32_t it = imap.get(r.iid, r.ih, (int32_t)inames.size(), fresh);
                if (fresh) { inames.push_back(r.iid); ipart.push_back((int8_t)q); }
                tmp.push_back(Tmp{u, it, r.rating, r.when});
            }
            std::vector<Rec>().swap(parts[k]);
        }
        if (!grouped) {                                         // stable counting sort by the part's first-seen user order
            std::vector<int64_t> at(first_in_part.size() + 1, 0);
            for (const Tmp &t : tmp) at[local[t.u] + 1]++;
            for (size_t x = 0; x < first_in_part.size(); x++) at[x + 1] += at[x];
            std::vector<Tmp> tmp2(tmp.size());
            for (const Tmp &t : tmp) tmp2[at[local[t.u]]++] = t;
            tmp.swap(tmp2);
        }
        // one rating per (user, item): where the current user's entry of an item sits (-1: none yet) -- O(1) per rating
        // whatever the length of the profile (a scan of the user's entries per rating made a crawler account of 10^5
        // ratings 5*10^9 comparisons); the table is reset through the entries the user touched
        std::vector<int64_t> at_item(inames.size(), -1);
        size_t y = 0;
        while (y < tmp.size()) {
            const int32_t u = tmp[y].u;
            const size_t first = ent.size();
            for (; y < tmp.size() && tmp[y].u == u; y++) {
                const Tmp &t = tmp[y];
                const int64_t z = at_item[t.it];
                if (z >= 0) {
                    if (t.when > ent[z].when) { ent[z].rating = t.rating; ent[z].when = t.when; }
                } else {
                    at_item[t.it] = (int64_t)ent.size();
                    ent.push_back(Entry{t.it, t.rating, t.when});
                }
            }
            for (size_t z = first; z < ent.size(); z++) at_item[ent[z].item] = -1;
            runs[q].push_back(Run{u, (int64_t)first, (int64_t)ent.size()});
        }
    }
    lap("phase 2 (dictionaries, latest rating)");
    // clean_data per part (the reference cleans every domain on its own), then the order of xmap_feed_merge: the users a
    // part keeps in that part's order, parts in the order given; a user's kept runs of the parts one after the other
    const size_t NU = unames.size();
    std::vector<int32_t> final_of(NU, -1), order_u;
    std::vector<std::vector<int32_t>> run_of(n_parts);
    std::vector<char> used(inames.size(), 0);
    size_t nnz = 0;
    for (int q = 0; q < n_parts; q++) {
        run_of[q].assign(NU, -1);
        for (size_t x = 0; x < runs[q].size(); x++) {
            const Run &R_ = runs[q][x];
            if (R_.hi - R_.lo < (int64_t)min_ratings) continue;
            run_of[q][R_.user] = (int32_t)x;
            if (final_of[R_.user] < 0) { final_of[R_.user] = (int32_t)order_u.size(); order_u.push_back(R_.user); }
            for (int64_t e = R_.lo; e < R_.hi; e++) used[ent[e].item] = 1;
            nnz += (size_t)(R_.hi - R_.lo);
        }
    }
    // items in lexicographic order of id + label
    std::vector<int32_t> order;
    for (size_t i = 0; i < inames.size(); i++) if (used[i]) order.push_back((int32_t)i);
    std::vector<std::string> labs;
    for (int q = 0; q < n_parts; q++) labs.emplace_back(labels[q]);
    auto less = [&](int32_t a_, int32_t b_) {
        const std::string_view x = inames[a_], y = inames[b_];
        if (x.size() == y.size() && ipart[a_] == ipart[b_]) return memcmp(x.data(), y.data(), x.size()) < 0;
        const size_t m = x.size() < y.size() ? x.size() : y.size();
        const int c = memcmp(x.data(), y.data(), m);
        if (c) return c < 0;
        return (std::string(x) + labs[ipart[a_]]) < (std::string(y) + labs[ipart[b_]]);
    };
    std::sort(order.begin(), order.end(), less);
    std::vector<int32_t> remap(inames.size(), -1);
    F.iids.reserve(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        remap[order[k]] = (int32_t)k;
        F.iids.push_back(std::string(inames[order[k]]) + labs[ipart[order[k]]]);
        if (k && !(F.iids[k - 1] < F.iids[k])) {
            set_error("the item id %s occurs in two of the texts", F.iids[k].c_str());
            return XMAP_ERR_ARG;
        }
    }
    F.uids.reserve(order_u.size()); F.ptr.reserve(order_u.size() + 1); F.item.resize(nnz); F.rating.resize(nnz); F.when.resize(nnz);
    F.ptr.push_back(0);
    {
        size_t w = 0;
        for (const int32_t u : order_u) {
            F.uids.emplace_back(unames[u]);
            for (int q = 0; q < n_parts; q++) {
                const int32_t x = run_of[q][u];
                if (x < 0) continue;
                for (int64_t e = runs[q][x].lo; e < runs[q][x].hi; e++, w++) {
                    F.item[w] = remap[ent[e].item]; F.rating[w] = ent[e].rating; F.when[w] = ent[e].when;
                }
            }
            F.ptr.push_back((int64_t)w);
        }
    }
    lap("phase 3 (latest rating, order, CSR)");
    predicates(F);
    lap("predicates");
    if (F.iids.size() && *std::max_element(F.suffix_cls.begin(), F.suffix_cls.end()) >= 32) {
        set_error("more than 32 distinct 2-char id suffixes (domain labels)");
        return XMAP_ERR_ARG;
    }
    *out = H.release();
    return XMAP_OK;
}

/* no C++ exception crosses the C ABI: running out of host memory is XMAP_ERR_CAPACITY */
#define XM_FEED_GUARD(call)                                                                         \
    try { return call; }                                                                            \
    catch (const std::bad_alloc &) { set_error("feeder: out of host memory"); return XMAP_ERR_CAPACITY; } \
    catch (const std::exception &e) { set_error("feeder: %s", e.what()); return XMAP_ERR_CAPACITY; }

int xmap_feed_texts(int32_t n_parts, const char *const *texts, const int64_t *lens, const char *const *labels, int32_t year_from,
                    int32_t year_to, int32_t min_ratings, xmap_feed **out) {
    XM_ARG(out && n_parts >= 1 && n_parts <= 30 && texts && lens && labels);
    for (int q = 0; q < n_parts; q++) XM_ARG((texts[q] || lens[q] == 0) && lens[q] >= 0 && labels[q]);
    *out = nullptr;
    XM_FEED_GUARD(feed_texts_impl(n_parts, texts, lens, labels, year_from, year_to, min_ratings, out))
}

int xmap_feed_text(const char *text, int64_t len, int32_t year_from, int32_t year_to, const char *label, int32_t min_ratings,
                   xmap_feed **out) {
    XM_ARG(label);
    return xmap_feed_texts(1, &text, &len, &label, year_from, year_to, min_ratings, out);
}

/* the union of two feeds (source + target domain of one problem): users of a in a's order, then the users only b has;
 * a user both have gets a's entries followed by b's; item ids must be disjoint (different domain labels) */
static int feed_merge_impl(const xmap_feed *a, const xmap_feed *b, xmap_feed **out) {
    const Feed &A = a->F, &B = b->F;
    std::unique_ptr<xmap_feed> H(new xmap_feed());
    Feed &F = H->F;
    F.n_lines = A.n_lines + B.n_lines; F.n_in_period = A.n_in_period + B.n_in_period;
    // items: merge of the two sorted tables
    std::vector<int32_t> ra(A.iids.size()), rb(B.iids.size());
    {
        size_t i = 0, j = 0;
        while (i < A.iids.size() || j < B.iids.size()) {
            if (j == B.iids.size() || (i < A.iids.size() && A.iids[i] < B.iids[j])) { ra[i] = (int32_t)F.iids.size(); F.iids.push_back(A.iids[i++]); }
            else if (i == A.iids.size() || B.iids[j] < A.iids[i]) { rb[j] = (int32_t)F.iids.size(); F.iids.push_back(B.iids[j++]); }
            else { set_error("the two feeds share the item id %s", A.iids[i].c_str()); return XMAP_ERR_ARG; }
        }
    }
    size_t cap = 1 << 10;
    while (cap < B.uids.size() * 2 + 16) cap <<= 1;
    StrMap inb(cap);
    for (size_t u = 0; u < B.uids.size(); u++) { bool fresh; inb.get(B.uids[u], hash_bytes(B.uids[u]), (int32_t)u, fresh); }
    std::vector<char> taken(B.uids.size(), 0);
    F.uids.reserve(A.uids.size() + B.uids.size());
    F.ptr.reserve(A.uids.size() + B.uids.size() + 1);
    F.item.reserve(A.item.size() + B.item.size()); F.rating.reserve(A.item.size() + B.item.size());
    F.when.reserve(A.item.size() + B.item.size());
    F.ptr.push_back(0);
    auto append = [&](const Feed &S, const std::vector<int32_t> &rm, size_t u) {
        for (int64_t e = S.ptr[u]; e < S.ptr[u + 1]; e++) {
            F.item.push_back(rm[S.item[e]]); F.rating.push_back(S.rating[e]); F.when.push_back(S.when[e]);
        }
    };
    for (size_t u = 0; u < A.uids.size(); u++) {
        F.uids.push_back(A.uids[u]);
        append(A, ra, u);
        const int32_t ub = inb.find(A.uids[u], hash_bytes(A.uids[u]));
        if (ub >= 0) { append(B, rb, (size_t)ub); taken[ub] = 1; }
        F.ptr.push_back((int64_t)F.item.size());
    }
    for (size_t u = 0; u < B.uids.size(); u++) {
        if (taken[u]) continue;
        F.uids.push_back(B.uids[u]);
        append(B, rb, u);
        F.ptr.push_back((int64_t)F.item.size());
    }
    predicates(F);
    if (F.iids.size() && *std::max_element(F.suffix_cls.begin(), F.suffix_cls.end()) >= 32) {
        set_error("more than 32 distinct 2-char id suffixes (domain labels)");
        return XMAP_ERR_ARG;
    }
    *out = H.release();
    return XMAP_OK;
}

int xmap_feed_merge(const xmap_feed *a, const xmap_feed *b, xmap_feed **out) {
    XM_ARG(a && b && out);
    *out = nullptr;
    XM_FEED_GUARD(feed_merge_impl(a, b, out))
}

int xmap_feed_sizes(const xmap_feed *f, int64_t *sizes /*[7]*/) {
    XM_ARG(f && sizes);
    const Feed &F = f->F;
    int64_t ub = 0, ib = 0;
    for (const std::string &s : F.uids) ub += (int64_t)s.size();
    for (const std::string &s : F.iids) ib += (int64_t)s.size();
    sizes[0] = (int64_t)F.uids.size(); sizes[1] = (int64_t)F.iids.size(); sizes[2] = (int64_t)F.item.size();
    sizes[3] = ub; sizes[4] = ib; sizes[5] = F.n_lines; sizes[6] = F.n_in_period;
    return XMAP_OK;
}

int xmap_feed_arrays(const xmap_feed *f, int64_t *user_ptr, int32_t *item, double *rating, double *when, int32_t *prefix_cls,
                     int32_t *suffix_cls, uint32_t *contains_mask, uint8_t *flags) {
    XM_ARG(f && user_ptr);
    const Feed &F = f->F;
    memcpy(user_ptr, F.ptr.data(), sizeof(int64_t) * F.ptr.size());
    if (item) memcpy(item, F.item.data(), sizeof(int32_t) * F.item.size());
    if (rating) memcpy(rating, F.rating.data(), sizeof(double) * F.rating.size());
    if (when) memcpy(when, F.when.data(), sizeof(double) * F.when.size());
    if (prefix_cls) memcpy(prefix_cls, F.prefix_cls.data(), sizeof(int32_t) * F.prefix_cls.size());
    if (suffix_cls) memcpy(suffix_cls, F.suffix_cls.data(), sizeof(int32_t) * F.suffix_cls.size());
    if (contains_mask) memcpy(contains_mask, F.contains.data(), sizeof(uint32_t) * F.contains.size());
    if (flags) memcpy(flags, F.flags.data(), F.flags.size());
    return XMAP_OK;
}

int xmap_feed_ids(const xmap_feed *f, int32_t which, char *bytes, int64_t *offsets) {
    XM_ARG(f && (which & ~3) == 0 && (offsets || (which & 2)));
    const std::vector<std::string> &T = (which & 1) ? f->F.iids : f->F.uids;
    const bool nl = (which & 2) != 0;          // a newline behind every id (ids are whitespace-free fields): bytes + n in all
    int64_t o = 0;
    for (size_t k = 0; k < T.size(); k++) {
        if (offsets) offsets[k] = o;
        if (bytes) memcpy(bytes + o, T[k].data(), T[k].size());
        o += (int64_t)T[k].size();
        if (nl) { if (bytes) bytes[o] = '\n'; o++; }
    }
    if (offsets) offsets[T.size()] = o;
    return XMAP_OK;
}

void xmap_feed_free(xmap_feed *f) { delete f; }

/* coarse ABI: the ratings of a feed -> the context (xmap_ctx_upload_ratings with the feed's arrays; `time` = the position of
 * the rating in the feed, like the Python engine, which keeps the time objects on the host) */
int xmap_ctx_upload_feed(xmap_ctx *ctx, const xmap_feed *f) {
    XM_ARG(ctx && f);
    const Feed &F = f->F;
    std::vector<float> r32(F.rating.size());
    std::vector<int64_t> tpos(F.rating.size());
    for (size_t e = 0; e < F.rating.size(); e++) { r32[e] = (float)F.rating[e]; tpos[e] = (int64_t)e; }
    static const int64_t zero = 0;
    return xmap_ctx_upload_ratings(ctx, (int64_t)F.uids.size(), (int32_t)F.iids.size(), F.ptr.empty() ? &zero : F.ptr.data(),
                                   F.item.data(), r32.data(), tpos.data(), F.prefix_cls.data(), F.suffix_cls.data(),
                                   F.contains.data(), F.flags.data());
}

/* test / bench utility: CSR + id tables -> Amazon-format text (the inverse of the feeder for one domain) */
int xmap_feed_format(int64_t n_users, const int64_t *user_ptr, const int32_t *item, const float *rating, const int64_t *when,
                     const char *uid_fmt, const char *iid_fmt, const int64_t *item_number, int32_t item_lo, int32_t item_hi,
                     char *out, int64_t cap, int64_t *written) {
    XM_ARG(user_ptr && item && rating && when && uid_fmt && iid_fmt && item_number && written);
    int64_t w = 0;
    char line[256];
    for (int64_t u = 0; u < n_users; u++) {
        char uid[64];
        snprintf(uid, sizeof(uid), uid_fmt, (long long)u);
        for (int64_t e = user_ptr[u]; e < user_ptr[u + 1]; e++) {
            if (item[e] < item_lo || item[e] >= item_hi) continue;
            char iid[64];
            snprintf(iid, sizeof(iid), iid_fmt, (long long)item_number[item[e]]);
            const int n = snprintf(line, sizeof(line), "%s\t%s\t%.1f\t%lld\n", uid, iid, (double)rating[e], (long long)when[e]);
            if (out) {
                if (w + n > cap) { set_error("text buffer too small"); return XMAP_ERR_CAPACITY; }
                memcpy(out + w, line, (size_t)n);
            }
            w += n;
        }
    }
    *written = w;
    return XMAP_OK;
}
}
