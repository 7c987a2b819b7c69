// item_stats.h -- get_universal_item_info (reference core/baselinerSim.py:40-82) for one item on a group of lanes; shared by
// the CSC-driven kernel of stage_a.hip and the rater-record-driven one of stage_a2.hip (one transposition per pass).
#pragma once
#include "common.h"

namespace xmap {

// one wave per item: lane-strided partial sums, fixed butterfly reduction (deterministic)
// stats of item i on a group of G lanes (G = 16: four items per wave; G = 64: the whole wave); gl = lane in the group.
// All lanes of the wave call it (the reductions are wave instructions); `on` says whether this group has an item.
// Src: where an item's raters come from -- src.load(p, rating, user) of rater p (CSC position) and src.uavg(user):
// the CSC arrays (CscSrc) or the rater records of the pair kernel (stage_a2.hip: RcSrc / RcWideSrc).
struct CscSrc {
    const int *iuser; const float *irating; const double *u_avg;
    static constexpr bool has_flags = false;
    __device__ __forceinline__ void load(long long p, double &r, int &u) const { r = (double)irating[p]; u = iuser[p]; }
    __device__ __forceinline__ double uavg(int u) const { return u_avg[u]; }
    __device__ __forceinline__ void set_flag(long long, bool) const {}
};
// POP = false: the caller never brings an item with more than 64 * 8 raters (its 8-way unrolled walk is compiled out)
template <int G, typename Src, bool POP = true>
__device__ __forceinline__ void item_stats_group(bool on, int i, int gl, int I, const long long *iptr, const Src src,
                                                 double *info, double *norms, int *ia_user, double *partial = nullptr) {
    long long p0 = 0, p1 = 0;
    if (on) { p0 = iptr[i]; p1 = iptr[i + 1]; }
    double s = 0.0, q = 0.0, a2 = 0.0, a2lo = 0.0;
    if (G < 64 || !POP || p1 - p0 <= 64 * 8) {
        for (long long p = p0 + gl; p < p1; p += G) {
            double r; int u;
            src.load(p, r, u);
            double d = r - src.uavg(u);
            s += r;
            q += r * r;
            dd_add(a2, a2lo, d * d);   // exact sum of the fp64 squares (order-independent)
        }
    } else {
        // popular items (up to 1e5 raters): 8 independent accumulators keep 8 gathers in flight per lane instead of
        // a chain of 1300 dependent round trips; the partials are merged exactly below
        constexpr int UN = 8;
        double su[UN], qu[UN], ah[UN], al[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) { su[t] = 0.0; qu[t] = 0.0; ah[t] = 0.0; al[t] = 0.0; }
        // software pipeline: the ratings and users of the NEXT round are requested before this round's user averages
        // are gathered, so a round costs one dependent round trip, not two (156 rounds for the most popular item: the
        // kernel's tail).  Same partial sums in the same order.
        double rr[UN], nr[UN];
        int uu[UN], nu[UN];
        auto fetch = [&](long long p, double *r_, int *u_) {
#pragma unroll
            for (int t = 0; t < UN; t++) {
                const long long pp = p + 64 * t;
                r_[t] = 0.0; u_[t] = -1;
                if (pp < p1) src.load(pp, r_[t], u_[t]);
            }
        };
        fetch(p0 + gl, nr, nu);
        for (long long p = p0 + gl; p < p1; p += 64 * UN) {
#pragma unroll
            for (int t = 0; t < UN; t++) { rr[t] = nr[t]; uu[t] = nu[t]; }
            fetch(p + 64 * UN, nr, nu);      // out-of-range entries come back as (0, -1)
            double av[UN];
#pragma unroll
            for (int t = 0; t < UN; t++) av[t] = uu[t] >= 0 ? src.uavg(uu[t]) : 0.0;
#pragma unroll
            for (int t = 0; t < UN; t++) {
                if (uu[t] < 0) continue;
                const double r = rr[t];
                const double d = r - av[t];
                su[t] += r;
                qu[t] += r * r;
                dd_add(ah[t], al[t], d * d);
            }
        }
#pragma unroll
        for (int t = 0; t < UN; t++) {
            s += su[t];
            q += qu[t];
            dd_add(a2, a2lo, ah[t]);
            dd_add(a2, a2lo, al[t]);
        }
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); q += __shfl_xor(q, m, 64); }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
        double oh = __shfl_down(a2, m, G), ol = __shfl_down(a2lo, m, G);
        dd_add(a2, a2lo, oh);
        dd_add(a2, a2lo, ol);
    }
    if (partial) {     // user-sharded input: this rank's share of the item's sums (k_item_merge adds the shares up)
        if (on && gl == 0) {
            double *o = partial + (size_t)i * 5;
            o[0] = s; o[1] = q; o[2] = a2; o[3] = a2lo; o[4] = (double)(p1 - p0);
        }
        return;
    }
    a2 = __shfl(a2, 0, G);
    double n = (double)(p1 - p0);
    double avg = (p1 > p0) ? 1.0 * s / n : 0.0;
    if (on && gl == 0) {
        info[(size_t)i * 4 + 0] = avg;
        info[(size_t)i * 4 + 1] = sqrt(q);
        info[(size_t)i * 4 + 2] = sqrt(a2);
        info[(size_t)i * 4 + 3] = 1.0 * n;
        if (norms) {   // dense copies of the two norms: 8 B per item stays L2-resident for the per-pair gathers
            norms[i] = sqrt(q);
            norms[(size_t)I + i] = sqrt(a2);
        }
    }
    if (Src::has_flags)     // the rater records carry `rating >= item average` (retrieve_path_info, baselinerSim.py:97-113)
        for (long long p = p0 + gl; p < p1; p += G) {
            double r; int u;
            src.load(p, r, u);
            src.set_flag(p, r >= avg);
        }
    if (!ia_user) return;   // the flag-packed copies are read by the complete-rows formulation only
    for (long long p = p0 + gl; p < p1; p += G) {
        double r; int u;
        src.load(p, r, u);
        unsigned ge = (r >= avg) ? 0x80000000u : 0u;
        ia_user[p] = (int)((unsigned)u | ge);
    }
}


}  // namespace xmap
