// stage_d.hip -- dense item-factor variant of the cross-domain similarity (BASELINE.json configs[4]; not in the
// reference: SURVEY.md 8c/8f-4).  Items carry K-dimensional factors; xsim(t, s) = cosine(F_t, F_s) is a dense
// (n_t x K) x (K x n_s) contraction on the fp32 matrix cores with the per-row top-k fused behind it, so the
// n_t x n_s similarity matrix (1.6e11 B at 200k x 200k) never exists in memory.
//
//   k_dense_normalize : Fn = F / ||F||  (norm in fp64, k order), one thread per row
//   k_dense_topk      : workgroup = 8 waves = 256 target rows; wave w keeps the A fragments of its 32 rows for the
//                       whole K in registers (K/2 VGPRs), the source items stream through LDS in tiles of 32 rows
//                       (register-prefetched, padded rows: conflict-free ds_read_b32), one v_mfma_f32_32x32x2_f32
//                       per k pair -> a 32x32 tile of similarities per wave and tile.  An MFMA f32 accumulation
//                       is bit for bit the k-ordered fmaf chain, which is what the oracle computes.
//                       Epilogue: each of the 16 accumulator registers holds one row's 32 columns on 32 lanes;
//                       values that beat the row's current k-th best (|sim| desc, index asc) are inserted into
//                       the row's unsorted top-k list in LDS (replace-the-worst, worst found by a wave min);
//                       rows are private to a wave, so no workgroup barrier is needed for the lists.
// MFMA-bound by design: 2 K n_t n_s flop against the 157 TFLOP/s fp32 matrix peak.
#include "common.h"

namespace xmap {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D_WAVES = 8;       // waves per workgroup: two per SIMD, so one wave's epilogue hides behind the other's MFMAs
constexpr int D_ROWS = 32 * D_WAVES;   // target rows per workgroup (32 per wave)
constexpr int D_THREADS = 64 * D_WAVES;
constexpr int D_TILE = 32;       // source items per tile
constexpr int D_TOPK = 64;       // list capacity per row (k <= 64: one lane per entry)

__global__ __launch_bounds__(256) void k_dense_normalize(int n, int K, const float *F, float *Fn) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double q = 0.0;
    for (int k = 0; k < K; k++) { double x = (double)F[(size_t)i * K + k]; q += x * x; }
    const double nrm = sqrt(q);
    for (int k = 0; k < K; k++)
        Fn[(size_t)i * K + k] = (nrm > 0.0) ? (float)((double)F[(size_t)i * K + k] / nrm) : 0.f;
}

// (|v|, idx) order: a is better than b
__device__ __forceinline__ bool d_better(float va, int ia, float vb, int ib) {
    const float aa = fabsf(va), ab = fabsf(vb);
    return (aa > ab) || (aa == ab && ia < ib);
}

template <int K>
__global__ __launch_bounds__(D_THREADS) void k_dense_topk(int n_t, int n_s, const float *Ft, const float *Fs, int k,
                                                    int *out_idx, float *out_val) {
    __shared__ float Bs[D_TILE][K + 1];
    __shared__ float Lval[D_ROWS][D_TOPK];
    __shared__ int Lidx[D_ROWS][D_TOPK];
    __shared__ int Lcnt[D_ROWS];
    __shared__ float Wval[D_ROWS];     // the row's current worst kept entry (valid once the list is full)
    __shared__ int Widx[D_ROWS];
    __shared__ int Wpos[D_ROWS];

    const int w = threadIdx.x >> 6, lane = lane_id();
    const int row0 = blockIdx.x * D_ROWS + w * 32;
    for (int r = lane; r < 32; r += 64) { Lcnt[w * 32 + r] = 0; }
    // A fragments: lane l holds A[i = l&31][k = 2 kk + (l>>5)]
    float a[K / 2];
    {
        const int i = row0 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < K / 2; kk++)
            a[kk] = (i < n_t) ? Ft[(size_t)i * K + 2 * kk + (lane >> 5)] : 0.f;
    }
    // tile staging: in step q thread t moves float q*D_THREADS + t of the 32 x K tile (coalesced loads, conflict-free stores)
    constexpr int PER = D_TILE * K / D_THREADS;
    float pre[PER];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = q * D_THREADS + threadIdx.x;
            const int jr = e / K, kc = e % K;
            const int j = c0 + jr;
            pre[q] = (j < n_s) ? Fs[(size_t)j * K + kc] : 0.f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < n_s; c0 += D_TILE) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = q * D_THREADS + threadIdx.x;
            Bs[e / K][e % K] = pre[q];
        }
        __syncthreads();
        if (c0 + D_TILE < n_s) fetch(c0 + D_TILE);   // overlaps with the MFMA loop below
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < K / 2; kk++) {
            const float b = Bs[lane & 31][2 * kk + (lane >> 5)];   // B[k = 2kk + (l>>5)][j = l&31]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc, 0, 0, 0);
        }
        // epilogue: register r holds row (r&3) + 8 (r>>2) + 4 (l>>5) of the wave's 32, column l&31
        const int j = c0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rl = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int gi = blockIdx.x * D_ROWS + rl;
            const float v = acc[r];
            bool cand = (gi < n_t) && (j < n_s);
            if (cand && Lcnt[rl] >= k) cand = d_better(v, j, Wval[rl], Widx[rl]);
            unsigned long long m = __ballot(cand);
            while (m) {   // insert the candidates one by one (all lanes cooperate on one row's list)
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int rr = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                const float cv = rlf(v, l);
                const int cj = c0 + (l & 31);
                int cnt = Lcnt[rr];
                if (cnt < k) {
                    if (lane == 0) { Lval[rr][cnt] = cv; Lidx[rr][cnt] = cj; Lcnt[rr] = cnt + 1; }
                    cnt++;
                    if (cnt < k) continue;
                } else {
                    // a later candidate of the same ballot may no longer beat the updated worst
                    if (!d_better(cv, cj, Wval[rr], Widx[rr])) continue;
                    if (lane == 0) { Lval[rr][Wpos[rr]] = cv; Lidx[rr][Wpos[rr]] = cj; }
                }
                // list is full: find its worst entry (lane per entry, wave min in the (|v|, idx) order)
                float wv = 0.f; int wi = -1, wp = lane;
                bool have = lane < k;
                if (have) { wv = Lval[rr][lane]; wi = Lidx[rr][lane]; }
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) {
                    const float ov = __shfl_xor(wv, s, 64);
                    const int oi = __shfl_xor(wi, s, 64), op = __shfl_xor(wp, s, 64);
                    const int oh = __shfl_xor((int)have, s, 64);
                    if (oh && (!have || d_better(wv, wi, ov, oi))) { wv = ov; wi = oi; wp = op; have = true; }
                }
                if (lane == 0) { Wval[rr] = wv; Widx[rr] = wi; Wpos[rr] = wp; }
            }
        }
    }
    // sort every row's list by (|v| desc, idx asc): rank by counting, lane per entry
    for (int r = 0; r < 32; r++) {
        const int rl = w * 32 + r;
        const int gi = blockIdx.x * D_ROWS + rl;
        if (gi >= n_t) break;
        const int cnt = Lcnt[rl];
        float v = 0.f; int id = 0x7fffffff;
        if (lane < cnt) { v = Lval[rl][lane]; id = Lidx[rl][lane]; }
        int rank = 0;
        for (int o = 0; o < cnt; o++) {
            const float ov = rlf(v, o);
            const int oi = rl32(id, o);
            rank += (lane < cnt && o != lane && d_better(ov, oi, v, id)) ? 1 : 0;
        }
        if (lane < cnt) { out_idx[(size_t)gi * k + rank] = id; out_val[(size_t)gi * k + rank] = v; }
        if (lane >= cnt && lane < k) { out_idx[(size_t)gi * k + lane] = -1; out_val[(size_t)gi * k + lane] = 0.f; }
    }
}

}  // namespace xmap

using namespace xmap;

extern "C" {

int xmap_dense_normalize(void *stream, int32_t n, int32_t dim, const float *F, float *Fn) {
    XM_ARG(F && Fn && n >= 0 && dim > 0);
    if (n == 0) return XMAP_OK;
    k_dense_normalize<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(n, dim, F, Fn);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}

int xmap_dense_topk(void *stream, int32_t n_t, int32_t n_s, int32_t dim, const float *Ft, const float *Fs, int32_t top_k,
                    int32_t *out_idx, float *out_val) {
    XM_ARG(Ft && Fs && out_idx && out_val && n_t >= 0 && n_s >= 0);
    XM_ARG(top_k >= 1 && top_k <= D_TOPK);
    if (dim != 128 && dim != 64) {
        set_error("dense variant: factor dimension %d not built (64 and 128 are)", dim);
        return XMAP_ERR_ARG;
    }
    if (n_t == 0) return XMAP_OK;
    dim3 grid((unsigned)((n_t + D_ROWS - 1) / D_ROWS)), block(D_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (dim == 128) k_dense_topk<128><<<grid, block, 0, st>>>(n_t, n_s, Ft, Fs, top_k, out_idx, out_val);
    else k_dense_topk<64><<<grid, block, 0, st>>>(n_t, n_s, Ft, Fs, top_k, out_idx, out_val);
    XM_LAUNCH_CHECK();
    return XMAP_OK;
}
}
