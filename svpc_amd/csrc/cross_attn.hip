// Decoder cross-attention over the ≤ 3 memory rows of a sentence, fused with its residual LayerNorm — forward and backward, one
// launch each per decoder layer (reference: src/rtransformer/model.py:657-658 inside BertDecoderLayerNoMemoryUntied.forward :630-663,
// the attention core :194-219, BertLayerNorm :143-156):
//
//   x2 = LayerNorm(x1 + MHA(query rows q = x1·Wqᵀ + bq, keys / values = the sentence's n_mem memory rows))
//
// "trivial; fuse".  With ≤ 3 keys per sentence the attention is a handful of 64-wide dot products per row and head:
// one workgroup per sentence, D threads — a thread owns ONE model column of all Lt rows (registers), a 64-lane wave is one head at
// dh = 64.  Scores are segment sums whose result every lane of the head receives, so softmax, dropout draw and the weighted sum of the
// value rows are computed in place with no LDS image and no barrier; the LayerNorm statistics are the only cross-wave step (row sums
// through a few hundred bytes of LDS).  Backward likewise: LayerNorm backward, dV / d p̃ (segment sums), softmax backward, dq and dk
// column-local, one launch instead of LayerNorm backward + attention backward.  It replaced, per decoder layer, two launches forward
// and two backward (12.8 + 10.4 µs and 11.8 + 19 µs at the headline shape).
// Arithmetic is fp32 on the stored values (split rows enter as hi + lo).  Dropout of the attention probabilities uses the attention
// kernels' draw (common.h: attn_drop_scale, row = (sentence·H + head)·Lt + query, k = key), recomputed in backward.
#include "common.h"

namespace {

constexpr int XA_NM = 3;        // memory rows per sentence (vivt / viv: 3, vi: 2, v: 1)

struct XaArgs {
    const void* q; int q_dt, ldq, loq;           // dt: 0 fp32, 1 bf16, 2 split (hi at col, lo at col + lo)
    const void* x1; int x_dt, ldx, lox;          // residual rows
    const void* k; const void* v; int kv_dt, ld_kv, lokv;      // memory rows: K / V column blocks of the memory projection
    const float* gamma; const float* beta; float eps;
    void* y; int y_dt, ldy, loy;
    float* probs;                                // (T·lt, H, 4) normalised probabilities BEFORE dropout (saved for backward)
    float* mean; float* rstd;                    // (T·lt)
    const int* row_off; const int* row_len;      // ragged sentences (both or neither): sentence s owns rows [row_off[s], row_off[s] + row_len[s]),
                                                 // row_len[s] <= lt; null: sentence s owns rows [s·lt, (s+1)·lt)
    int lt, nm, D, H;
    float scale, p_drop; uint32_t site; const u64* seed;
    // backward
    const void* dy; int dy_dt, lddy;             // dense gradient rows of y
    void* dq; void* dres; int dg_dt, lddg;       // dense gradient rows of q and of the residual x1 (same type and stride)
    void* dk; void* dv; int dkv_dt, ld_dkv;      // gradient rows of the memory projection's K / V blocks (dense planes)
    float* part_ln;                              // (T, 2D): per-sentence [dgamma | dbeta]
};

// the storage kind is a COMPILE-TIME parameter: a runtime test around every load puts each into a conditional region of its own with a
// drain of the memory counter behind it — the 2·Lt + 6 loads a thread issues "up front" were that many dependent round trips
// (measured: 58 µs forward / 82 µs backward per launch instead of ≈10 / 15)
template <int DT> __device__ __forceinline__ float xa_load(const void* p, size_t off, int lo) {
    if constexpr (DT == 0) return reinterpret_cast<const float*>(p)[off];
    const __bf16* b = reinterpret_cast<const __bf16*>(p);
    float v = (float)b[off];
    if constexpr (DT == 2) v += (float)b[off + lo];
    return v;
}
template <int DT> __device__ __forceinline__ void xa_store(void* p, size_t off, int lo, float v) {
    if constexpr (DT == 0) { reinterpret_cast<float*>(p)[off] = v; return; }
    __bf16* b = reinterpret_cast<__bf16*>(p);
    const __bf16 h = (__bf16)v;
    b[off] = h;
    if constexpr (DT == 2) b[off + lo] = (__bf16)(v - (float)h);
}
template <int DH> __device__ __forceinline__ float seg_sum(float v);
template <> __device__ __forceinline__ float seg_sum<64>(float v) { return wave_sum(v); }
template <> __device__ __forceinline__ float seg_sum<32>(float v) {      // sum over each 32-lane half (two heads per wave)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
    return v;
}

// softmax over the nm keys of one (row, head) from the three raw dot products; p[] normalised, pt[] = p · dropout multiplier
__device__ __forceinline__ void xa_softmax(const float (&sc)[XA_NM], int nm, float scale, float p_drop, u64 seed, uint32_t site, u64 row,
                                           float inv_keep, float (&p)[XA_NM], float (&pt)[XA_NM]) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) { p[j] = j < nm ? scale * sc[j] : -INFINITY; m = fmaxf(m, p[j]); }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) { p[j] = j < nm ? expf(p[j] - m) : 0.f; sum += p[j]; }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        p[j] *= inv;
        float mult = 1.0f;
        if (p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, site, row, (uint32_t)j, p_drop, inv_keep);
        pt[j] = p[j] * mult;
    }
}

// LDS: red[NW][LTM][2] + tot[LTM][2]
template <int DH, int NPL, int LTM, int KIND>
__global__ __launch_bounds__(64 * NPL) void xattn_ln_fwd_kernel(XaArgs a) {
    constexpr int NW = NPL;
    __shared__ float red[NW][LTM][2];
    __shared__ float tot[LTM][2];
    const int D = a.D, H = a.H, ltu = a.lt, nm = a.nm;          // ltu: the uniform (maximum) sentence length — stride of the dropout rows
    const int s = blockIdx.x, d = threadIdx.x;
    const int roff = a.row_off ? a.row_off[s] : s * ltu;         // this sentence's first row and its length (ragged: valid tokens only)
    const int lt = a.row_len ? a.row_len[s] : ltu;
    const int lane = d & 63, wave = d >> 6, h = d / DH;
    // every load of the thread's column up front (Lt query values, Lt residual values, nm keys and values): one memory round trip
    float qv[LTM], xv[LTM], kk[XA_NM], vv[XA_NM];
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        const size_t row = (size_t)(roff + min(t, lt - 1));
        qv[t] = xa_load<KIND>(a.q, row * a.ldq + d, a.loq);
        xv[t] = xa_load<KIND>(a.x1, row * a.ldx + d, a.lox);
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        const size_t row = (size_t)(s * nm + min(j, nm - 1));
        kk[j] = xa_load<KIND>(a.k, row * a.ld_kv + d, a.lokv);
        vv[j] = xa_load<KIND>(a.v, row * a.ld_kv + d, a.lokv);
    }
    const float gam = a.gamma[d], bet = a.beta[d];
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    // scores, softmax, dropout, weighted values, residual: per row, every lane of a head holds the head's probabilities
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        if (t < lt) {
            float sc[XA_NM], p[XA_NM], pt[XA_NM];
#pragma unroll
            for (int j = 0; j < XA_NM; ++j) sc[j] = seg_sum<DH>(qv[t] * kk[j]);
            xa_softmax(sc, nm, a.scale, a.p_drop, seed, a.site, (u64)(s * H + h) * ltu + t, inv_keep, p, pt);
            if ((d & (DH - 1)) == 0) {
                float* pr = a.probs + ((size_t)(roff + t) * H + h) * 4;
                pr[0] = p[0]; pr[1] = p[1]; pr[2] = p[2]; pr[3] = 0.f;
            }
            xv[t] += pt[0] * vv[0] + pt[1] * vv[1] + pt[2] * vv[2];
            const float ws = wave_sum(xv[t]);
            if (lane == 0) red[wave][t][0] = ws;
        }
    }
    __syncthreads();
    if (d < lt) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m += red[w][d][0];
        tot[d][0] = m / (float)D;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        if (t < lt) {
            const float c = xv[t] - tot[t][0];
            const float ws = wave_sum(c * c);
            if (lane == 0) red[wave][t][1] = ws;
        }
    }
    __syncthreads();
    if (d < lt) {
        float q2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) q2 += red[w][d][1];
        const float rstd = 1.0f / sqrtf(q2 / (float)D + a.eps);
        tot[d][1] = rstd;
        a.mean[(size_t)roff + d] = tot[d][0];
        a.rstd[(size_t)roff + d] = rstd;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < LTM; ++t)
        if (t < lt) xa_store<KIND>(a.y, (size_t)(roff + t) * a.ldy + d, a.loy, (xv[t] - tot[t][0]) * tot[t][1] * gam + bet);
}

template <int DH, int NPL, int LTM, int KIND>
__global__ __launch_bounds__(64 * NPL) void xattn_ln_bwd_kernel(XaArgs a) {
    constexpr int NW = NPL;
    __shared__ float red[NW][LTM][2];
    __shared__ float tot[LTM][2];
    __shared__ float stat[LTM][2];
    __shared__ __attribute__((aligned(16))) float pps[LTM][16][4];
    const int D = a.D, H = a.H, ltu = a.lt, nm = a.nm;          // ltu: the uniform (maximum) sentence length — stride of the dropout rows
    const int s = blockIdx.x, d = threadIdx.x;
    const int roff = a.row_off ? a.row_off[s] : s * ltu;         // this sentence's first row and its length (ragged: valid tokens only)
    const int lt = a.row_len ? a.row_len[s] : ltu;
    const int lane = d & 63, wave = d >> 6, h = d / DH;
    float qv[LTM], xh[LTM], g[LTM], kk[XA_NM], vv[XA_NM];
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        const size_t row = (size_t)(roff + min(t, lt - 1));
        qv[t] = xa_load<KIND>(a.q, row * a.ldq + d, a.loq);
        xh[t] = xa_load<KIND>(a.x1, row * a.ldx + d, a.lox);
        g[t] = xa_load<(KIND == 0 ? 0 : 1)>(a.dy, row * a.lddy + d, 0);
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        const size_t row = (size_t)(s * nm + min(j, nm - 1));
        kk[j] = xa_load<KIND>(a.k, row * a.ld_kv + d, a.lokv);
        vv[j] = xa_load<KIND>(a.v, row * a.ld_kv + d, a.lokv);
    }
    if (d < lt) { stat[d][0] = a.mean[(size_t)roff + d]; stat[d][1] = a.rstd[(size_t)roff + d]; }
    // (the saved probabilities of the sentence through LDS, one coalesced round trip: a load inside the per-row blocks below would be a
    // memory round trip per row on the one chain of Lt rows this workgroup is)
    for (int i = d; i < lt * H; i += 64 * NPL)
        *reinterpret_cast<float4*>(&pps[i / H][i % H][0]) = *reinterpret_cast<const float4*>(a.probs + ((size_t)roff * H + i) * 4);
    const float gam = a.gamma[d];
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    __syncthreads();
    // (1) x̂ recomputed from x1 + Σ p̃·v (the probabilities as the forward saved them, the dropout draw again); g = dy·γ; row sums Σ g, Σ g·x̂
    float dgam = 0.f, dbet = 0.f;
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        if (t < lt) {
            const float* pr = &pps[t][h][0];
            float pt[XA_NM];
#pragma unroll
            for (int j = 0; j < XA_NM; ++j) {
                float mult = 1.0f;
                if (a.p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, a.site, (u64)(s * H + h) * ltu + t, (uint32_t)j, a.p_drop, inv_keep);
                pt[j] = j < nm ? pr[j] * mult : 0.f;
            }
            const float yv = xh[t] + pt[0] * vv[0] + pt[1] * vv[1] + pt[2] * vv[2];
            const float xn = (yv - stat[t][0]) * stat[t][1];
            const float dyv = g[t];
            dgam += dyv * xn; dbet += dyv;
            xh[t] = xn;
            g[t] = dyv * gam;
            const float s1 = wave_sum(g[t]), s2 = wave_sum(g[t] * xn);
            if (lane == 0) { red[wave][t][0] = s1; red[wave][t][1] = s2; }
        }
    }
    a.part_ln[(size_t)s * 2 * D + d] = dgam;
    a.part_ln[(size_t)s * 2 * D + D + d] = dbet;
    __syncthreads();
    if (d < 2 * lt) {
        const int t = d >> 1, c = d & 1;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][t][c];
        tot[t][c] = v / (float)D;
    }
    __syncthreads();
    // (2) pre-LayerNorm gradient (= gradient of the residual rows and of the attended vector), then the attention backward, column-local
    float dkk[XA_NM] = {0.f, 0.f, 0.f}, dvv[XA_NM] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        if (t < lt) {
            const float dp = stat[t][1] * (g[t] - tot[t][0] - xh[t] * tot[t][1]);
            const size_t row = (size_t)(roff + t);
            xa_store<(KIND == 0 ? 0 : 1)>(a.dres, row * a.lddg + d, 0, dp);
            const float* pr = &pps[t][h][0];
            float p[XA_NM], dpt[XA_NM], dot = 0.f;
#pragma unroll
            for (int j = 0; j < XA_NM; ++j) {
                float mult = 1.0f;
                if (a.p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, a.site, (u64)(s * H + h) * ltu + t, (uint32_t)j, a.p_drop, inv_keep);
                p[j] = j < nm ? pr[j] : 0.f;
                dvv[j] += p[j] * mult * dp;                            // dV[j] += p̃[t, j]·d o[t]
                dpt[j] = j < nm ? seg_sum<DH>(dp * vv[j]) * mult : 0.f;  // d p[t, j] = <d o[t], v[j]>_head · dropout multiplier
                dot += p[j] * dpt[j];
            }
            float dqv = 0.f;
#pragma unroll
            for (int j = 0; j < XA_NM; ++j) {
                const float gs = a.scale * p[j] * (dpt[j] - dot);      // gradient of the raw dot product <q[t], k[j]>_head
                dqv += gs * kk[j];
                dkk[j] += gs * qv[t];
            }
            xa_store<(KIND == 0 ? 0 : 1)>(a.dq, row * a.lddg + d, 0, dqv);
        }
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        if (j < nm) {
            const size_t o = (size_t)(s * nm + j) * a.ld_dkv + d;
            xa_store<(KIND == 0 ? 0 : 1)>(a.dk, o, 0, dkk[j]);
            xa_store<(KIND == 0 ? 0 : 1)>(a.dv, o, 0, dvv[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dh = 64: butterfly reductions
// The kernels above spend their time in vector instructions (a wave64 VALU op occupies its 16-lane SIMD for 4 cycles: ≈7,000 instructions
// per wave × 3 waves per SIMD = 35 µs): every one of the 3·Lt score sums was a 64-lane reduction of its own (≈15 instructions) and every
// lane of a head evaluated the row's softmax and dropout draws redundantly.  Here a wave reduces 32 row values AT ONCE with a
// reduce-scatter butterfly — v_permlane32_swap / v_permlane16_swap exchange register halves across lane^32 / lane^16, DPP row_ror:8,
// row_half_mirror and the quad permutes finish inside a row; 70 instructions for 32 sums — after which lane l holds the 64-lane sum of
// row l >> 1.  So each (row, head) softmax / dropout draw / softmax backward is evaluated by ONE lane pair, and the results go back to
// the columns as wave-uniform scalars (v_readlane): ≈1,100 instructions per wave forward, ≈1,700 backward.
template <int CTRL> __device__ __forceinline__ float xa_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// in: v[i] = this lane's addend of value i (i < 32); out: the sum over the 64 lanes of value (lane >> 1)  (common.h)
__device__ __forceinline__ float xa_rs32(const float (&v)[32], int lane) { return wave_reduce_scatter32(v, lane); }
__device__ __forceinline__ float xa_bcast(float v, int src_lane) {       // wave-uniform copy of lane `src_lane`'s value (compile-time lane)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

constexpr int XA_LT = 24;       // rows per sentence of the fast form (Lt = 22 in the reference's scripts)

template <int NPL, int KIND>
__global__ __launch_bounds__(64 * NPL) void xattn64_fwd_kernel(XaArgs a) {
    constexpr int NW = NPL, LTM = XA_LT;
    __shared__ float red[NW][32];
    __shared__ float tot[2][32];
    const int D = a.D, H = a.H, ltu = a.lt, nm = a.nm;          // ltu: the uniform (maximum) sentence length — stride of the dropout rows
    const int s = blockIdx.x, d = threadIdx.x;
    const int roff = a.row_off ? a.row_off[s] : s * ltu;         // this sentence's first row and its length (ragged: valid tokens only)
    const int lt = a.row_len ? a.row_len[s] : ltu;
    const int lane = d & 63, wave = d >> 6, h = wave;
    const int r = lane >> 1;                    // the row this lane pair evaluates the softmax of
    float qv[LTM], xv[LTM], kk[XA_NM], vv[XA_NM];
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        const size_t row = (size_t)(roff + min(t, lt - 1));
        qv[t] = xa_load<KIND>(a.q, row * a.ldq + d, a.loq);
        xv[t] = xa_load<KIND>(a.x1, row * a.ldx + d, a.lox);
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        const size_t row = (size_t)(s * nm + min(j, nm - 1));
        kk[j] = xa_load<KIND>(a.k, row * a.ld_kv + d, a.lokv);
        vv[j] = j < nm ? xa_load<KIND>(a.v, row * a.ld_kv + d, a.lokv) : 0.f;
    }
    const float gam = a.gamma[d], bet = a.beta[d];
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    // the head's Lt × nm dot products, one butterfly per key; then ONE softmax per (row, head)
    float sc[XA_NM], p[XA_NM], pt[XA_NM];
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        float pr[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) pr[t] = t < LTM ? qv[t] * kk[j] : 0.f;
        sc[j] = xa_rs32(pr, lane);
    }
    xa_softmax(sc, nm, a.scale, a.p_drop, seed, a.site, (u64)(s * H + h) * ltu + min(r, lt - 1), inv_keep, p, pt);
    if ((lane & 1) == 0 && r < lt) {
        float* po = a.probs + ((size_t)(roff + r) * H + h) * 4;
        *reinterpret_cast<float4*>(po) = make_float4(p[0], p[1], p[2], 0.f);
    }
    // attended vector + residual per column: the row's three dropped-out probabilities as wave-uniform scalars
    {
        float part[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < LTM) {
                xv[t] += xa_bcast(pt[0], 2 * t) * vv[0] + xa_bcast(pt[1], 2 * t) * vv[1] + xa_bcast(pt[2], 2 * t) * vv[2];
                part[t] = xv[t];
            } else part[t] = 0.f;
        }
        const float ws = xa_rs32(part, lane);
        if ((lane & 1) == 0) red[wave][r] = ws;
    }
    __syncthreads();
    if (d < 32) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m += red[w][d];
        tot[0][d] = m / (float)D;
    }
    __syncthreads();
    {
        float part[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < LTM) { xv[t] -= tot[0][t]; part[t] = xv[t] * xv[t]; } else part[t] = 0.f;
        }
        const float ws = xa_rs32(part, lane);
        __syncthreads();                          // (everyone has read the means' partials in `red`)
        if ((lane & 1) == 0) red[wave][r] = ws;
    }
    __syncthreads();
    if (d < 32) {
        float q2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) q2 += red[w][d];
        const float rstd = 1.0f / sqrtf(q2 / (float)D + a.eps);
        tot[1][d] = rstd;
        if (d < lt) { a.mean[(size_t)roff + d] = tot[0][d]; a.rstd[(size_t)roff + d] = rstd; }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < LTM; ++t)
        if (t < lt) xa_store<KIND>(a.y, (size_t)(roff + t) * a.ldy + d, a.loy, xv[t] * tot[1][t] * gam + bet);
}

template <int NPL, int KIND>
__global__ __launch_bounds__(64 * NPL) void xattn64_bwd_kernel(XaArgs a) {
    constexpr int NW = NPL, LTM = XA_LT, GK = KIND == 0 ? 0 : 1;
    __shared__ float red[NW][2][32];
    __shared__ float tot[2][32];
    __shared__ float stat[2][32];
    __shared__ __attribute__((aligned(16))) float pps[32][16][4];
    const int D = a.D, H = a.H, ltu = a.lt, nm = a.nm;          // ltu: the uniform (maximum) sentence length — stride of the dropout rows
    const int s = blockIdx.x, d = threadIdx.x;
    const int roff = a.row_off ? a.row_off[s] : s * ltu;         // this sentence's first row and its length (ragged: valid tokens only)
    const int lt = a.row_len ? a.row_len[s] : ltu;
    const int lane = d & 63, wave = d >> 6, h = wave;
    const int r = lane >> 1, rc = min(r, lt - 1);
    float xh[LTM], g[LTM], kk[XA_NM], vv[XA_NM];
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        const size_t row = (size_t)(roff + min(t, lt - 1));
        xh[t] = xa_load<KIND>(a.x1, row * a.ldx + d, a.lox);
        g[t] = xa_load<GK>(a.dy, row * a.lddy + d, 0);
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        const size_t row = (size_t)(s * nm + min(j, nm - 1));
        kk[j] = j < nm ? xa_load<KIND>(a.k, row * a.ld_kv + d, a.lokv) : 0.f;
        vv[j] = j < nm ? xa_load<KIND>(a.v, row * a.ld_kv + d, a.lokv) : 0.f;
    }
    if (d < 32) { const int t = min(d, lt - 1); stat[0][d] = a.mean[(size_t)roff + t]; stat[1][d] = a.rstd[(size_t)roff + t]; }
    for (int i = d; i < lt * H; i += 64 * NPL)
        *reinterpret_cast<float4*>(&pps[i / H][i % H][0]) = *reinterpret_cast<const float4*>(a.probs + ((size_t)roff * H + i) * 4);
    const float gam = a.gamma[d];
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    __syncthreads();
    // this lane pair's row: probabilities as saved, the dropout multipliers again
    float p[XA_NM], mult[XA_NM], pt[XA_NM];
    {
        const float4 p4 = *reinterpret_cast<const float4*>(&pps[rc][h][0]);
        p[0] = p4.x; p[1] = p4.y; p[2] = p4.z;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            mult[j] = 1.0f;
            if (a.p_drop > 0.f && j < nm) mult[j] = attn_drop_scale(seed, a.site, (u64)(s * H + h) * ltu + rc, (uint32_t)j, a.p_drop, inv_keep);
            if (j >= nm) p[j] = 0.f;
            pt[j] = p[j] * mult[j];
        }
    }
    // (1) x̂ recomputed, g = dy·γ, LayerNorm row sums by two butterflies, dγ / dβ partial sums
    float dgam = 0.f, dbet = 0.f;
    {
        float p1[32], p2[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < LTM) {
                const float yv = xh[t] + xa_bcast(pt[0], 2 * t) * vv[0] + xa_bcast(pt[1], 2 * t) * vv[1] + xa_bcast(pt[2], 2 * t) * vv[2];
                const float xn = (yv - stat[0][t]) * stat[1][t];
                const float dyv = t < lt ? g[t] : 0.f;
                dgam += dyv * xn; dbet += dyv;
                xh[t] = xn;
                g[t] = dyv * gam;
                p1[t] = g[t]; p2[t] = g[t] * xn;
            } else { p1[t] = 0.f; p2[t] = 0.f; }
        }
        const float s1 = xa_rs32(p1, lane), s2 = xa_rs32(p2, lane);
        if ((lane & 1) == 0) { red[wave][0][r] = s1; red[wave][1][r] = s2; }
    }
    a.part_ln[(size_t)s * 2 * D + d] = dgam;
    a.part_ln[(size_t)s * 2 * D + D + d] = dbet;
    __syncthreads();
    if (d < 64) {
        const int c = d >> 5, t = d & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][c][t];
        tot[c][t] = v / (float)D;
    }
    __syncthreads();
    // (the query column is needed by the last phase only: requested here, it lands under phase 2 instead of occupying Lt registers
    // from the top of the kernel)
    float qv[LTM];
#pragma unroll
    for (int t = 0; t < LTM; ++t) qv[t] = xa_load<KIND>(a.q, (size_t)(roff + min(t, lt - 1)) * a.ldq + d, a.loq);
    // (2) pre-LayerNorm gradient; d p̃ by three butterflies; dV column sums
    float dpt[XA_NM], dvv[XA_NM] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        g[t] = stat[1][t] * (g[t] - tot[0][t] - xh[t] * tot[1][t]);          // g[] now holds d(x1 + o)
        if (t < lt) xa_store<GK>(a.dres, (size_t)(roff + t) * a.lddg + d, 0, g[t]);
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) dvv[j] += xa_bcast(pt[j], 2 * t) * (t < lt ? g[t] : 0.f);
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        float pr[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) pr[t] = t < LTM ? g[t] * vv[j] : 0.f;
        dpt[j] = xa_rs32(pr, lane) * mult[j];
    }
    // softmax backward for this lane pair's row: gs[j] = gradient of the raw dot product <q[r], k[j]>_head
    float gs[XA_NM];
    {
        const float dot = p[0] * dpt[0] + p[1] * dpt[1] + p[2] * dpt[2];
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) gs[j] = r < lt ? a.scale * p[j] * (dpt[j] - dot) : 0.f;
    }
    // (3) dq[t][d] = Σ_j gs[t, j]·k[j][d];  dk[j][d] = Σ_t gs[t, j]·q[t][d]
    float dkk[XA_NM] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LTM; ++t) {
        const float g0 = xa_bcast(gs[0], 2 * t), g1 = xa_bcast(gs[1], 2 * t), g2 = xa_bcast(gs[2], 2 * t);
        if (t < lt) xa_store<GK>(a.dq, (size_t)(roff + t) * a.lddg + d, 0, g0 * kk[0] + g1 * kk[1] + g2 * kk[2]);
        dkk[0] += g0 * qv[t]; dkk[1] += g1 * qv[t]; dkk[2] += g2 * qv[t];
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        if (j < nm) {
            const size_t o = (size_t)(s * nm + j) * a.ld_dkv + d;
            xa_store<GK>(a.dk, o, 0, dkk[j]);
            xa_store<GK>(a.dv, o, 0, dvv[j]);
        }
    }
}

template <int NPL, int KIND>
int xa64_launch(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    if (bwd) hipLaunchKernelGGL((xattn64_bwd_kernel<NPL, KIND>), dim3(T), dim3(64 * NPL), 0, s, a);
    else hipLaunchKernelGGL((xattn64_fwd_kernel<NPL, KIND>), dim3(T), dim3(64 * NPL), 0, s, a);
    return svpc_check_launch(bwd ? "cross_attn_ln_bwd" : "cross_attn_ln_fwd");
}
template <int NPL>
int xa64_kind(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    if (a.q_dt == 0) return xa64_launch<NPL, 0>(a, T, bwd, s);
    if (a.q_dt == 1) return xa64_launch<NPL, 1>(a, T, bwd, s);
    return xa64_launch<NPL, 2>(a, T, bwd, s);
}

template <int DH, int NPL, int LTM, int KIND>
int xa_launch(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    if (bwd) hipLaunchKernelGGL((xattn_ln_bwd_kernel<DH, NPL, LTM, KIND>), dim3(T), dim3(64 * NPL), 0, s, a);
    else hipLaunchKernelGGL((xattn_ln_fwd_kernel<DH, NPL, LTM, KIND>), dim3(T), dim3(64 * NPL), 0, s, a);
    return svpc_check_launch(bwd ? "cross_attn_ln_bwd" : "cross_attn_ln_fwd");
}
template <int DH, int NPL, int LTM>
int xa_launch_kind(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    if (a.q_dt == 0) return xa_launch<DH, NPL, LTM, 0>(a, T, bwd, s);
    if (a.q_dt == 1) return xa_launch<DH, NPL, LTM, 1>(a, T, bwd, s);
    return xa_launch<DH, NPL, LTM, 2>(a, T, bwd, s);
}
template <int DH, int NPL>
int xa_launch_lt(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    if (a.lt <= 8) return xa_launch_kind<DH, NPL, 8>(a, T, bwd, s);
    return xa_launch_kind<DH, NPL, 24>(a, T, bwd, s);
}
int xa_dispatch(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    const int dh = a.D / a.H, npl = a.D / 64;
    static int slow = -1;            // SVPC_XATTN_SLOW=1: the per-sum reductions for 64-wide heads too (A/B and test cross-check)
    if (slow < 0) { const char* e = getenv("SVPC_XATTN_SLOW"); slow = e ? atoi(e) : 0; }
    if (dh == 64 && !slow) {
        if (npl == 12) return xa64_kind<12>(a, T, bwd, s);
        if (npl == 8) return xa64_kind<8>(a, T, bwd, s);
        if (npl == 4) return xa64_kind<4>(a, T, bwd, s);
    }
    if (dh == 64 && npl == 12) return xa_launch_lt<64, 12>(a, T, bwd, s);
    if (dh == 64 && npl == 8) return xa_launch_lt<64, 8>(a, T, bwd, s);
    if (dh == 64 && npl == 4) return xa_launch_lt<64, 4>(a, T, bwd, s);
    if (dh == 32 && npl == 2) return xa_launch_lt<32, 2>(a, T, bwd, s);
    if (dh == 32 && npl == 4) return xa_launch_lt<32, 4>(a, T, bwd, s);
    svpc_set_error("cross_attn_ln: unsupported (hidden size, heads)");
    return -1;
}

}  // namespace

extern "C" {

int svpc_cross_attn_ln_fwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy,
                             float* probs, float* mean, float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const u64* seed, const int* row_off, const int* row_len, hipStream_t stream);
int svpc_cross_attn_ln_bwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* probs, const float* mean, const float* rstd,
                             const void* dy, int dy_dt, int lddy, void* dq, void* dres, int dg_dt, int lddg, void* dk, void* dv, int dkv_dt,
                             int ld_dkv, float* part_ln, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const u64* seed, const int* row_off, const int* row_len, hipStream_t stream);

// 1 if the fused cross-attention + LayerNorm kernels take this shape: D ∈ {256, 512, 768} with 64-wide heads or D ∈ {128, 256} with
// 32-wide heads, ≤ 24 sentence rows, ≤ 3 memory rows
int svpc_cross_attn_ln_supported(int D, int H, int lt, int nm) {
    if (H <= 0 || H > 16 || D % H || D % 64 || lt < 1 || lt > 24 || nm < 1 || nm > XA_NM) return 0;
    const int dh = D / H, npl = D / 64;
    return ((dh == 64 && (npl == 12 || npl == 8 || npl == 4)) || (dh == 32 && (npl == 2 || npl == 4))) ? 1 : 0;
}

// forward: q, x1 (T·lt rows each; dt 0 fp32 / 1 bf16 / 2 split with the lo plane `lo*` columns behind), k / v = the layer's key and value
// column blocks of the memory projection (T·nm rows, row stride ld_kv, kv_dt / lokv as above), LayerNorm gamma / beta → y (y_dt / ldy / loy),
// probs (T·lt, H, 4), mean / rstd (T·lt)
int svpc_cross_attn_ln_fwd(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                           int kv_dt, int ld_kv, int lokv, const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy,
                           float* probs, float* mean, float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                           const u64* seed, hipStream_t stream) {
    return svpc_cross_attn_ln_fwd_r(q, q_dt, ldq, loq, x1, x_dt, ldx, lox, k, v, kv_dt, ld_kv, lokv, gamma, beta, eps, y, y_dt, ldy, loy, probs, mean,
                                    rstd, T, lt, nm, D, H, scale, p_drop, site, seed, nullptr, nullptr, stream);
}
// the same over RAGGED sentences (valid tokens only): sentence s owns the rows [row_off[s], row_off[s] + row_len[s]) of q, x1, y, probs,
// mean, rstd, row_len[s] <= lt (lt: the padded sentence length — bounds the kernel's registers and is the stride of the dropout rows)
int svpc_cross_attn_ln_fwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy,
                             float* probs, float* mean, float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const u64* seed, const int* row_off, const int* row_len, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(svpc_cross_attn_ln_supported(D, H, lt, nm) == 1, "cross_attn_ln: unsupported shape");
    SVPC_REQUIRE(q_dt == x_dt && q_dt == kv_dt && q_dt == y_dt && q_dt >= 0 && q_dt <= 2, "cross_attn_ln: one storage kind for q, x1, k / v and y");
    SVPC_REQUIRE((row_off == nullptr) == (row_len == nullptr), "cross_attn_ln: row_off and row_len come together");
    XaArgs a{};
    a.row_off = row_off; a.row_len = row_len;
    a.q = q; a.q_dt = q_dt; a.ldq = ldq; a.loq = loq; a.x1 = x1; a.x_dt = x_dt; a.ldx = ldx; a.lox = lox; a.k = k; a.v = v; a.kv_dt = kv_dt;
    a.ld_kv = ld_kv; a.lokv = lokv; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = y; a.y_dt = y_dt; a.ldy = ldy; a.loy = loy;
    a.probs = probs; a.mean = mean; a.rstd = rstd; a.lt = lt; a.nm = nm; a.D = D; a.H = H; a.scale = scale; a.p_drop = p_drop; a.site = site;
    a.seed = seed;
    return xa_dispatch(a, T, false, stream);
}
// backward: dy (T·lt rows, dense; dy_dt 0 fp32 / 1 bf16) → dq and dres (gradients of the query rows and of the residual rows: dense, type
// dg_dt, row stride lddg), dk / dv (T·nm rows of type dkv_dt, row stride ld_dkv: e.g. the layer's column blocks of the memory projection's
// gradient buffer), part_ln (T, 2D): per-sentence partial sums of [dgamma | dbeta] for svpc_multi_finalize
int svpc_cross_attn_ln_bwd(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                           int kv_dt, int ld_kv, int lokv, const float* gamma, const float* probs, const float* mean, const float* rstd,
                           const void* dy, int dy_dt, int lddy, void* dq, void* dres, int dg_dt, int lddg, void* dk, void* dv, int dkv_dt,
                           int ld_dkv, float* part_ln, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                           const u64* seed, hipStream_t stream) {
    return svpc_cross_attn_ln_bwd_r(q, q_dt, ldq, loq, x1, x_dt, ldx, lox, k, v, kv_dt, ld_kv, lokv, gamma, probs, mean, rstd, dy, dy_dt, lddy, dq,
                                    dres, dg_dt, lddg, dk, dv, dkv_dt, ld_dkv, part_ln, T, lt, nm, D, H, scale, p_drop, site, seed, nullptr, nullptr,
                                    stream);
}
int svpc_cross_attn_ln_bwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* probs, const float* mean, const float* rstd,
                             const void* dy, int dy_dt, int lddy, void* dq, void* dres, int dg_dt, int lddg, void* dk, void* dv, int dkv_dt,
                             int ld_dkv, float* part_ln, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const u64* seed, const int* row_off, const int* row_len, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(svpc_cross_attn_ln_supported(D, H, lt, nm) == 1, "cross_attn_ln: unsupported shape");
    SVPC_REQUIRE((row_off == nullptr) == (row_len == nullptr), "cross_attn_ln: row_off and row_len come together");
    const int gdt = q_dt == 0 ? 0 : 1;
    SVPC_REQUIRE(q_dt == x_dt && q_dt == kv_dt && q_dt >= 0 && q_dt <= 2, "cross_attn_ln: one storage kind for q, x1 and k / v");
    SVPC_REQUIRE(dg_dt == gdt && dkv_dt == gdt && dy_dt == gdt, "cross_attn_ln: gradients are dense rows, fp32 for fp32 storage, bf16 otherwise");
    XaArgs a{};
    a.q = q; a.q_dt = q_dt; a.ldq = ldq; a.loq = loq; a.x1 = x1; a.x_dt = x_dt; a.ldx = ldx; a.lox = lox; a.k = k; a.v = v; a.kv_dt = kv_dt;
    a.ld_kv = ld_kv; a.lokv = lokv; a.gamma = gamma; a.probs = const_cast<float*>(probs); a.mean = const_cast<float*>(mean);
    a.rstd = const_cast<float*>(rstd); a.lt = lt; a.nm = nm; a.D = D; a.H = H; a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    a.dy = dy; a.dy_dt = dy_dt; a.lddy = lddy; a.dq = dq; a.dres = dres; a.dg_dt = dg_dt; a.lddg = lddg; a.dk = dk; a.dv = dv;
    a.dkv_dt = dkv_dt; a.ld_dkv = ld_dkv; a.part_ln = part_ln; a.row_off = row_off; a.row_len = row_len;
    return xa_dispatch(a, T, true, stream);
}

}  // extern "C"
