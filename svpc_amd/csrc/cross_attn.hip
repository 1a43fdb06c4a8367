// Decoder cross-attention over the ≤ 3 memory rows of a sentence, fused with its residual LayerNorm — forward and backward, one
// launch each per decoder layer (reference: src/rtransformer/model.py:657-658 inside BertDecoderLayerNoMemoryUntied.forward :630-663,
// the attention core :194-219, BertLayerNorm :143-156).
//
//   x2 = LayerNorm(x1 + MHA(query = x1·Wqᵀ + bq, keys / values = the sentence's n_mem memory rows))
//
// With n_mem ≤ 3 keys the QUERY PROJECTION never has to be formed: per head h,
//   score[t, j, h] = <Wq_h·x1[t] + bq_h, k[j, h]> / sqrt(dh) = (<x1[t], U[j, h]> + c[j, h]) / sqrt(dh),
//   U[j, h] = Wq_hᵀ·k[j, h]  (a D-vector per key and head: one small grouped GEMM over the n_mem·T memory rows for ALL layers, 7× fewer
//   FLOPs than projecting the T·Lt sentence rows),  c[j, h] = <bq_h, k[j, h]>.
// So the (T·Lt, D) query projection, its dgrad and its 4,224-row weight gradient disappear together with the attention launch and the
// LayerNorm launch: forward = this kernel; backward = this file's second kernel, which also returns dU — the weight / key gradients
// follow from it by two more small grouped GEMMs over the memory rows (dWq_h = Σ k[j,h] ⊗ dU[j,h], dk[j,h] = Wq_h·dU[j,h] + dc·bq_h).
//
// One workgroup per sentence, D threads (thread = one model column; a 64-lane wave = one head at dh = 64).  The Lt sentence rows live in
// LDS as fp32; scores are wave reductions, everything else is column-local.  Arithmetic is fp32 on the stored values (split rows enter as
// hi + lo): exact to rounding in every arithmetic mode.  Dropout of the attention probabilities uses the attention kernels' draw
// (common.h: attn_drop_scale, row = (sentence·H + head)·Lt + query, k = key), recomputed in backward.
#include "common.h"

namespace {

constexpr int XA_NM = 3;        // memory rows per sentence (vivt / viv: 3, vi: 2, v: 1)
constexpr int XA_LT = 32;       // sentence rows

struct XaArgs {
    const void* x1; int x_dt, ldx, lox;          // dt: 0 fp32, 1 bf16, 2 split (hi at col, lo at col + lo)
    const float* U;                              // (T·nm, H, D) of this layer
    const float* kv; int ld_kv;                  // this layer's [K | V] block of the memory projection: K at +0, V at +D
    const float* bq;
    const float* gamma; const float* beta; float eps;
    void* y; int y_dt, ldy, loy;
    float* probs;                                // (T·lt, H, 4) normalised probabilities BEFORE dropout (saved for backward)
    float* mean; float* rstd;                    // (T·lt)
    int lt, nm, D, H;
    float scale, p_drop; uint32_t site; const u64* seed;
    // backward
    const void* dy; int dy_dt, lddy;
    void* dx1; int dx_dt, lddx;
    float* dU;                                   // (T·nm, H, D)
    float* dkv; int ld_dkv;                      // [dK | dV] block (dK: the bias part dc·bq; the U part is added by the caller's GEMM)
    float* part_ln;                              // (T, 2D): per-sentence [dgamma | dbeta]
    float* part_bq;                              // (T, D): per-sentence d bq
};

__device__ __forceinline__ float xa_load(const void* p, int dt, size_t off, int lo) {
    if (dt == 0) return reinterpret_cast<const float*>(p)[off];
    const __bf16* b = reinterpret_cast<const __bf16*>(p);
    float v = (float)b[off];
    if (dt == 2) v += (float)b[off + lo];
    return v;
}
__device__ __forceinline__ void xa_store(void* p, int dt, size_t off, int lo, float v) {
    if (dt == 0) { reinterpret_cast<float*>(p)[off] = v; return; }
    __bf16* b = reinterpret_cast<__bf16*>(p);
    const __bf16 h = (__bf16)v;
    b[off] = h;
    if (dt == 2) b[off + lo] = (__bf16)(v - (float)h);
}
template <int DH> __device__ __forceinline__ float seg_sum(float v);
template <> __device__ __forceinline__ float seg_sum<64>(float v) { return wave_sum(v); }
template <> __device__ __forceinline__ float seg_sum<32>(float v) {      // sum over each 32-lane half (two heads per wave)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
    return v;
}

// LDS: xs[lt][D] + sc[lt][H][4] + cs[4][H]
template <int DH, int NPL>
__global__ __launch_bounds__(64 * NPL) void xattn_ln_fwd_kernel(XaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    const int D = a.D, H = a.H, lt = a.lt, nm = a.nm;
    float* xs = xsm;                           // lt × D
    float* sc = xs + (size_t)lt * D;           // lt × H × 4
    float* cs = sc + (size_t)lt * H * 4;       // 4 × H
    const int s = blockIdx.x, d = threadIdx.x;
    const int lane = d & 63, wave = d >> 6;
    constexpr int NW = NPL;
    // (1) the sentence's rows → LDS (fp32); (2) c[j, h] = <bq_h, k[j, h]>
    for (int t = 0; t < lt; ++t) xs[(size_t)t * D + d] = xa_load(a.x1, a.x_dt, (size_t)(s * lt + t) * a.ldx + d, a.lox);
    for (int i = d; i < nm * H; i += D) {
        const int j = i / H, h = i - j * H;
        const float* kr = a.kv + (size_t)(s * nm + j) * a.ld_kv + h * DH;
        float c = 0.f;
        for (int q = 0; q < DH; ++q) c += kr[q] * a.bq[h * DH + q];
        cs[j * H + h] = c;
    }
    __syncthreads();
    // (3) scores: a wave per head (dh = 32: the same, heads strided over the waves), lanes over the D columns of U[j, h]
    for (int h = wave; h < H; h += NW) {
        float uv[XA_NM][NPL];
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            const float* ur = a.U + ((size_t)(s * nm + min(j, nm - 1)) * H + h) * D;
#pragma unroll
            for (int i = 0; i < NPL; ++i) uv[j][i] = ur[lane + 64 * i];
        }
        for (int t = 0; t < lt; ++t) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const float xv = xs[(size_t)t * D + lane + 64 * i];
                p0 += xv * uv[0][i]; p1 += xv * uv[1][i]; p2 += xv * uv[2][i];
            }
            p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
            if (lane == 0) { float* o = sc + ((size_t)t * H + h) * 4; o[0] = p0; o[1] = p1; o[2] = p2; o[3] = 0.f; }
        }
    }
    __syncthreads();
    // (4) softmax over the nm keys, dropout of the probabilities
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    for (int i = d; i < lt * H; i += D) {
        const int t = i / H, h = i - t * H;
        float* o = sc + (size_t)i * 4;
        float av[XA_NM], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            av[j] = j < nm ? a.scale * (o[j] + cs[j * H + h]) : -INFINITY;
            m = fmaxf(m, av[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) { av[j] = j < nm ? expf(av[j] - m) : 0.f; sum += av[j]; }
        const float inv = 1.0f / sum;
        float* pr = a.probs + ((size_t)(s * lt + t) * H + h) * 4;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            const float p = av[j] * inv;
            pr[j] = p;
            float mult = 1.0f;
            if (a.p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, a.site, (u64)(s * H + h) * lt + t, (uint32_t)j, a.p_drop, inv_keep);
            o[j] = p * mult;
        }
        pr[3] = 0.f; o[3] = 0.f;
    }
    __syncthreads();
    // (5) attended vector + residual, column-local: xs[t][d] += Σ_j p̃[t, h(d), j]·v[j][d]
    {
        const int h = d / DH;
        float vv[XA_NM];
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) vv[j] = j < nm ? a.kv[(size_t)(s * nm + j) * a.ld_kv + D + d] : 0.f;
        for (int t = 0; t < lt; ++t) {
            const float4 p4 = *reinterpret_cast<const float4*>(sc + ((size_t)t * H + h) * 4);
            xs[(size_t)t * D + d] += p4.x * vv[0] + p4.y * vv[1] + p4.z * vv[2];
        }
    }
    __syncthreads();
    // (6) LayerNorm, a wave per row (two-pass variance, eps inside the square root: model.py:143-156)
    for (int t = wave; t < lt; t += NW) {
        float v[NPL], sum = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { v[i] = xs[(size_t)t * D + lane + 64 * i]; sum += v[i]; }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { const float c = v[i] - mean; sq += c * c; }
        const float var = wave_sum(sq) / (float)D;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        const size_t row = (size_t)(s * lt + t);
        if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            xa_store(a.y, a.y_dt, row * a.ldy + c, a.loy, (v[i] - mean) * rstd * a.gamma[c] + a.beta[c]);
        }
    }
}

// LDS: xs[lt][D] (x1 rows) + ds[lt][D] (g = dy·γ, then the pre-LayerNorm gradient) + pp, pt, dpt [lt][H][4] + gs[lt][3H] + red[NW][lt][2] +
// tot[lt][2] + dcs[4][H]
template <int DH, int NPL>
__global__ __launch_bounds__(64 * NPL) void xattn_ln_bwd_kernel(XaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    constexpr int NW = NPL, H = NPL * 64 / DH, JH = XA_NM * H;
    const int D = a.D, lt = a.lt, nm = a.nm;
    float* xs = xsm;                               // lt × D
    float* ds = xs + (size_t)lt * D;               // lt × D
    float* pp = ds + (size_t)lt * D;               // lt × H × 4 : p
    float* pt = pp + (size_t)lt * H * 4;           // lt × H × 4 : p̃ = p·dropout
    float* dpt = pt + (size_t)lt * H * 4;          // lt × H × 4 : d p̃
    float* gs = dpt + (size_t)lt * H * 4;          // lt × JH : scale·ds, index j·H + h
    float* red = gs + (size_t)lt * JH;             // NW × lt × 2
    float* tot = red + (size_t)NW * lt * 2;        // lt × 2
    float* dcs = tot + (size_t)lt * 2;             // XA_NM × H
    const int s = blockIdx.x, d = threadIdx.x;
    const int lane = d & 63, wave = d >> 6;
    const int h = d / DH;
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    for (int t = 0; t < lt; ++t) xs[(size_t)t * D + d] = xa_load(a.x1, a.x_dt, (size_t)(s * lt + t) * a.ldx + d, a.lox);
    for (int i = d; i < lt * H; i += D) {
        const int t = i / H, hh = i - t * H;
        const float* pr = a.probs + ((size_t)(s * lt + t) * H + hh) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = pr[j];
            float mult = 1.0f;
            if (a.p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, a.site, (u64)(s * H + hh) * lt + t, (uint32_t)j, a.p_drop, inv_keep);
            pp[(size_t)i * 4 + j] = p;
            pt[(size_t)i * 4 + j] = j < nm ? p * mult : 0.f;
        }
    }
    float vv[XA_NM], kk[XA_NM];
#pragma unroll
    for (int j = 0; j < XA_NM; ++j) {
        const float* r = a.kv + (size_t)(s * nm + min(j, nm - 1)) * a.ld_kv;
        vv[j] = j < nm ? r[D + d] : 0.f;
        kk[j] = j < nm ? r[d] : 0.f;
    }
    const float gam = a.gamma[d];
    __syncthreads();
    // (A1) LayerNorm backward, pass 1 (column-local): x̂ recomputed from x1 + Σ p̃·v; g = dy·γ kept in LDS; row sums Σ g, Σ g·x̂ by waves;
    //      dγ / dβ partial sums of the sentence
    float dgam = 0.f, dbet = 0.f;
    for (int t = 0; t < lt; ++t) {
        const size_t row = (size_t)(s * lt + t);
        const float4 p4 = *reinterpret_cast<const float4*>(pt + ((size_t)t * H + h) * 4);
        const float yv = xs[(size_t)t * D + d] + p4.x * vv[0] + p4.y * vv[1] + p4.z * vv[2];
        const float xh = (yv - a.mean[row]) * a.rstd[row];
        const float dyv = xa_load(a.dy, a.dy_dt, row * a.lddy + d, 0);
        dgam += dyv * xh; dbet += dyv;
        const float g = dyv * gam;
        ds[(size_t)t * D + d] = g;
        const float s1 = wave_sum(g), s2 = wave_sum(g * xh);
        if (lane == 0) { red[((size_t)wave * lt + t) * 2] = s1; red[((size_t)wave * lt + t) * 2 + 1] = s2; }
    }
    a.part_ln[(size_t)s * 2 * D + d] = dgam;
    a.part_ln[(size_t)s * 2 * D + D + d] = dbet;
    __syncthreads();
    for (int i = d; i < lt * 2; i += D) {
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += red[(size_t)w * lt * 2 + i];
        tot[i] = v;
    }
    __syncthreads();
    // (A2) pass 2: dpre = rstd·(g − Σg/D − x̂·Σ(g·x̂)/D) → ds;  (B) d p̃[t, h, j] = Σ_{d in head} dpre·v[j] (segment sums), dV column sums
    float dv[XA_NM] = {0.f, 0.f, 0.f};
    const float invD = 1.0f / (float)D;
    for (int t = 0; t < lt; ++t) {
        const size_t row = (size_t)(s * lt + t);
        const float4 p4 = *reinterpret_cast<const float4*>(pt + ((size_t)t * H + h) * 4);
        const float yv = xs[(size_t)t * D + d] + p4.x * vv[0] + p4.y * vv[1] + p4.z * vv[2];
        const float rs = a.rstd[row];
        const float xh = (yv - a.mean[row]) * rs;
        const float dp = rs * (ds[(size_t)t * D + d] - tot[2 * t] * invD - xh * tot[2 * t + 1] * invD);
        ds[(size_t)t * D + d] = dp;
        dv[0] += p4.x * dp; dv[1] += p4.y * dp; dv[2] += p4.z * dp;
        const float q0 = seg_sum<DH>(dp * vv[0]), q1 = seg_sum<DH>(dp * vv[1]), q2 = seg_sum<DH>(dp * vv[2]);
        if ((d & (DH - 1)) == 0) { float* o = dpt + ((size_t)t * H + h) * 4; o[0] = q0; o[1] = q1; o[2] = q2; o[3] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < XA_NM; ++j)
        if (j < nm) a.dkv[(size_t)(s * nm + j) * a.ld_dkv + D + d] = dv[j];
    __syncthreads();
    // (C) softmax backward per (row, head): g[t, j, h] = scale·p_j·(dp_j − Σ_k p_k·dp_k), dp_j = d p̃_j · dropout multiplier
    for (int i = d; i < lt * H; i += D) {
        const int t = i / H, hh = i - t * H;
        float dp[XA_NM], dot = 0.f;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            float mult = 1.0f;
            if (a.p_drop > 0.f && j < nm) mult = attn_drop_scale(seed, a.site, (u64)(s * H + hh) * lt + t, (uint32_t)j, a.p_drop, inv_keep);
            dp[j] = j < nm ? dpt[(size_t)i * 4 + j] * mult : 0.f;
            dot += pp[(size_t)i * 4 + j] * dp[j];
        }
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) gs[(size_t)t * JH + j * H + hh] = j < nm ? a.scale * pp[(size_t)i * 4 + j] * (dp[j] - dot) : 0.f;
    }
    __syncthreads();
    for (int i = d; i < XA_NM * H; i += D) {         // dc[j, h] = Σ_t g[t, j, h]
        float c = 0.f;
        for (int t = 0; t < lt; ++t) c += gs[(size_t)t * JH + i];
        dcs[i] = c;
    }
    __syncthreads();
    {   // bias paths, column-local: dk[j][d] (the c = <bq, k> part) = dc[j, h]·bq[d];  d bq[d] partial = Σ_j dc[j, h]·k[j][d]
        const float bqd = a.bq[d];
        float pb = 0.f;
#pragma unroll
        for (int j = 0; j < XA_NM; ++j) {
            if (j < nm) {
                const float c = dcs[j * H + h];
                a.dkv[(size_t)(s * nm + j) * a.ld_dkv + d] = c * bqd;
                pb += c * kk[j];
            }
        }
        a.part_bq[(size_t)s * D + d] = pb;
    }
    // (D) column phase: dx1[t][d] = dpre[t][d] + Σ_{j,h} g[t, j, h]·U[j, h][d];   dU[j, h][d] = Σ_t g[t, j, h]·x1[t][d]
    float u[JH], du[JH];
#pragma unroll
    for (int i = 0; i < JH; ++i) {
        const int j = i / H, hh = i - j * H;
        u[i] = j < nm ? a.U[((size_t)(s * nm + j) * H + hh) * D + d] : 0.f;
        du[i] = 0.f;
    }
    for (int t = 0; t < lt; ++t) {
        float acc = ds[(size_t)t * D + d];
        const float xv = xs[(size_t)t * D + d];
        const float* g = gs + (size_t)t * JH;
#pragma unroll
        for (int q = 0; q < JH / 4; ++q) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + 4 * q);
            acc += g4.x * u[4 * q] + g4.y * u[4 * q + 1] + g4.z * u[4 * q + 2] + g4.w * u[4 * q + 3];
            du[4 * q] += g4.x * xv; du[4 * q + 1] += g4.y * xv; du[4 * q + 2] += g4.z * xv; du[4 * q + 3] += g4.w * xv;
        }
        xa_store(a.dx1, a.dx_dt, (size_t)(s * lt + t) * a.lddx + d, 0, acc);
    }
#pragma unroll
    for (int i = 0; i < JH; ++i) {
        const int j = i / H, hh = i - j * H;
        if (j < nm) a.dU[((size_t)(s * nm + j) * H + hh) * D + d] = du[i];
    }
}

size_t xa_fwd_lds(int lt, int D, int H) { return ((size_t)lt * D + (size_t)lt * H * 4 + 4 * H) * sizeof(float); }
size_t xa_bwd_lds(int lt, int D, int H, int NW) {
    return ((size_t)2 * lt * D + (size_t)3 * lt * H * 4 + (size_t)lt * XA_NM * H + (size_t)NW * lt * 2 + (size_t)lt * 2 + XA_NM * H) * sizeof(float);
}

template <int DH, int NPL>
int xa_launch(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    const int NW = NPL;
    const size_t lds = bwd ? xa_bwd_lds(a.lt, a.D, a.H, NW) : xa_fwd_lds(a.lt, a.D, a.H);
    if (lds > 160 * 1024) { svpc_set_error("cross_attn_ln: the sentence rows do not fit LDS"); return -1; }
    const void* fn = bwd ? (const void*)xattn_ln_bwd_kernel<DH, NPL> : (const void*)xattn_ln_fwd_kernel<DH, NPL>;
    int rc = svpc_raise_lds_once(fn, "cross_attn_ln");
    if (rc) return rc;
    if (bwd) hipLaunchKernelGGL((xattn_ln_bwd_kernel<DH, NPL>), dim3(T), dim3(64 * NPL), lds, s, a);
    else hipLaunchKernelGGL((xattn_ln_fwd_kernel<DH, NPL>), dim3(T), dim3(64 * NPL), lds, s, a);
    return svpc_check_launch(bwd ? "cross_attn_ln_bwd" : "cross_attn_ln_fwd");
}
int xa_dispatch(const XaArgs& a, int T, bool bwd, hipStream_t s) {
    const int dh = a.D / a.H, npl = a.D / 64;
    if (dh == 64 && npl == 12) return xa_launch<64, 12>(a, T, bwd, s);
    if (dh == 64 && npl == 8) return xa_launch<64, 8>(a, T, bwd, s);
    if (dh == 64 && npl == 4) return xa_launch<64, 4>(a, T, bwd, s);
    if (dh == 32 && npl == 2) return xa_launch<32, 2>(a, T, bwd, s);
    if (dh == 32 && npl == 4) return xa_launch<32, 4>(a, T, bwd, s);
    svpc_set_error("cross_attn_ln: unsupported (hidden size, heads)");
    return -1;
}

}  // namespace

extern "C" {

// 1 if the fused cross-attention + LayerNorm kernels take this shape: D ∈ {256, 512, 768} with 64-wide heads or D ∈ {128, 256} with
// 32-wide heads, ≤ 32 sentence rows, ≤ 3 memory rows
int svpc_cross_attn_ln_supported(int D, int H, int lt, int nm) {
    if (H <= 0 || D % H || D % 64 || lt < 1 || lt > XA_LT || nm < 1 || nm > XA_NM) return 0;
    const int dh = D / H, npl = D / 64;
    const bool shape = (dh == 64 && (npl == 12 || npl == 8 || npl == 4)) || (dh == 32 && (npl == 2 || npl == 4));
    if (!shape || H % 4) return 0;
    return xa_bwd_lds(lt, D, H, npl) <= 160 * 1024 ? 1 : 0;
}

// forward: x1 (T·lt rows; x_dt 0 fp32 / 1 bf16 / 2 split with lo plane `lox` columns behind), U (T·nm, H, D) fp32, kv = the layer's
// [K | V] block of the memory projection (fp32, row stride ld_kv), bq, LayerNorm gamma / beta → y (same row count; y_dt / ldy / loy),
// probs (T·lt, H, 4), mean / rstd (T·lt)
int svpc_cross_attn_ln_fwd(const void* x1, int x_dt, int ldx, int lox, const float* U, const float* kv, int ld_kv, const float* bq,
                           const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy, float* probs, float* mean,
                           float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site, const u64* seed,
                           hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(svpc_cross_attn_ln_supported(D, H, lt, nm) == 1, "cross_attn_ln: unsupported shape");
    SVPC_REQUIRE((((uintptr_t)U) & 15) == 0, "cross_attn_ln: 16-byte aligned U");
    XaArgs a{};
    a.x1 = x1; a.x_dt = x_dt; a.ldx = ldx; a.lox = lox; a.U = U; a.kv = kv; a.ld_kv = ld_kv; a.bq = bq; a.gamma = gamma; a.beta = beta;
    a.eps = eps; a.y = y; a.y_dt = y_dt; a.ldy = ldy; a.loy = loy; a.probs = probs; a.mean = mean; a.rstd = rstd; a.lt = lt; a.nm = nm;
    a.D = D; a.H = H; a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    return xa_dispatch(a, T, false, stream);
}
// backward: dy (T·lt rows, dense; dy_dt 0 fp32 / 1 bf16) → dx1 (dense, dx_dt), dU (T·nm, H, D), dkv = the [dK | dV] block of the memory
// projection's gradient (dV complete; dK = the bias part dc·bq — the caller's grouped GEMM adds Wq_h·dU), part_ln (T, 2D) and part_bq (T, D):
// per-sentence partial sums of [dgamma | dbeta] and d bq for the table-driven finalizer
int svpc_cross_attn_ln_bwd(const void* x1, int x_dt, int ldx, int lox, const float* U, const float* kv, int ld_kv, const float* bq,
                           const float* gamma, const float* probs, const float* mean, const float* rstd, const void* dy, int dy_dt,
                           int lddy, void* dx1, int dx_dt, int lddx, float* dU, float* dkv, int ld_dkv, float* part_ln, float* part_bq, int T,
                           int lt, int nm, int D, int H, float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(svpc_cross_attn_ln_supported(D, H, lt, nm) == 1, "cross_attn_ln: unsupported shape");
    XaArgs a{};
    a.x1 = x1; a.x_dt = x_dt; a.ldx = ldx; a.lox = lox; a.U = U; a.kv = kv; a.ld_kv = ld_kv; a.bq = bq; a.gamma = gamma;
    a.probs = const_cast<float*>(probs); a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.lt = lt; a.nm = nm; a.D = D;
    a.H = H; a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed; a.dy = dy; a.dy_dt = dy_dt; a.lddy = lddy; a.dx1 = dx1;
    a.dx_dt = dx_dt; a.lddx = lddx; a.dU = dU; a.dkv = dkv; a.ld_dkv = ld_dkv; a.part_ln = part_ln; a.part_bq = part_bq;
    return xa_dispatch(a, T, true, stream);
}

}  // extern "C"
