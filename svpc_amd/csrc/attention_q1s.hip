// One-query attention over bf16 / split-stored keys and values, forward and backward — the last clip-encoder layer of the training
// forward, which is evaluated only for the [CLS] row of every clip (reference: src/rtransformer/model.py:1062-1064 consumes nothing
// else of that layer; core :194-219): ONE fp32 query per (sequence, head) against ≤ 128 keys whose K | V rows are the projection's
// output in place (bf16, or split16 = hi + lo planes in the bf16x3 mode).
//
// A 100-key × 64-column problem per (sequence, head) is ≈25 KB of K / V and 13 KFLOP: no tiles, no LDS, no matrix cores — one wave per
// pair, exact fp32 arithmetic on the values read.  Scores with a KEY per lane (each lane's dot product walks its own 128-byte row pieces),
// the weighted sum and the query gradient with a COLUMN PAIR per lane and two keys in flight per wave-instruction (lanes 0-31 even keys,
// 32-63 odd keys; rows are read as whole 128-byte lines).  HBM-bound: forward reads K, V once; backward reads K, V and writes dK, dV once.
// Dropout draws as every other attention kernel (element index ((s·H + h)·max_q + 0)·max_k + key).
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Q1sArgs {
    const float* Q; int ldq; const __bf16* K; int ldk; const __bf16* V; int ldv; int k_lo, v_lo;      // *_lo: lo-plane offsets (0 = plain bf16)
    float* O; int ldo; float* LSE; const int* seq; int n_seq, H, max_q, max_k;
    const float* key_mask; float scale; float p_drop; uint32_t site; const u64* seed;
    const float* dO; int lddo; float* dQ; int lddq; __bf16* dK; int lddk; __bf16* dV; int lddv;
};

template <bool SPLIT>
__device__ __forceinline__ void q1s_row8(const __bf16* p, int lo, float* v) {      // 8 consecutive elements of a row, fp32 (hi + lo if split)
    const bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    if (SPLIT) {
        const bf16x8 l = *reinterpret_cast<const bf16x8*>(p + lo);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
    }
}
template <bool SPLIT>
__device__ __forceinline__ float2 q1s_pair(const __bf16* p, int lo) {              // 2 consecutive elements
    const bf16x2 h = *reinterpret_cast<const bf16x2*>(p);
    float2 r = make_float2((float)h[0], (float)h[1]);
    if (SPLIT) {
        const bf16x2 l = *reinterpret_cast<const bf16x2*>(p + lo);
        r.x += (float)l[0]; r.y += (float)l[1];
    }
    return r;
}
__device__ __forceinline__ float q1s_drop(const Q1sArgs& a, u64 seed, u64 row, int key) {          // row = index of the probability row (common.h)
    if (a.p_drop <= 0.f) return 1.0f;
    return attn_drop_scale(seed, a.site, row, (uint32_t)key, a.p_drop, 1.0f / (1.0f - a.p_drop));
}

// scores of this lane's keys (lane, lane + 64) → normalised probabilities p[2] (0 past k_len), returns the log-sum-exp
template <int DH, bool SPLIT>
__device__ __forceinline__ float q1s_probs(const Q1sArgs& a, const float* __restrict__ qv, const __bf16* Kp, int k_len, const float* km,
                                           int lane, float (&p)[2]) {
    float sc[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = lane + 64 * kk;
        sc[kk] = -INFINITY;
        if (k < k_len) {
            const __bf16* row = Kp + (size_t)k * a.ldk;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                float v[8];
                q1s_row8<SPLIT>(row + 8 * c, a.k_lo, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = fmaf(qv[8 * c + j], v[j], dot);
            }
            sc[kk] = dot * a.scale + (1.0f - (km ? km[k] : 1.0f)) * -10000.0f;
        }
    }
    const float mx = wave_max(fmaxf(sc[0], sc[1]));
    p[0] = sc[0] > -INFINITY ? expf(sc[0] - mx) : 0.f;
    p[1] = sc[1] > -INFINITY ? expf(sc[1] - mx) : 0.f;
    const float sum = wave_sum(p[0] + p[1]);
    const float inv = 1.0f / sum;
    p[0] *= inv; p[1] *= inv;
    return mx + logf(sum);
}

// acc[c] += Σ_k w[k]·X[k][c0 + c] with a column pair (DH = 64) / one column (DH = 32) per lane, lanes 0-31 the even keys, 32-63 the odd
// keys; w[k] lives in lane k & 63, slot k >> 6 of `w`.  The row pieces of EIGHT key pairs are requested before the first is consumed:
// walking the keys with the load inside the loop body made every pair a memory round trip of its own (50 in a row for 100 keys).
template <int DH, bool SPLIT>
__device__ __forceinline__ void q1s_weighted_sum(const __bf16* __restrict__ X, int ldx, int x_lo, int k_len, const float (&w)[2], int lane,
                                                 float* acc) {
    constexpr int CPL = DH / 32, UN = 8;
    const int half = lane >> 5, c0 = (lane & 31) * CPL;
    for (int k2 = 0; k2 < k_len; k2 += 2 * UN) {
        float2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = min(k2 + 2 * u + half, k_len - 1);          // (clamped: the weight of a key past the end is zeroed below)
            const __bf16* row = X + (size_t)kc * ldx + c0;
            if (CPL == 2) v[u] = q1s_pair<SPLIT>(row, x_lo);
            else { float t = (float)row[0]; if (SPLIT) t += (float)row[x_lo]; v[u] = make_float2(t, 0.f); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int ke = k2 + 2 * u, k = ke + half;
            const float pe0 = __shfl(w[0], ke & 63, 64), po0 = __shfl(w[0], (ke + 1) & 63, 64);
            const float pe1 = __shfl(w[1], ke & 63, 64), po1 = __shfl(w[1], (ke + 1) & 63, 64);
            float wk = half ? ((ke + 1) >> 6 ? po1 : po0) : (ke >> 6 ? pe1 : pe0);
            if (k >= k_len) wk = 0.f;
            acc[0] = fmaf(wk, v[u].x, acc[0]);
            if (CPL == 2) acc[CPL - 1] = fmaf(wk, v[u].y, acc[CPL - 1]);
        }
    }
}

template <int DH, bool SPLIT>
__global__ __launch_bounds__(256) void attn_q1s_fwd_kernel(Q1sArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sh = (int)blockIdx.x * 4 + wave;
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const float* qp = a.Q + (size_t)q_off * a.ldq + h * DH;
    float qv[DH];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
        qv[4 * c] = t.x; qv[4 * c + 1] = t.y; qv[4 * c + 2] = t.z; qv[4 * c + 3] = t.w;
    }
    const __bf16* Kp = a.K + (size_t)k_off * a.ldk + h * DH;
    const __bf16* Vp = a.V + (size_t)k_off * a.ldv + h * DH;
    float p[2];
    const float lse = q1s_probs<DH, SPLIT>(a, qv, Kp, k_len, a.key_mask ? a.key_mask + k_off : nullptr, lane, p);
    if (lane == 0 && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q] = lse;
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const u64 row_base = (u64)(s * a.H + h) * a.max_q;        // the pair's single probability row
    p[0] *= q1s_drop(a, seed, row_base, lane);
    p[1] *= q1s_drop(a, seed, row_base, lane + 64);
    // O[d] = Σ_k p̃[k]·V[k][d]: lanes 0-31 take even keys, lanes 32-63 odd keys, each lane a column pair (DH = 64) / one column (DH = 32)
    constexpr int CPL = DH / 32;                  // columns per lane
    const int half = lane >> 5, c0 = (lane & 31) * CPL;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    q1s_weighted_sum<DH, SPLIT>(Vp, a.ldv, a.v_lo, k_len, p, lane, acc);
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] += __shfl_xor(acc[j], 32, 64);
    if (half == 0) {
        float* op = a.O + (size_t)q_off * a.ldo + h * DH + c0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) op[j] = acc[j];
    }
}

template <int DH, bool SPLIT>
__global__ __launch_bounds__(256) void attn_q1s_bwd_kernel(Q1sArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sh = (int)blockIdx.x * 4 + wave;
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const float* qp = a.Q + (size_t)q_off * a.ldq + h * DH;
    const float* gp = a.dO + (size_t)q_off * a.lddo + h * DH;
    float qv[DH], gv[DH];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
        const float4 u = *reinterpret_cast<const float4*>(gp + 4 * c);
        qv[4 * c] = t.x; qv[4 * c + 1] = t.y; qv[4 * c + 2] = t.z; qv[4 * c + 3] = t.w;
        gv[4 * c] = u.x; gv[4 * c + 1] = u.y; gv[4 * c + 2] = u.z; gv[4 * c + 3] = u.w;
    }
    const __bf16* Kp = a.K + (size_t)k_off * a.ldk + h * DH;
    const __bf16* Vp = a.V + (size_t)k_off * a.ldv + h * DH;
    float p[2];
    q1s_probs<DH, SPLIT>(a, qv, Kp, k_len, a.key_mask ? a.key_mask + k_off : nullptr, lane, p);
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const u64 row_base = (u64)(s * a.H + h) * a.max_q;        // the pair's single probability row
    float m[2], dpt[2];
    // dP̃[k] = m[k]·(dO·V[k]) with this lane's keys' V rows; dV[k] = p[k]·m[k]·dO (row store)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = lane + 64 * kk;
        m[kk] = q1s_drop(a, seed, row_base, k);
        dpt[kk] = 0.f;
        if (k < k_len) {
            const __bf16* row = Vp + (size_t)k * a.ldv;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                float v[8];
                q1s_row8<SPLIT>(row + 8 * c, a.v_lo, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = fmaf(gv[8 * c + j], v[j], dot);
            }
            dpt[kk] = m[kk] * dot;
            const float w = p[kk] * m[kk];
            __bf16* dvr = a.dV + (size_t)(k_off + k) * a.lddv + h * DH;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)(w * gv[8 * c + j]);
                *reinterpret_cast<bf16x8*>(dvr + 8 * c) = o;
            }
        }
    }
    const float delta = wave_sum(p[0] * dpt[0] + p[1] * dpt[1]);
    float ds[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = lane + 64 * kk;
        ds[kk] = p[kk] * (dpt[kk] - delta) * a.scale;           // gradient of the UNSCALED q·k product
        if (k < k_len) {
            __bf16* dkr = a.dK + (size_t)(k_off + k) * a.lddk + h * DH;
#pragma unroll
            for (int c = 0; c < DH / 8; ++c) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)(ds[kk] * qv[8 * c + j]);
                *reinterpret_cast<bf16x8*>(dkr + 8 * c) = o;
            }
        }
    }
    // dQ[d] = Σ_k dS[k]·K[k][d]: column pair per lane, two keys in flight (as the forward's weighted sum)
    constexpr int CPL = DH / 32;
    const int half = lane >> 5, c0 = (lane & 31) * CPL;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
    q1s_weighted_sum<DH, SPLIT>(Kp, a.ldk, a.k_lo, k_len, ds, lane, acc);
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] += __shfl_xor(acc[j], 32, 64);
    if (half == 0) {
        float* dq = a.dQ + (size_t)q_off * a.lddq + h * DH + c0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) dq[j] = acc[j];
    }
}

// ---- head dim 64, rows read as whole lines (round 5).  The kernels above give every lane a key ROW: a 16-byte load instruction then
// touches 64 different 128-byte lines and eight instructions are needed to use each of them — the CU's address path, not HBM, bounds the
// launch (forward 40 µs, backward 80 µs for 59 MB read / 118 MB moved at the headline shape).  Here a row is read by EIGHT lanes
// (lane = 8·g + c8: row group g, 16-byte piece c8), one instruction covers 8 consecutive keys as 8 whole lines, 13 instructions the 100
// keys; a key's dot product is finished over its 8 lanes with three DPP adds, sums over the keys over the 8 row groups (lane ^ 8, 16, 32).
// dK / dV rows leave the same way (whole lines).  Same arithmetic as above (fp32 on the values read), same dropout draws.
template <bool SPLIT>
__device__ __forceinline__ void q1r_row(const __bf16* p, int lo, float (&v)[8]) {
    const bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
    if (SPLIT) {
        const bf16x8 l = *reinterpret_cast<const bf16x8*>(p + lo);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    }
}
__device__ __forceinline__ float q1r_sum8(float v) {          // over the 8 lanes of a row
    v = dpp_add_<0xB1>(v); v = dpp_add_<0x4E>(v); v = dpp_add_<0x141>(v);
    return v;
}
__device__ __forceinline__ float q1r_sum_groups(float v) {    // over the 8 row groups (lanes with equal lane & 7)
    v += dpp_mov_<0x128>(v);                                   // lane ^ 8 (row_ror:8)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float q1r_max_groups(float v) {
    v = fmaxf(v, dpp_mov_<0x128>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
constexpr int Q1R_NI = 16;          // 16 instructions × 8 keys = 128 keys

// probabilities of this lane's keys (key 8·i + g in slot i; 0 past k_len), every lane of a row holds the same values; returns the LSE
template <bool SPLIT>
__device__ __forceinline__ float q1r_probs(const Q1sArgs& a, const float (&q)[8], const __bf16* Kp, int k_off, int k_len, int g,
                                           float (&p)[Q1R_NI]) {
    float mk[Q1R_NI];
#pragma unroll
    for (int i = 0; i < Q1R_NI; ++i) mk[i] = 0.f;
    if (a.key_mask) {
#pragma unroll
        for (int i = 0; i < Q1R_NI; ++i) mk[i] = (1.0f - a.key_mask[k_off + min(8 * i + g, k_len - 1)]) * -10000.0f;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 || k_len > 64) {
            float kv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q1r_row<SPLIT>(Kp + (size_t)min(8 * (8 * b + u) + g, k_len - 1) * a.ldk, a.k_lo, kv[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = fmaf(q[j], kv[u][j], dot);
                dot = q1r_sum8(dot);
                p[8 * b + u] = (8 * (8 * b + u) + g) < k_len ? dot * a.scale + mk[8 * b + u] : -INFINITY;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[8 * b + u] = -INFINITY;
        }
    }
    float mx = p[0];
#pragma unroll
    for (int i = 1; i < Q1R_NI; ++i) mx = fmaxf(mx, p[i]);
    mx = q1r_max_groups(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < Q1R_NI; ++i) { p[i] = p[i] > -INFINITY ? expf(p[i] - mx) : 0.f; sum += p[i]; }
    sum = q1r_sum_groups(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < Q1R_NI; ++i) p[i] *= inv;
    return mx + logf(sum);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void attn_q1r_fwd_kernel(Q1sArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sh = (int)blockIdx.x * 4 + wave;
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const int g = lane >> 3, c8 = lane & 7;
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    float q[8];
    {
        const float* qp = a.Q + (size_t)q_off * a.ldq + h * 64 + 8 * c8;
        const float4 t0 = *reinterpret_cast<const float4*>(qp), t1 = *reinterpret_cast<const float4*>(qp + 4);
        q[0] = t0.x; q[1] = t0.y; q[2] = t0.z; q[3] = t0.w; q[4] = t1.x; q[5] = t1.y; q[6] = t1.z; q[7] = t1.w;
    }
    const __bf16* Kp = a.K + (size_t)k_off * a.ldk + h * 64 + 8 * c8;
    const __bf16* Vp = a.V + (size_t)k_off * a.ldv + h * 64 + 8 * c8;
    float p[Q1R_NI];
    const float lse = q1r_probs<SPLIT>(a, q, Kp, k_off, k_len, g, p);
    if (lane == 0 && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q] = lse;
    const u64 row_base = (u64)(s * a.H + h) * a.max_q;
#pragma unroll
    for (int i = 0; i < Q1R_NI; ++i) p[i] *= q1s_drop(a, seed, row_base, 8 * i + g);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 || k_len > 64) {
            float vv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q1r_row<SPLIT>(Vp + (size_t)min(8 * (8 * b + u) + g, k_len - 1) * a.ldv, a.v_lo, vv[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(p[8 * b + u], vv[u][j], acc[j]);          // (p is 0 past k_len)
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = q1r_sum_groups(acc[j]);
    if (g == 0) {
        float* op = a.O + (size_t)q_off * a.ldo + h * 64 + 8 * c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) op[j] = acc[j];
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void attn_q1r_bwd_kernel(Q1sArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sh = (int)blockIdx.x * 4 + wave;
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const int g = lane >> 3, c8 = lane & 7;
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    float q[8], go[8];
    {
        const float* qp = a.Q + (size_t)q_off * a.ldq + h * 64 + 8 * c8;
        const float* gp = a.dO + (size_t)q_off * a.lddo + h * 64 + 8 * c8;
        const float4 t0 = *reinterpret_cast<const float4*>(qp), t1 = *reinterpret_cast<const float4*>(qp + 4);
        const float4 u0 = *reinterpret_cast<const float4*>(gp), u1 = *reinterpret_cast<const float4*>(gp + 4);
        q[0] = t0.x; q[1] = t0.y; q[2] = t0.z; q[3] = t0.w; q[4] = t1.x; q[5] = t1.y; q[6] = t1.z; q[7] = t1.w;
        go[0] = u0.x; go[1] = u0.y; go[2] = u0.z; go[3] = u0.w; go[4] = u1.x; go[5] = u1.y; go[6] = u1.z; go[7] = u1.w;
    }
    const __bf16* Kp = a.K + (size_t)k_off * a.ldk + h * 64 + 8 * c8;
    const __bf16* Vp = a.V + (size_t)k_off * a.ldv + h * 64 + 8 * c8;
    float p[Q1R_NI];
    q1r_probs<SPLIT>(a, q, Kp, k_off, k_len, g, p);
    const u64 row_base = (u64)(s * a.H + h) * a.max_q;
    // dP̃[k] = m[k]·(dO·V[k]);  dV[k] = p[k]·m[k]·dO (whole rows)
    float dpt[Q1R_NI];
    float part = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 || k_len > 64) {
            float vv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q1r_row<SPLIT>(Vp + (size_t)min(8 * (8 * b + u) + g, k_len - 1) * a.ldv, a.v_lo, vv[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 8 * b + u, k = 8 * i + g;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dot = fmaf(go[j], vv[u][j], dot);
                dot = q1r_sum8(dot);
                const float m = q1s_drop(a, seed, row_base, k);
                dpt[i] = m * dot;
                part = fmaf(p[i], dpt[i], part);                          // (p is 0 past k_len)
                if (k < k_len) {
                    const float w = p[i] * m;
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (__bf16)(w * go[j]);
                    *reinterpret_cast<bf16x8*>(a.dV + (size_t)(k_off + k) * a.lddv + h * 64 + 8 * c8) = o;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) dpt[8 * b + u] = 0.f;
        }
    }
    const float delta = q1r_sum_groups(part);            // every lane of a row holds the row's term: the sum runs over the row groups only
    // dS[k] = p[k]·(dP̃[k] − δ)·scale;  dK[k] = dS[k]·q (whole rows);  dQ = Σ_k dS[k]·K[k]
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 || k_len > 64) {
            float kv[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q1r_row<SPLIT>(Kp + (size_t)min(8 * (8 * b + u) + g, k_len - 1) * a.ldk, a.k_lo, kv[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 8 * b + u, k = 8 * i + g;
                const float ds = p[i] * (dpt[i] - delta) * a.scale;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(ds, kv[u][j], acc[j]);
                if (k < k_len) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (__bf16)(ds * q[j]);
                    *reinterpret_cast<bf16x8*>(a.dK + (size_t)(k_off + k) * a.lddk + h * 64 + 8 * c8) = o;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = q1r_sum_groups(acc[j]);
    if (g == 0) {
        float* dq = a.dQ + (size_t)q_off * a.lddq + h * 64 + 8 * c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) dq[j] = acc[j];
    }
}

static bool q1r_on() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SVPC_Q1R"); on = e ? atoi(e) : 1; }
    return on != 0;
}

static bool q1s_ok(int dh, int max_k, int ldq, int ldk, int ldv, int k_lo, int v_lo, const void* Q, const void* K, const void* V) {
    return (dh == 64 || dh == 32) && max_k >= 1 && max_k <= 128 && ldq % 4 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && k_lo % 8 == 0 && v_lo % 8 == 0 &&
           ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V)) & 15) == 0;
}

extern "C" {

// 1 if the one-query kernels take this problem (head dim 32 / 64, ≤ 128 keys, 16-byte aligned row pieces)
int svpc_attn_q1s_supported(int dh, int max_k, int ldq, int ldk, int ldv, int k_lo, int v_lo) {
    return q1s_ok(dh, max_k, ldq, ldk, ldv, k_lo, v_lo, nullptr, nullptr, nullptr) ? 1 : 0;
}

// ONE fp32 query per sequence (seq's q_len must be 1) against bf16 keys / values (k_lo = v_lo = 0) or split ones (lo planes k_lo / v_lo
// elements behind the hi planes); O, LSE fp32.  Non-causal (a single query sees every key), key-pad mask, dropout.
int svpc_attn_q1s_fwd(const float* Q, int ldq, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, float* O, int ldo, float* LSE,
                      const int* seq, int n_seq, int H, int dh, int max_k, const float* key_mask, float scale, float p_drop, unsigned site,
                      const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(q1s_ok(dh, max_k, ldq, ldk, ldv, k_lo, v_lo, Q, K, V) && ldo % 2 == 0, "attn_q1s: unsupported shape / alignment");
    SVPC_REQUIRE((k_lo == 0) == (v_lo == 0), "attn_q1s: keys and values must both be split or both be plain bf16");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "attn_q1s: dropout needs a seed pointer");
    Q1sArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = (const __bf16*)K; a.ldk = ldk; a.V = (const __bf16*)V; a.ldv = ldv; a.k_lo = k_lo; a.v_lo = v_lo;
    a.O = O; a.ldo = ldo; a.LSE = LSE; a.seq = seq; a.n_seq = n_seq; a.H = H; a.max_q = 1; a.max_k = max_k; a.key_mask = key_mask;
    a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    const dim3 grid(ceil_div(n_seq * H, 4)), block(256);
    const bool split = k_lo != 0;
    if (dh == 64 && q1r_on() && ldq % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)O) & 15) == 0) {
        if (split) hipLaunchKernelGGL((attn_q1r_fwd_kernel<true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1r_fwd_kernel<false>), grid, block, 0, stream, a);
    } else if (dh == 64) { if (split) hipLaunchKernelGGL((attn_q1s_fwd_kernel<64, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1s_fwd_kernel<64, false>), grid, block, 0, stream, a); }
    else { if (split) hipLaunchKernelGGL((attn_q1s_fwd_kernel<32, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1s_fwd_kernel<32, false>), grid, block, 0, stream, a); }
    return svpc_check_launch("attn_q1s_fwd");
}

// backward: dQ fp32 (one row per sequence), dK / dV dense bf16 rows (every key row of every sequence is written)
int svpc_attn_q1s_bwd(const float* Q, int ldq, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, const float* dO, int lddo,
                      float* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int* seq, int n_seq, int H, int dh, int max_k,
                      const float* key_mask, float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(q1s_ok(dh, max_k, ldq, ldk, ldv, k_lo, v_lo, Q, K, V) && lddo % 4 == 0 && lddq % 2 == 0 && lddk % 8 == 0 && lddv % 8 == 0 &&
                     ((((uintptr_t)dO) | ((uintptr_t)dK) | ((uintptr_t)dV)) & 15) == 0,
                 "attn_q1s_bwd: unsupported shape / alignment");
    SVPC_REQUIRE((k_lo == 0) == (v_lo == 0), "attn_q1s: keys and values must both be split or both be plain bf16");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "attn_q1s: dropout needs a seed pointer");
    Q1sArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = (const __bf16*)K; a.ldk = ldk; a.V = (const __bf16*)V; a.ldv = ldv; a.k_lo = k_lo; a.v_lo = v_lo;
    a.seq = seq; a.n_seq = n_seq; a.H = H; a.max_q = 1; a.max_k = max_k; a.key_mask = key_mask; a.scale = scale; a.p_drop = p_drop;
    a.site = site; a.seed = seed; a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = (__bf16*)dK; a.lddk = lddk;
    a.dV = (__bf16*)dV; a.lddv = lddv;
    const dim3 grid(ceil_div(n_seq * H, 4)), block(256);
    const bool split = k_lo != 0;
    if (dh == 64 && q1r_on() && lddq % 4 == 0 && (((uintptr_t)dQ) & 15) == 0) {
        if (split) hipLaunchKernelGGL((attn_q1r_bwd_kernel<true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1r_bwd_kernel<false>), grid, block, 0, stream, a);
    } else if (dh == 64) { if (split) hipLaunchKernelGGL((attn_q1s_bwd_kernel<64, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1s_bwd_kernel<64, false>), grid, block, 0, stream, a); }
    else { if (split) hipLaunchKernelGGL((attn_q1s_bwd_kernel<32, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((attn_q1s_bwd_kernel<32, false>), grid, block, 0, stream, a); }
    return svpc_check_launch("attn_q1s_bwd");
}

}  // extern "C"
