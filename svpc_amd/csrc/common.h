// Shared device helpers for the svpc gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SVPC_WAVE 64

typedef unsigned long long u64;

// ---- status plumbing for the C-ABI -------------------------------------------------------------
extern "C" void svpc_set_error(const char* msg);
extern "C" int svpc_raise_lds_once(const void* fn, const char* who);   // api.cpp
static inline int svpc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        svpc_set_error(buf);
        return (int)e;
    }
    return 0;
}
#define SVPC_REQUIRE(cond, msg)                 \
    do {                                        \
        if (!(cond)) {                          \
            svpc_set_error(msg);                \
            return -1;                          \
        }                                       \
    } while (0)

// ---- wave / block reductions -------------------------------------------------------------------
// Sum over the 64 lanes, result in every lane.  Four DPP steps inside each row of 16 lanes (quad swaps, half-mirror, mirror — VALU
// cross-lane moves, no LDS) and three scalar adds of the four row sums (v_readlane): ≈10× shorter dependency chain than six
// __shfl_xor (= ds_bpermute through the LDS crossbar).  All 64 lanes must be active (every caller is in wave-uniform control flow).
template <int CTRL>
__device__ __forceinline__ float dpp_add_(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(r);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add_<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add_<0x141>(v);     // row_half_mirror
    v = dpp_add_<0x140>(v);     // row_mirror: every lane now holds the sum of its 16-lane row
    const int iv = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}
// Reduce-scatter butterfly: 32 per-lane addends in, ONE value out per lane pair — lane l ends with the 64-lane sum of value l >> 1.
// v_permlane32_swap / v_permlane16_swap exchange register halves across lane^32 / lane^16, DPP row_ror:8, row_half_mirror and the quad
// permutes finish inside a row: ≈70 instructions for 32 sums (32 separate wave_sum()s: ≈350).  Fixed order: deterministic.
template <int CTRL> __device__ __forceinline__ float dpp_mov_(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_reduce_scatter32(const float (&v)[32], int lane) {
    float w[16], x[8], y[4], z[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        w[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // lanes < 32: value i over both halves; lanes >= 32: value i + 16
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[i]), __float_as_uint(w[i + 8]), false, false);
        x[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // even rows of 16 lanes: value i (+16); odd rows: value i + 8 (+16)
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = (b3 ? x[i + 4] : x[i]) + dpp_mov_<0x128>(b3 ? x[i] : x[i + 4]);          // lane ^ 8 (row_ror:8)
#pragma unroll
    for (int i = 0; i < 2; ++i) z[i] = (b2 ? y[i + 2] : y[i]) + dpp_mov_<0x141>(b2 ? y[i] : y[i + 2]);          // lane ^ 7 (row_half_mirror)
    float u = (b1 ? z[1] : z[0]) + dpp_mov_<0x4E>(b1 ? z[0] : z[1]);                                             // lane ^ 2
    u += dpp_mov_<0xB1>(u);                                                                                       // lane ^ 1
    return u;
}
template <int CTRL>
__device__ __forceinline__ float dpp_max_(float v) {
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false);
    return fmaxf(v, __int_as_float(r));
}
__device__ __forceinline__ float wave_max(float v) {
    v = dpp_max_<0xB1>(v);
    v = dpp_max_<0x4E>(v);
    v = dpp_max_<0x141>(v);
    v = dpp_max_<0x140>(v);
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}
// block-wide sum for blockDim.x == 256 (4 waves); `red` is 4 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- counter-based RNG (dropout masks, Gumbel noise) ---------------------------------------------
// One uniform per (seed, site, element index); recomputed in backward, never stored.  The generator is a 32-bit
// integer hash (lowbias32: 2 multiplies + 3 xor-shifts ≈ 8 VALU ops) — the dropout masks of the LayerNorm / attention kernels
// are drawn per element inside HBM-bound kernels, so a 64-bit mixer would make them VALU-bound.
__device__ __host__ __forceinline__ uint32_t svpc_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __host__ __forceinline__ uint32_t svpc_hash32(u64 seed, uint32_t site, u64 idx) {
    const uint32_t key = (uint32_t)(seed ^ (seed >> 32)) + (site + 1u) * 0x9E3779B9u;
    const uint32_t hi = (uint32_t)(idx >> 32);
    return svpc_mix32(((uint32_t)idx ^ key) + hi * 0x85EBCA6Bu);
}
// ---- dropout draws of the ATTENTION PROBABILITIES (model.py:213): element (row, k) of the (sequence·head·query, key) tensor.
// The full mixer runs once per ROW, A = mix32(row ^ key); an element costs one add, one xor-shift and ONE full-rate 24-bit multiply:
//     x = A + k·φ,  y = x ^ (x >> 16),  draw = (y[23:0] · M) >> 16   (16 bits, compared with p·65536).
// The attention kernels are bound by vector-instruction issue, and the two quarter-rate 32-bit multiplies + three xor-shifts of a
// per-element lowbias32 were a fifth (bf16x3) to a third (bf16) of the clip-encoder forward's instructions.  Forward kernels keep a
// query per lane (A once per lane and pair), backward kernels a key per lane (k·φ once per lane, A per register from a per-pair LDS
// table): the same draw either way.  Statistics (keep rate, row / column variance against the binomial, lag and 2×2 correlations)
// checked against the lowbias32 draw in tools/dbg/attn_draw_stats.py.
__device__ __host__ __forceinline__ uint32_t svpc_attn_row_hash(uint32_t key, u64 row) {
    return svpc_mix32(((uint32_t)row ^ key) + (uint32_t)(row >> 32) * 0x85EBCA6Bu);
}
constexpr uint32_t SVPC_ATTN_PHI = 0x9E3779B1u, SVPC_ATTN_M24 = 0xB5297Bu;
__device__ __host__ __forceinline__ uint32_t svpc_attn_draw16(uint32_t a_plus_kphi) {
    uint32_t x = a_plus_kphi;
#if defined(__HIP_DEVICE_COMPILE__)
    // (the xor-shift as ONE sdwa instruction the compiler cannot look into: knowing that the multiply reads 24 bits it otherwise masks x
    // first — (x & 0xffffff) ^ (x >> 16) — one more instruction per draw)
    uint32_t y;
    asm("v_xor_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(y) : "v"(x));
    return __umul24(y, SVPC_ATTN_M24) >> 16;
#else
    x ^= x >> 16;
    return (uint32_t)(((u64)(x & 0xFFFFFFu) * SVPC_ATTN_M24) & 0xFFFFFFFFull) >> 16;
#endif
}
__device__ __host__ __forceinline__ uint32_t svpc_drop_key(u64 seed, uint32_t site) { return (uint32_t)(seed ^ (seed >> 32)) + (site + 1u) * 0x9E3779B9u; }
// multiplier (0 or 1/(1-p)) of attention-probability element (row, k)
__device__ __forceinline__ float attn_drop_scale(u64 seed, uint32_t site, u64 row, uint32_t k, float p, float inv_keep) {
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    return svpc_attn_draw16(svpc_attn_row_hash(svpc_drop_key(seed, site), row) + k * SVPC_ATTN_PHI) >= thr ? inv_keep : 0.0f;
}
// keep-probability 1-p (16-bit resolution); returns the multiplier (0 or 1/(1-p))
__device__ __forceinline__ float drop_scale(u64 seed, uint32_t site, u64 idx, float p, float inv_keep) {
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    return (svpc_hash32(seed, site, idx) >> 16) >= thr ? inv_keep : 0.0f;
}
__device__ __forceinline__ float gumbel_noise(u64 seed, uint32_t site, u64 idx) {
    float u = ((float)svpc_hash32(seed, site, idx) + 0.5f) * (1.0f / 4294967296.0f);
    u = fminf(fmaxf(u, 1e-10f), 1.0f - 6e-8f);
    return -logf(-logf(u));
}

// ---- activations -------------------------------------------------------------------------------
#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_GELU 2
#define ACT_SIGMOID 3

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    // d/dx [x Φ(x)] = Φ(x) + x φ(x)
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// The same derivative with the Abramowitz–Stegun 7.1.26 erf (|error| ≤ 1.5e-7, far below the bf16 rounding of what is stored), whose
// exponential exp(-x²/2) is also the density term: ≈16 instructions instead of erff + expf (≈90).  For GEMM epilogues on bf16
// streams, where a tile's tail evaluates it tens of thousands of times with nothing left to overlap.
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float u = x * 0.70710678118654752f, au = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-au * au);
    const float cdf = 0.5f * (1.0f + copysignf(fmaf(-p * t, e, 1.0f), u));
    return fmaf(x * 0.3989422804014327f, e, cdf);
}
// act'(.) from what the forward kept: the pre-activation z for GELU, the activated output y for ReLU / sigmoid (as svpc_act_bwd)
__device__ __forceinline__ float act_grad_from_aux(float aux, int act, bool fast_gelu) {
    switch (act) {
        case ACT_RELU: return aux > 0.f ? 1.0f : 0.0f;
        case ACT_GELU: return fast_gelu ? gelu_grad_fast(aux) : gelu_erf_grad(aux);
        case ACT_SIGMOID: return aux * (1.0f - aux);
        default: return 1.0f;
    }
}
__device__ __forceinline__ float apply_act(float z, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(z, 0.0f);
        case ACT_GELU: return gelu_erf(z);
        case ACT_SIGMOID: return sigmoidf_(z);
        default: return z;
    }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
