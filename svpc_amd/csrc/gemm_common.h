// Shared GEMM epilogue (bias, activation, dropout, pre-activation store, accumulate) and split-K reducer.
#pragma once
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Epi {
    const float* bias; int act; float p_drop; uint32_t site; const u64* seed; int accumulate;
    float* Z;   // optional pre-activation output (same ldc)
    const void* R;   // optional addend of C's type and layout, C = act(acc + bias) + R (a residual-path gradient joining a dgrad).  bf16: only
                     // the 16-byte row-store epilogues of gemm_glds.hip apply it (the launcher refuses R where they cannot run);
                     // fp32: the element-wise epilogues below (svpc_gemm_l32_r)
    const void* G; int gact;   // optional: C = (A·B) ⊙ gact'(G) (+ R) — G of C's type and layout is what the forward of activation `gact` kept
                               // (z for GELU, y for ReLU / sigmoid): a dgrad whose output feeds an activation's backward applies it.
                               // bf16: like a bf16 R only in the 16-byte row-store epilogues of gemm_glds.hip (the launcher refuses it
                               // elsewhere); fp32: the element-wise epilogue below (svpc_gemm_l32_rg)
};

__device__ __forceinline__ void epilogue_store(float v, int row, int col, float* __restrict__ C, int ldc, const Epi& e,
                                               u64 seed, float inv_keep) {
    if (e.bias) v += e.bias[col];
    const size_t o = (size_t)row * ldc + col;
    if (e.Z) e.Z[o] = v;
    v = apply_act(v, e.act);
    if (e.p_drop > 0.f) v *= drop_scale(seed, e.site, o, e.p_drop, inv_keep);
    if (e.G) v *= act_grad_from_aux(reinterpret_cast<const float*>(e.G)[o], e.gact, false);      // (fp32 G: svpc_gemm_l32_rg)
    if (e.accumulate) v += C[o];
    if (e.R) v += reinterpret_cast<const float*>(e.R)[o];
    C[o] = v;
}


// The same epilogue for NE elements of one thread at once: every load it needs (bias, the old C of an accumulate, R, G) is issued before
// the first is consumed.  Called element by element (epilogue_store in a loop) each conditional load sits in a region of its own and the
// compiler drains the memory counter behind every one of them — NE × (features) dependent round trips per thread.  `row` / `col` must be
// valid (clamped) coordinates for every element; `ok[k]` says whether element k is stored.  Same arithmetic, same order.
template <int NE>
__device__ __forceinline__ void epilogue_store_n(const float (&vin)[NE], const int (&row)[NE], const int (&col)[NE], const bool (&ok)[NE],
                                                 float* __restrict__ C, int ldc, const Epi& e, u64 seed, float inv_keep) {
    size_t o[NE];
    float b[NE], old[NE], r[NE], g[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) { o[k] = (size_t)row[k] * ldc + col[k]; b[k] = 0.f; old[k] = 0.f; r[k] = 0.f; g[k] = 0.f; }
    if (e.bias) {
#pragma unroll
        for (int k = 0; k < NE; ++k) b[k] = e.bias[col[k]];
    }
    if (e.accumulate) {
#pragma unroll
        for (int k = 0; k < NE; ++k) old[k] = C[o[k]];
    }
    if (e.R) {
#pragma unroll
        for (int k = 0; k < NE; ++k) r[k] = reinterpret_cast<const float*>(e.R)[o[k]];
    }
    if (e.G) {
#pragma unroll
        for (int k = 0; k < NE; ++k) g[k] = reinterpret_cast<const float*>(e.G)[o[k]];
    }
    float v[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) v[k] = vin[k] + b[k];
    if (e.Z) {
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (ok[k]) e.Z[o[k]] = v[k];
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        float t = apply_act(v[k], e.act);
        if (e.p_drop > 0.f) t *= drop_scale(seed, e.site, o[k], e.p_drop, inv_keep);
        if (e.G) t *= act_grad_from_aux(g[k], e.gact, false);
        v[k] = (t + old[k]) + r[k];
    }
    bool all_ok = true;
#pragma unroll
    for (int k = 0; k < NE; ++k) all_ok = all_ok && ok[k];
    if (__all(all_ok)) {                                  // interior (wave-uniform): stores back to back, no exec-masked regions
#pragma unroll
        for (int k = 0; k < NE; ++k) C[o[k]] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (ok[k]) C[o[k]] = v[k];
    }
}

// typed variant: C and the optional pre-activation copy Z are stored as TC (float or __bf16)
template <typename TC>
__device__ __forceinline__ void epilogue_store_t(float v, int row, int col, TC* __restrict__ C, int ldc, const Epi& e, u64 seed,
                                                 float inv_keep) {
    if (e.bias) v += e.bias[col];
    const size_t o = (size_t)row * ldc + col;
    if (e.Z) reinterpret_cast<TC*>(e.Z)[o] = (TC)v;
    v = apply_act(v, e.act);
    if (e.p_drop > 0.f) v *= drop_scale(seed, e.site, o, e.p_drop, inv_keep);
    if (e.accumulate) v += (float)C[o];
    if (sizeof(TC) == 4 && e.R) v += reinterpret_cast<const float*>(e.R)[o];      // (bf16 addends: gemm_glds.hip's row-store epilogues)
    C[o] = (TC)v;
}

// XCD-aware bijective remap of a linear workgroup id: ids that share an XCD (id % 8) get a contiguous tile range,
// so tiles sharing an operand panel hit the same private L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}


// ---- split-K tail shared by the MFMA GEMM families: C = epi( Σ_k slabs[k] ), slabs summed in k order (deterministic).
// The vector form handles 4 consecutive columns per thread with 4 slab loads in flight (N % 4 == 0); the sum order per
// element is the same as in the scalar form, so both give identical bits.
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce1_kernel(const float* __restrict__ slabs, int splitk, TC* __restrict__ C, int ldc,
                                                             int M, int N, Epi epi) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i - (size_t)row * N);
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += slabs[(size_t)k * M * N + i];
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    epilogue_store_t<TC>(s, row, col, C, ldc, epi, seed, inv_keep);
}
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ slabs, int splitk, TC* __restrict__ C, int ldc,
                                                             int M, int N, Epi epi) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x, nq = (size_t)M * N / 4;
    if (q >= nq) return;
    const size_t i = q * 4;
    const int row = (int)(i / N), col = (int)(i - (size_t)row * N);
    const float4* p = reinterpret_cast<const float4*>(slabs) + q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= splitk; k += 4) {
        const float4 a = p[(size_t)k * nq], b = p[(size_t)(k + 1) * nq], c = p[(size_t)(k + 2) * nq], d = p[(size_t)(k + 3) * nq];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < splitk; ++k) {
        const float4 a = p[(size_t)k * nq];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    epilogue_store_t<TC>(s.x, row, col, C, ldc, epi, seed, inv_keep);
    epilogue_store_t<TC>(s.y, row, col + 1, C, ldc, epi, seed, inv_keep);
    epilogue_store_t<TC>(s.z, row, col + 2, C, ldc, epi, seed, inv_keep);
    epilogue_store_t<TC>(s.w, row, col + 3, C, ldc, epi, seed, inv_keep);
}
template <typename TC>
static inline void launch_splitk_reduce(const float* slabs, int splitk, TC* C, int ldc, int M, int N, const Epi& epi, hipStream_t stream) {
    const size_t n = (size_t)M * N;
    if (N % 4 == 0 && ((((uintptr_t)slabs) & 15) == 0))
        hipLaunchKernelGGL(splitk_reduce4_kernel<TC>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, splitk, C, ldc, M, N,
                           epi);
    else
        hipLaunchKernelGGL(splitk_reduce1_kernel<TC>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, splitk, C, ldc, M, N, epi);
}
