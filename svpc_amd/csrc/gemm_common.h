// Shared GEMM epilogue (bias, activation, dropout, pre-activation store, accumulate) and split-K reducer.
#pragma once
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Epi {
    const float* bias; int act; float p_drop; uint32_t site; const u64* seed; int accumulate;
    float* Z;   // optional pre-activation output (same ldc)
};

__device__ __forceinline__ void epilogue_store(float v, int row, int col, float* __restrict__ C, int ldc, const Epi& e,
                                               u64 seed, float inv_keep) {
    if (e.bias) v += e.bias[col];
    const size_t o = (size_t)row * ldc + col;
    if (e.Z) e.Z[o] = v;
    v = apply_act(v, e.act);
    if (e.p_drop > 0.f) v *= drop_scale(seed, e.site, o, e.p_drop, inv_keep);
    if (e.accumulate) v += C[o];
    C[o] = v;
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
    C[o] = (TC)v;
}

// XCD-aware bijective remap of a linear workgroup id: ids that share an XCD (id % 8) get a contiguous tile range,
// so tiles sharing an operand panel hit the same private L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
