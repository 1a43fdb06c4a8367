// Types and device helpers shared by the MFMA attention kernels (attention_mfma.hip, attention_pipe.hip): launch arguments, the
// counter-based dropout draw of the probability tensor (recomputed by the backward: every forward kernel must draw identically) and
// the 32x32 accumulator's register -> row map.  Reference semantics: src/rtransformer/model.py:194-219.
#pragma once
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

struct MAttnArgs {
    const void* Q; int ldq; const void* K; int ldk; const void* V; int ldv;
    void* O; int ldo; float* LSE;
    const int* seq; int n_seq, H, max_q, max_k;
    const float* key_mask; int causal; float scale; float p_drop; uint32_t site; const u64* seed;
    const void* dO; int lddo; void* dQ; int lddq; void* dK; int lddk; void* dV; int lddv;
};

constexpr int AT_MAX = 128;
constexpr float LOG2E = 1.4426950408889634f;      // exp(x - m) = exp2(x·log2e - m·log2e): one v_fma + one v_exp per score   // rows per image: 128 (clip encoder) or 32 (22-token decoder, ≤12-step sequences, ≤3-slot memory)

// Dropout of the (sequence, head, query, key) probability tensor: the library's row-hash draw (common.h, svpc_attn_draw16) — the full
// mixer once per probability ROW (index (s·H + h)·max_q + q), one add + xor-shift + 24-bit multiply per element.  Forward kernels hold
// a query per lane (row hash once per lane); the backward holds a key per lane and reads the row hashes of its query registers from a
// per-pair LDS table.
struct DropCtx {
    uint32_t key, thr; float ik, p;
    __device__ __forceinline__ DropCtx(const u64* seed_ptr, uint32_t site_, float p_) {
        p = p_;
        const u64 seed = p_ > 0.f ? seed_ptr[0] : 0ull;
        ik = p_ > 0.f ? 1.0f / (1.0f - p_) : 1.0f;
        key = svpc_drop_key(seed, site_);
        thr = (uint32_t)(p_ * 65536.0f);
    }
    __device__ __forceinline__ uint32_t row(u64 row_index) const { return svpc_attn_row_hash(key, row_index); }
    // the 16 probabilities of one 32-key accumulator tile of the row with hash `arow` (element e ↔ key key0 + acc_row(e, lane)) times
    // their dropout multipliers
    __device__ __forceinline__ void mul16(uint32_t arow, int key0, int lane, float* pv) const {
        const uint32_t b = arow + (uint32_t)(key0 + 4 * (lane >> 5)) * SVPC_ATTN_PHI;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            pv[e] = svpc_attn_draw16(b + (uint32_t)((e & 3) + 8 * (e >> 2)) * SVPC_ATTN_PHI) >= thr ? pv[e] * ik : 0.0f;
    }
    // multipliers of key k (kphi = k·φ) in four rows whose hashes are a4[0..3]: the backward's four consecutive queries of one key
    __device__ __forceinline__ void mul4(const uint32_t* a4, uint32_t kphi, float* dm) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) dm[j] = svpc_attn_draw16(a4[j] + kphi) >= thr ? ik : 0.0f;
    }
};

// bf16x3 (split) operands: the hi planes at the MAttnArgs pointers, the lo planes these many elements behind them
struct X3AttnArgs {
    MAttnArgs m;
    int q_lo, k_lo, v_lo, o_lo;      // element offsets of the lo planes from the hi-plane pointers
    int dbg;                         // attention_pipe.hip timing experiments (SVPC_PP_DBG; 0 in production): 1 no DMA, 2 no compute, 4 no O stores, 8 no Q loads,
                                     // 16 cycle stamps of workgroups 0..15 into dbg_buf
    unsigned long long* dbg_buf;
};
// v_mfma_f32_32x32x16 accumulator: register e of lane l holds row acc_row(e, l), column l & 31
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
