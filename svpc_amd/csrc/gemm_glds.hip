// bf16 × bf16 GEMM for the bf16 activation streams with DIRECT-TO-LDS operand staging (global_load_lds_dwordx4): no staging
// registers, no conversion, no ds_write — the loads of the next k-tiles stay in flight behind a counted vmcnt while the matrix
// cores work on the current one (LDS ring, raw s_barrier).
//
//   C[M,N] = epi( sum_k A(m,k) · B(n,k) ),  A: a_kc ? [M][lda] : [K][lda],  B: b_kc ? [N][ldb] : [K][ldb]   (bf16)
// Any M, N: rows past the edge are clamped to the last valid row / 16-byte chunk (their products are never stored).  K: whole
// 32-deep tiles for a k-contiguous operand (K is a feature width there); for a k-strided operand (wgrad: K = the row count of the
// activation stream, arbitrary) the k-rows past K are fetched from a block of zeros instead.
//
// Three tile forms, 512 threads = 8 waves each, MFMA v_mfma_f32_32x32x16_bf16:
//   128×128×32  waves 2×4, wave tile 64×32, 3-stage ring (48 KiB, three workgroups per CU)      — the decoder's 4,224-row launches
//   256×128×32  waves 4×2, wave tile 64×64, 3-stage ring (72 KiB, two per CU)                    — mid-sized grids
//   256×256×32  two wave groups in ping-pong, wave tile 128×64, 4-stage ring (128 KiB, one per CU) — ≥ 150 such tiles: every
//               19,200-row forward / dgrad launch and the grouped weight gradients (glds_tile_pp below)
// One k-tile of one 128-row operand image is 8 KiB = 8 wave-instructions of 1 KiB (64 lanes × 16 B, LDS destination = wave-uniform
// base + lane·16); wave w issues instruction w of every image.  The LDS image is linear in that order; bank conflicts are removed by
// permuting the 16-byte chunks on the SOURCE side (per-lane global address) and applying the same XOR on the read side:
//   k-contiguous operand: [128 rows][64 B], chunk c of row r stored at c ^ ((r>>2)&3)      → ds_read_b128 fragments, conflict-free
//   k-strided operand:    [32 k-rows][256 B], chunk c of k-row q stored at c ^ ((q&3)<<2)  → ds_read_b64_tr_b16 fragments
// bf16 outputs are accumulated transposed (a row on a lane) and leave as 16-byte row pieces / whole 128-byte lines (glds_store_*).
// Weights come from the bf16 shadow arena the fused optimizer maintains next to the fp32 masters.
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int GL_BM = 128, GL_BN = 128, GL_BK = 32;
constexpr int GL_OP = 8192;                 // bytes of one 128-row operand tile

typedef const void __attribute__((address_space(1))) * gl_gptr;
typedef void __attribute__((address_space(3))) * gl_lptr;

__device__ __attribute__((aligned(16))) const float glds_zeros[4] = {0.f, 0.f, 0.f, 0.f};     // source of k-rows past K

// per-lane source of this wave's 1-KiB piece of an operand tile; `krow` = the lane's k-row inside the tile (k-strided operands)
// (k-contiguous: piece `wave` = tile rows 16·wave…; a 256-row tile is two 128-row halves, `half` selects one.  k-strided: piece
//  `wave` = k-rows 4·wave…, 256 B = 128 columns per k-row; `half` selects columns 128·half…)
template <bool KC>
__device__ __forceinline__ const __bf16* glds_src(const __bf16* P, int ld, int m0, int rows, int k0, int wave, int lane, size_t& step,
                                                  int& krow, int half = 0) {
    m0 += 128 * half;
    if (KC) {
        const int row = 16 * wave + (lane >> 2), pos = lane & 3;
        const int c = pos ^ ((row >> 2) & 3);
        step = GL_BK;
        krow = 0;
        return P + (size_t)min(m0 + row, rows - 1) * ld + k0 + 8 * c;
    } else {
        krow = 4 * wave + (lane >> 4);
        const int pos = lane & 15;
        const int c = pos ^ ((krow & 3) << 2);
        step = (size_t)GL_BK * ld;
        return P + (size_t)(k0 + krow) * ld + min(m0 + 8 * c, rows - 8);      // rows % 8 == 0 (host check)
    }
}

// MFMA 32x32x16 operand fragment (rows row0..row0+31 of the tile, k-step ks) from a staged image
template <bool KC>
__device__ __forceinline__ bf16x8 glds_fragment(const char* __restrict__ img, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 31);
        const int c = (2 * ks + (lane >> 5)) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const bf16x8*>(img + row * 64 + c * 16);
    } else {
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g >> 1;
        const int krow = ks * 16 + 8 * h + q;                       // (krow & 3) == q for both halves
        const int col = row0 + 16 * (g & 1) + 4 * p;
        const int off = (((col >> 3) ^ (q << 2)) << 4) + (col & 7) * 2;
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + krow * 256 + off));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (krow + 4) * 256 + off));
        union { short s[8]; bf16x8 v; } u;
        u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
        u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
        return u.v;
    }
}

// ---- bf16 epilogue with an output ROW on a lane.  With the operands swapped (mfma(Bfrag, Afrag)) a 32×32 accumulator block holds
// row `lane & 31` on the lane and columns (e&3) + 8·(e>>2) + 4·(lane>>5) in its 16 registers: four runs of 4 consecutive
// columns.  Two v_permlane32_swap per pair of runs give every lane 8 consecutive columns → ONE 16-byte store per lane and pair
// (2 store instructions per block instead of 8 dword stores through LDS-crossbar shuffles), the activation is a template constant.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    union { __bf16 h[2]; uint32_t u; } pk;
    pk.h[0] = (__bf16)lo; pk.h[1] = (__bf16)hi;
    return pk.u;
}
__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b) {       // (a.lo + b.lo, a.hi + b.hi) in fp32, rounded once
    return pack_bf16x2(__uint_as_float(a << 16) + __uint_as_float(b << 16), __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u));
}
__device__ __forceinline__ uint4 add_bf16x8(uint4 a, uint4 b) {
    return make_uint4(add_bf16x2(a.x, b.x), add_bf16x2(a.y, b.y), add_bf16x2(a.z, b.z), add_bf16x2(a.w, b.w));
}
__device__ __forceinline__ uint32_t mul_dact_bf16x2(uint32_t v, uint32_t g, int gact) {   // v ⊙ gact'(g), both halves, rounded once
    return pack_bf16x2(__uint_as_float(v << 16) * act_grad_from_aux(__uint_as_float(g << 16), gact, true),
                       __uint_as_float(v & 0xffff0000u) * act_grad_from_aux(__uint_as_float(g & 0xffff0000u), gact, true));
}
__device__ __forceinline__ uint4 mul_dact_bf16x8(uint4 v, uint4 g, int gact) {
    return make_uint4(mul_dact_bf16x2(v.x, g.x, gact), mul_dact_bf16x2(v.y, g.y, gact), mul_dact_bf16x2(v.z, g.z, gact),
                      mul_dact_bf16x2(v.w, g.w, gact));
}
template <int TMF, int TNF, int ACT>
__device__ __forceinline__ void glds_store_rows(const floatx16 (&acc)[TMF][TNF], __bf16* __restrict__ C, int ldc, __bf16* __restrict__ Z,
                                                const float* __restrict__ bias, int row0, int col0, int M, int N, int lane,
                                                const __bf16* __restrict__ Radd = nullptr, const __bf16* __restrict__ Gd = nullptr,
                                                int gact = 0) {
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < TNF; ++j) {
        const int cb = col0 + j * 32;
        float bb[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = cb + 8 * g + 4 * lhi;
            if (bias && c + 4 <= N) t = *reinterpret_cast<const float4*>(bias + c);
            bb[4 * g] = t.x; bb[4 * g + 1] = t.y; bb[4 * g + 2] = t.z; bb[4 * g + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int row = row0 + i * 32 + l31;
#pragma unroll
            for (int k = 0; k < 4; k += 2) {            // runs k and k+1 → 8-column chunks k (lanes 0-31) and k+1 (lanes 32-63)
                uint32_t y[4], z[4];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int e = 4 * (k + g);
                    const float z0 = acc[i][j][e] + bb[e], z1 = acc[i][j][e + 1] + bb[e + 1];
                    const float z2 = acc[i][j][e + 2] + bb[e + 2], z3 = acc[i][j][e + 3] + bb[e + 3];
                    y[2 * g] = pack_bf16x2(apply_act(z0, ACT), apply_act(z1, ACT));
                    y[2 * g + 1] = pack_bf16x2(apply_act(z2, ACT), apply_act(z3, ACT));
                    if (Z) { z[2 * g] = pack_bf16x2(z0, z1); z[2 * g + 1] = pack_bf16x2(z2, z3); }
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
                const int cc = cb + 8 * (k + lhi);
                const bool ok = row < M && cc + 8 <= N;
                const size_t o = (size_t)row * ldc + cc;
                if (ok) {
                    uint4 v = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    if (Gd) v = mul_dact_bf16x8(v, *reinterpret_cast<const uint4*>(Gd + o), gact);
                    if (Radd) v = add_bf16x8(v, *reinterpret_cast<const uint4*>(Radd + o));
                    *reinterpret_cast<uint4*>(C + o) = v;
                }
                if (Z) {
                    auto q0 = __builtin_amdgcn_permlane32_swap(z[0], z[2], false, false);
                    auto q1 = __builtin_amdgcn_permlane32_swap(z[1], z[3], false, false);
                    if (ok) *reinterpret_cast<uint4*>(Z + o) = make_uint4(q0[0], q1[0], q0[1], q1[1]);
                }
                __builtin_amdgcn_sched_barrier(0);      // one chunk at a time: keeps the register footprint of the tail small
            }
        }
    }
}
// the same packing, but through a wave-private LDS image [32·TMF rows][128 B] so that the global stores cover whole 128-byte
// lines (8 lanes × 16 B per row, 8 rows per instruction) instead of 32-byte pieces of 32 different lines: the piecewise form
// measured 1.25–1.35× the algorithmic write bytes at HBM (partial-line write-backs).  TNF = 2 (a 64-column = 128-byte wave tile).
// 16-byte chunk c of row r is kept at c ^ (r & 7): conflict-free for the row-per-lane writes and for the line-per-8-lanes reads.
template <int TMF, int ACT, bool PRE>
__device__ __forceinline__ void glds_store_rows_lds_pass(const floatx16 (&acc)[TMF][2], __bf16* __restrict__ C, int ldc,
                                                         const float* __restrict__ bias, int row0, int col0, int M, int N, int lane,
                                                         char* __restrict__ wl, const __bf16* __restrict__ Radd = nullptr,
                                                         const __bf16* __restrict__ Gd = nullptr, int gact = 0) {
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cb = col0 + j * 32;
        float bb[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = cb + 8 * g + 4 * lhi;
            if (bias && c + 4 <= N) t = *reinterpret_cast<const float4*>(bias + c);
            bb[4 * g] = t.x; bb[4 * g + 1] = t.y; bb[4 * g + 2] = t.z; bb[4 * g + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int r = i * 32 + l31;
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                uint32_t y[4];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int e = 4 * (k + g);
                    const float z0 = acc[i][j][e] + bb[e], z1 = acc[i][j][e + 1] + bb[e + 1];
                    const float z2 = acc[i][j][e + 2] + bb[e + 2], z3 = acc[i][j][e + 3] + bb[e + 3];
                    y[2 * g] = PRE ? pack_bf16x2(z0, z1) : pack_bf16x2(apply_act(z0, ACT), apply_act(z1, ACT));
                    y[2 * g + 1] = PRE ? pack_bf16x2(z2, z3) : pack_bf16x2(apply_act(z2, ACT), apply_act(z3, ACT));
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
                const int chunk = j * 4 + k + lhi;
                *reinterpret_cast<uint4*>(wl + r * 128 + ((chunk ^ (r & 7)) << 4)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // same wave wrote and reads: LDS operations of a wave complete in order
    const int chunk = lane & 7, cc = col0 + 8 * chunk;
#pragma unroll
    for (int it = 0; it < 4 * TMF; ++it) {
        const int r = it * 8 + (lane >> 3), row = row0 + r;
        uint4 v = *reinterpret_cast<const uint4*>(wl + r * 128 + ((chunk ^ (r & 7)) << 4));
        if (row < M && cc + 8 <= N) {
            if (Gd) v = mul_dact_bf16x8(v, *reinterpret_cast<const uint4*>(Gd + (size_t)row * ldc + cc), gact);
            if (Radd) v = add_bf16x8(v, *reinterpret_cast<const uint4*>(Radd + (size_t)row * ldc + cc));
            *reinterpret_cast<uint4*>(C + (size_t)row * ldc + cc) = v;
        }
    }
}
template <int TMF, int ACT>
__device__ __forceinline__ void glds_store_rows_lds(const floatx16 (&acc)[TMF][2], __bf16* __restrict__ C, int ldc, __bf16* __restrict__ Z,
                                                    const float* __restrict__ bias, int row0, int col0, int M, int N, int lane,
                                                    char* __restrict__ wl, const __bf16* __restrict__ Radd,
                                                    const __bf16* __restrict__ Gd = nullptr, int gact = 0) {
    glds_store_rows_lds_pass<TMF, ACT, false>(acc, C, ldc, bias, row0, col0, M, N, lane, wl, Radd, Gd, gact);
    if (Z) glds_store_rows_lds_pass<TMF, ACT, true>(acc, Z, ldc, bias, row0, col0, M, N, lane, wl);
}
// generic element-wise form of the same orientation (dropout, accumulate, split-K slabs, unaligned or ragged-by-less-than-8 outputs)
template <int TMF, int TNF, typename TC>
__device__ __forceinline__ void glds_store_rows_generic(const floatx16 (&acc)[TMF][TNF], TC* __restrict__ C, int ldc, const Epi& epi,
                                                        int row0, int col0, int M, int N, int lane, int splitk, int ks_id,
                                                        float* __restrict__ slabs) {
    const u64 seed = (epi.p_drop > 0.f && splitk == 1) ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < TNF; ++j)
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int row = row0 + i * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int col = col0 + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row >= M || col >= N) continue;
                if (splitk == 1) epilogue_store_t<TC>(acc[i][j][e], row, col, C, ldc, epi, seed, inv_keep);
                else slabs[((size_t)ks_id * M + row) * N + col] = acc[i][j][e];
            }
        }
}
template <int TMF, int TNF, typename TC>
__device__ __forceinline__ void glds_store_tr(const floatx16 (&acc)[TMF][TNF], TC* __restrict__ C, int ldc, const Epi& epi, int row0,
                                              int col0, int M, int N, int lane, int splitk, int ks_id, float* __restrict__ slabs,
                                              char* __restrict__ wl = nullptr) {
    const bool fast = splitk == 1 && epi.p_drop <= 0.f && !epi.accumulate && (N & 7) == 0 && (ldc & 7) == 0 &&
                      ((((uintptr_t)C) | ((uintptr_t)epi.Z) | ((uintptr_t)epi.bias) | ((uintptr_t)epi.R) | ((uintptr_t)epi.G)) & 15) == 0;
    if (!fast) {
        glds_store_rows_generic<TMF, TNF, TC>(acc, C, ldc, epi, row0, col0, M, N, lane, splitk, ks_id, slabs);
        return;
    }
    __bf16* Cb = reinterpret_cast<__bf16*>(C);
    __bf16* Zb = reinterpret_cast<__bf16*>(epi.Z);
    const __bf16* Rb = reinterpret_cast<const __bf16*>(epi.R);
    const __bf16* Gb = reinterpret_cast<const __bf16*>(epi.G);
    const int ga = epi.gact;
    if constexpr (TNF == 2) {
        if (wl) {       // whole-line stores through the wave's LDS image
            switch (epi.act) {
                case ACT_RELU: glds_store_rows_lds<TMF, ACT_RELU>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, wl, Rb, Gb, ga); break;
                case ACT_GELU: glds_store_rows_lds<TMF, ACT_GELU>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, wl, Rb, Gb, ga); break;
                case ACT_SIGMOID: glds_store_rows_lds<TMF, ACT_SIGMOID>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, wl, Rb, Gb, ga); break;
                default: glds_store_rows_lds<TMF, ACT_NONE>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, wl, Rb, Gb, ga); break;
            }
            return;
        }
    }
    switch (epi.act) {
        case ACT_RELU: glds_store_rows<TMF, TNF, ACT_RELU>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, Rb, Gb, ga); break;
        case ACT_GELU: glds_store_rows<TMF, TNF, ACT_GELU>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, Rb, Gb, ga); break;
        case ACT_SIGMOID: glds_store_rows<TMF, TNF, ACT_SIGMOID>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, Rb, Gb, ga); break;
        default: glds_store_rows<TMF, TNF, ACT_NONE>(acc, Cb, ldc, Zb, epi.bias, row0, col0, M, N, lane, Rb, Gb, ga); break;
    }
}


// BM = 128: waves 2 (m) × 4 (n), wave tile 64×32.  BM = 256: the A tile is two 128-row images, waves 4 (m) × 2 (n), wave tile
// 64×64 — 4 MFMAs per 4 fragment reads instead of 2 per 3, and half as many workgroups, which matters when 128-row tiles would
// leave a mostly empty second round of workgroups (900 tiles on 768 resident slots → 450 on 512).
// one (tile, k-slice) of one problem; `smem` = NS·(BM/128 + 1)·8 KiB of LDS
template <bool A_KC, bool B_KC, typename TC, int NS, int BM, int BN = 128>
__device__ __forceinline__ void glds_tile(char* __restrict__ smem, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                          TC* __restrict__ C, int ldc, int M, int N, int K, const Epi& epi, int tm, int tn, int ks_id,
                                          int splitk, int k_chunk, float* __restrict__ slabs) {
    constexpr int AH = BM / 128;                         // 128-row halves of the A tile
    constexpr int BH = BN / 128;                         // 128-column halves of the B tile
    constexpr int STAGE = (AH + BH) * GL_OP;
    constexpr bool BIG = BM == 256 && BN == 256;         // waves 2 (m) × 4 (n), wave tile 128×64
    constexpr int TMF = BIG ? 4 : 2;                     // A fragments (32 rows each) per wave
    constexpr int TNF = BM == 256 ? 2 : 1;               // B fragments (32 columns each) per wave
    constexpr int P = AH + BH;                           // LDS-DMA pieces per wave and k-tile
    constexpr bool TR = sizeof(TC) == 2;                 // bf16 outputs are accumulated transposed (operands swapped)
    constexpr int WROWS = 32 * TMF;                      // rows of a wave tile
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = ks_id * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin + GL_BK - 1) / GL_BK;     // a partial last tile exists only with k-strided operands

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (BM == 256 && !BIG) ? wave >> 1 : wave >> 2, wc = (BM == 256 && !BIG) ? wave & 1 : wave & 3;

    size_t stepA, stepB;
    int kra, krb;
    const __bf16* ga[AH];
#pragma unroll
    for (int h = 0; h < AH; ++h) ga[h] = glds_src<A_KC>(A, lda, m0, M, k_begin, wave, lane, stepA, kra, h);
    const __bf16* gb[BH];
#pragma unroll
    for (int h = 0; h < BH; ++h) gb[h] = glds_src<B_KC>(B, ldb, n0, N, k_begin, wave, lane, stepB, krb, h);
    const __bf16* const zsrc = reinterpret_cast<const __bf16*>(glds_zeros);
    kra += k_begin; krb += k_begin;                           // absolute k-row of this lane in tile 0
    char* const my = smem + wave * 1024;        // this wave's 1-KiB slice inside an operand image

    floatx16 acc[TMF][TNF];
#pragma unroll
    for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < TNF; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define GL_ISSUE(t)                                                                                                       \
    do {                                                                                                                  \
        char* st = my + ((t) % NS) * STAGE;                                                                               \
        const bool za_ = !A_KC && kra + (t) * GL_BK >= k_end;                                                             \
        _Pragma("unroll") for (int h = 0; h < AH; ++h) {                                                                  \
            __builtin_amdgcn_global_load_lds((gl_gptr)(za_ ? zsrc : ga[h]), (gl_lptr)(st + h * GL_OP), 16, 0, 0);        \
            ga[h] += stepA;                                                                                               \
        }                                                                                                                 \
        const bool zb_ = !B_KC && krb + (t) * GL_BK >= k_end;                                                             \
        _Pragma("unroll") for (int h = 0; h < BH; ++h) {                                                                  \
            __builtin_amdgcn_global_load_lds((gl_gptr)(zb_ ? zsrc : gb[h]), (gl_lptr)(st + (AH + h) * GL_OP), 16, 0, 0); \
            gb[h] += stepB;                                                                                               \
        }                                                                                                                 \
    } while (0)

    for (int t = 0; t < NS - 1 && t < nk; ++t) GL_ISSUE(t);

    for (int t = 0; t < nk; ++t) {
        // this wave's P loads of tile t have landed once at most P·(tiles issued after t) remain outstanding
        const int rem = nk - 1 - t;
        if (NS >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
        else if (NS >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave's slice of tile t is in LDS; every wave is done with tile t-1
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) GL_ISSUE(t + NS - 1);  // refills the stage tile t-1 used
        const char* sa = smem + (t % NS) * STAGE;
        const char* sb = sa + AH * GL_OP;
        // this wave's 64 A rows: BM = 128 → rows 64·wr of the single image; BM = 256 → image wr>>1, rows 64·(wr&1)
        // (256×256: the wave's 128 A rows are image wr, its 64 B columns are columns 64·(wc&1) of image wc>>1)
        const char* sa_w = BIG ? sa + wr * GL_OP : BM == 256 ? sa + (wr >> 1) * GL_OP : sa;
        const int arow = BIG ? 0 : BM == 256 ? (wr & 1) * 64 : wr * 64;
        const char* sb_w = BIG ? sb + (wc >> 1) * GL_OP : sb;
        const int bcol = BIG ? (wc & 1) * 64 : wc * 32 * TNF;
#pragma unroll
        for (int ks = 0; ks < GL_BK / 16; ++ks) {
            bf16x8 bfr[TNF];
#pragma unroll
            for (int j = 0; j < TNF; ++j) bfr[j] = glds_fragment<B_KC>(sb_w, bcol + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < TMF; ++i) {
                const bf16x8 af = glds_fragment<A_KC>(sa_w, arow + i * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < TNF; ++j)
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af, acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#undef GL_ISSUE

    if (TR) {      // bf16 output: a row on a lane (see glds_store_rows)
        glds_store_tr<TMF, TNF, TC>(acc, C, ldc, epi, m0 + wr * 32 * TMF, n0 + wc * 32 * TNF, M, N, lane, splitk, ks_id, slabs);
        return;
    }

    const u64 seed = (epi.p_drop > 0.f && splitk == 1) ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < TNF; ++j) {
        const int col = n0 + wc * 32 * TNF + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * WROWS + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row >= M || col >= N) continue;
                if (splitk == 1) epilogue_store_t<TC>(acc[i][j][e], row, col, C, ldc, epi, seed, inv_keep);
                else slabs[((size_t)ks_id * M + row) * N + col] = acc[i][j][e];
            }
    }
}

// ---- 256×256 tile, two wave groups in ping-pong ("pp").  Waves 0-3 and 4-7 (one of each per SIMD) own the upper / lower 128 rows;
// a wave tile is 128×64 = 4×2 MFMA tiles.  Time is cut into barrier intervals; in every interval ONE group runs the 16 MFMAs of a
// 32-deep k-tile (512 matrix-pipe cycles) while the OTHER reads its next 12 fragments from LDS and issues its share of the LDS-DMA
// prefetch — the matrix pipe of a SIMD always has exactly one wave feeding it and the LDS reads hide behind the partner's MFMAs.
//   group 0: reads k-tile t in interval 2t, multiplies it in interval 2t+1;  group 1: one interval later
//   stage ring of 4 × 32 KiB (A0 A1 B0 B1 images of one 32-deep k-tile); tile t+3 is issued from the read phase of tile t
//   (its stage held tile t-1, whose last read retired — lgkmcnt(0) before the barrier — in interval 2t-1);
//   tile t+1 is waited for (counted vmcnt, never 0 in steady state) by every wave at the END of interval 2t+1, one barrier before
//   its first read in interval 2t+2.
template <bool A_KC, bool B_KC, typename TC, bool STAMP = false>
__device__ __forceinline__ void glds_tile_pp(char* __restrict__ smem, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B,
                                             int ldb, TC* __restrict__ C, int ldc, int M, int N, int K, const Epi& epi, int tm, int tn,
                                             unsigned long long* __restrict__ stamps = nullptr) {
    constexpr int NS = 4, STAGE = 4 * GL_OP;
    constexpr bool TR = sizeof(TC) == 2;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = (K + GL_BK - 1) / GL_BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's group

    size_t stepA, stepB;
    int kra, krb;
    const __bf16* ga[2];
    const __bf16* gb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) ga[h] = glds_src<A_KC>(A, lda, m0, M, 0, wave, lane, stepA, kra, h);
#pragma unroll
    for (int h = 0; h < 2; ++h) gb[h] = glds_src<B_KC>(B, ldb, n0, N, 0, wave, lane, stepB, krb, h);
    const __bf16* const zsrc = reinterpret_cast<const __bf16*>(glds_zeros);
    char* const my = smem + wave * 1024;

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the two A pieces (which = 0) or the two B pieces (which = 1) of k-tile t
#define PP_ISSUE(t, which)                                                                                                \
    do {                                                                                                                  \
        char* st = my + ((t) % NS) * STAGE;                                                                               \
        if ((which) == 0) {                                                                                               \
            const bool z_ = !A_KC && kra + (t) * GL_BK >= K;                                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                               \
                __builtin_amdgcn_global_load_lds((gl_gptr)(z_ ? zsrc : ga[h]), (gl_lptr)(st + h * GL_OP), 16, 0, 0);    \
                ga[h] += stepA;                                                                                           \
            }                                                                                                             \
        } else {                                                                                                          \
            const bool z_ = !B_KC && krb + (t) * GL_BK >= K;                                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                               \
                __builtin_amdgcn_global_load_lds((gl_gptr)(z_ ? zsrc : gb[h]), (gl_lptr)(st + (2 + h) * GL_OP), 16, 0, 0); \
                gb[h] += stepB;                                                                                           \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
    // every load of k-tile `t` issued by this wave has landed (tiles t+1 … t+2 may stay in flight)
#define PP_WAIT(t)                                                                                                        \
    do {                                                                                                                  \
        const int rem_ = nk - 1 - (t);                                                                                    \
        if (rem_ >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                   \
        else if (rem_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                              \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    } while (0)

    // development aid: s_memtime stamps of workgroup 0, waves 0 and 4, five per k-tile, kept in the LDS tail and copied out at the end
    unsigned long long* const slog = reinterpret_cast<unsigned long long*>(smem + NS * STAGE) + (wr * 6 * 64);
    const bool stamping = STAMP && blockIdx.x == 0 && (wave & 3) == 0;
#define PP_STAMP(t, i)                                                                                                    \
    do {                                                                                                                  \
        if (STAMP && stamping && (t) < 64) slog[(i) * 64 + (t)] = __builtin_amdgcn_s_memtime();                           \
    } while (0)

    for (int t = 0; t < NS - 1 && t < nk; ++t) { PP_ISSUE(t, 0); PP_ISSUE(t, 1); }
    PP_WAIT(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) {                                       // group 1 runs one interval behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int t = 0; t < nk; ++t) {
        const char* sa_w = smem + (t % NS) * STAGE + wr * GL_OP;
        const char* sb_w = smem + (t % NS) * STAGE + (2 + (wc >> 1)) * GL_OP;
        const int bcol = (wc & 1) * 64;
        // ---- read interval: the 12 fragments of k-tile t, then this wave's 4 pieces of k-tile t+3
        PP_STAMP(t, 0);
        bf16x8 bfr[2][2], afr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[ks][j] = glds_fragment<B_KC>(sb_w, bcol + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[ks][i] = glds_fragment<A_KC>(sa_w, i * 32, ks, lane);
        }
        const bool more = t + NS - 1 < nk;
        if (more) { PP_ISSUE(t + NS - 1, 0); PP_ISSUE(t + NS - 1, 1); }   // (issued between the MFMAs instead they cost more: measured)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_STAMP(t, 1);
        if (wr == 1 && t + 1 < nk) PP_WAIT(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA interval
        PP_STAMP(t, 2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], afr[ks][i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(t, 3);
        if (wr == 0 && t + 1 < nk) PP_WAIT(t + 1);
        PP_STAMP(t, 4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(t, 5);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();           // both groups pass the same number of barriers
#undef PP_ISSUE
#undef PP_WAIT
#undef PP_STAMP
    if (STAMP && stamping && stamps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 6 * 64; i += 64) stamps[wr * 6 * 64 + i] = slog[i];
    }

    if (TR) {      // every wave is past its last LDS read and every LDS-DMA has landed: the ring is free, 16 KiB per wave
        glds_store_tr<4, 2, TC>(acc, C, ldc, epi, m0 + wr * 128, n0 + wc * 64, M, N, lane, 1, 0, nullptr, smem + wave * 16384);
        return;
    }
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row >= M || col >= N) continue;
                epilogue_store_t<TC>(acc[i][j][e], row, col, C, ldc, epi, seed, inv_keep);
            }
    }
}

template <bool A_KC, bool B_KC, typename TC, bool STAMP = false>
__global__ __launch_bounds__(512) void gemm_glds_pp_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                           TC* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_m,
                                                           int tiles_n, int remap, unsigned long long* __restrict__ stamps = nullptr) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * 4 * GL_OP + (STAMP ? 2 * 6 * 64 * 8 : 0)];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    glds_tile_pp<A_KC, B_KC, TC, STAMP>(smem, A, lda, B, ldb, C, ldc, M, N, K, epi, tm, tn, stamps);
}

// (second launch bound = waves per SIMD the register budget must allow: 3 / 2 / 1 workgroups of 8 waves per CU)
template <bool A_KC, bool B_KC, typename TC, int NS, int BM, int BN = 128>
__global__ __launch_bounds__(512, (BM == 256 && BN == 256) ? 2 : BM == 256 ? 4 : NS == 3 ? 6 : 4) void gemm_glds_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                        TC* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_m,
                                                        int tiles_n, int splitk, int k_chunk, float* __restrict__ slabs, int remap) {
    constexpr int STAGE = (BM / 128 + BN / 128) * GL_OP;
    __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n * splitk) : (int)blockIdx.x;
    const int ks_id = wg / (tiles_m * tiles_n);
    const int tile = wg - ks_id * (tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    glds_tile<A_KC, B_KC, TC, NS, BM, BN>(smem, A, lda, B, ldb, C, ldc, M, N, K, epi, tm, tn, ks_id, splitk, k_chunk, slabs);
}

// ---- grouped weight gradients of the bf16 activation streams: up to 48 problems dW_p[n_out, n_in] += dz_pᵀ·x_p (bf16 operands,
// k-strided, any row count) in ONE launch, every 128² tile running its WHOLE k-loop (no split-K slabs, no reduce launches):
// the ≈40 wgrads of a step (22 × K = 19,200 rows, 18 × K = 4,224) are ≈2,300 tiles — enough to fill the chip several times over.
struct G16Prob {
    const __bf16* dz; const __bf16* x; float* dw; int n_out, n_in, rows, ld_dz, ld_x, ld_dw, tile0, tiles_n;
    int whole, split, slab0, stile0;      // ping-pong form: leading tiles run whole, the others in `split` k-parts → slabs (see below)
};
constexpr int G16_MAX = 48;
struct G16Args { int n; int total; G16Prob p[G16_MAX]; };

template <int BM>
__global__ __launch_bounds__(512) void gemm_group_wgrad16_kernel(G16Args g) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * (BM / 128 + 1) * GL_OP];
    const int wg = xcd_remap(blockIdx.x, g.total);
    int pi = 0;
    while (pi + 1 < g.n && wg >= g.p[pi + 1].tile0) ++pi;
    const G16Prob& q = g.p[pi];
    const int tile = wg - q.tile0;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, 1, nullptr};
    glds_tile<false, false, float, 3, BM>(smem, q.dz, q.ld_dz, q.x, q.ld_x, q.dw, q.ld_dw, q.n_out, q.n_in, q.rows, epi, tm, tn, 0, 1,
                                          ((q.rows + GL_BK - 1) / GL_BK) * GL_BK, nullptr);
}

// the same table on 256×256 ping-pong tiles (problems at least 256 wide both ways); deep problems are dealt first.
// Balance: with T tiles of very different depth on 256 CUs, whatever starts after the first round of deep tiles decides the
// makespan (324 tiles of 600 k-tiles: the 68 left over double it).  Tiles dealt after the first round that are still deep
// are therefore cut into `split` k-parts, each a work item of its own writing a 256×256 fp32 slab; `wgrad16_fixup_kernel`
// adds a tile's slabs in part order into dW (deterministic, no atomics).  `tile0` counts work items in this form.
__global__ __launch_bounds__(512) void gemm_group_wgrad16_pp_kernel(G16Args g, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * 4 * GL_OP];
    const int wg = blockIdx.x;
    int pi = 0;
    while (pi + 1 < g.n && wg >= g.p[pi + 1].tile0) ++pi;
    const G16Prob& q = g.p[pi];
    const int item = wg - q.tile0;
    if (item < q.whole) {
        const int tm = item / q.tiles_n, tn = item - tm * q.tiles_n;
        Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, 1, nullptr};
        glds_tile_pp<false, false, float>(smem, q.dz, q.ld_dz, q.x, q.ld_x, q.dw, q.ld_dw, q.n_out, q.n_in, q.rows, epi, tm, tn);
        return;
    }
    const int st = (item - q.whole) / q.split, part = (item - q.whole) - st * q.split;
    const int tile = q.whole + st;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    const int units = (q.rows + GL_BK - 1) / GL_BK, upp = (units + q.split - 1) / q.split;
    const int k0 = min(q.rows, part * upp * GL_BK), k1 = min(q.rows, k0 + upp * GL_BK);
    const int m0 = tm * 256, n0 = tn * 256;
    Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, 0, nullptr};
    float* slab = slabs + (size_t)(q.slab0 + st * q.split + part) * 65536;
    glds_tile_pp<false, false, float>(smem, q.dz + (size_t)k0 * q.ld_dz + m0, q.ld_dz, q.x + (size_t)k0 * q.ld_x + n0, q.ld_x, slab, 256,
                                      min(256, q.n_out - m0), min(256, q.n_in - n0), k1 - k0, epi, 0, 0);
}
__global__ __launch_bounds__(256) void wgrad16_fixup_kernel(G16Args g, const float* __restrict__ slabs) {
    const int bt = blockIdx.x >> 4, chunk = blockIdx.x & 15;     // 16 workgroups per tile, 16 rows each
    int pi = 0;
    while (pi + 1 < g.n && bt >= g.p[pi + 1].stile0) ++pi;
    const G16Prob& q = g.p[pi];
    const int st = bt - q.stile0, tile = q.whole + st;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    const float4* sl = reinterpret_cast<const float4*>(slabs + (size_t)(q.slab0 + st * q.split) * 65536);
#pragma unroll
    for (int it = 0; it < 4; ++it) {                             // 16 rows × 64 float4 per workgroup
        const int i = chunk * 1024 + it * 256 + threadIdx.x;
        const int row = i >> 6, c4 = i & 63;
        const int gr = tm * 256 + row, gc = tn * 256 + 4 * c4;
        if (gr >= q.n_out || gc >= q.n_in) continue;              // n_in % 8 == 0: a float4 is inside or outside as a whole
        float4 v = sl[i];
        for (int p = 1; p < q.split; ++p) {
            const float4 t = sl[(size_t)p * 16384 + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float4* d = reinterpret_cast<float4*>(q.dw + (size_t)gr * q.ld_dw + gc);
        float4 o = *d;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *d = o;
    }
}

// gemm_p8.hip: the 8-phase 256×256×64 form for forward projections (both operands k-contiguous, bf16 output, plain epilogue)
bool glds_p8_supported(int a_kc, int b_kc, int c_dt, int lda, int ldb, int ldc, int M, int N, int K, const Epi& epi, const void* A,
                       const void* B, const void* C);
int glds_p8_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const Epi& epi, int remap,
                   hipStream_t stream);

extern "C" {

// 1 if (shape, layout) can run on the direct-to-LDS kernel: bf16 A and B with 16-byte aligned rows; any M and N (even N) for
// k-contiguous operands, multiples of 8 for k-strided ones (their 16-byte chunks run along the rows); K a multiple of 32 unless
// BOTH operands are k-strided (wgrad), where any K works (k-rows past K come from a block of zeros).
int svpc_gemm_glds_supported(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || lda % 8 != 0 || ldb % 8 != 0 || N % 2 != 0) return 0;
    if (!a_kc && M % 8 != 0) return 0;
    if (!b_kc && N % 8 != 0) return 0;
    if ((a_kc || b_kc) && K % GL_BK != 0) return 0;
    return 1;
}

struct HostWgrad16Problem { const void* dz; const void* x; float* dw; float* db; int n_out, n_in, rows, ld_dz, ld_x, ld_dw; };

// workspace (optional): room for the k-part slabs of the balanced ping-pong form; without it every tile runs whole
int svpc_gemm_group_wgrad_bf16_ws(const void* problems, int n, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n == 0) return 0;
    SVPC_REQUIRE(n > 0 && n <= G16_MAX, "gemm_group_wgrad_bf16: 1..48 problems per launch");
    const HostWgrad16Problem* hp = reinterpret_cast<const HostWgrad16Problem*>(problems);
    G16Args g{};
    g.n = n;
    int tiles = 0;
    static int bm_env = -1;
    if (bm_env < 0) { const char* e = getenv("SVPC_GROUP16_BM"); bm_env = e ? atoi(e) : 256; }
    int BMv = bm_env == 128 ? 128 : 256;          // 256-row tiles (wave tile 64×64) unless a problem is narrower than that
    static int pp_env = -1, split_env = -1, cus = -1;
    if (pp_env < 0) { const char* e = getenv("SVPC_GROUP16_PP"); pp_env = e ? atoi(e) : 1; }
    if (split_env < 0) { const char* e = getenv("SVPC_GROUP16_SPLIT"); split_env = e ? atoi(e) : 160; }   // target k-tiles per part; 0 = never split
    if (cus < 0) {
        hipDeviceProp_t prop; int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    bool pp = pp_env != 0;                        // 256×256 ping-pong tiles when every problem is at least that wide
    for (int i = 0; i < n; ++i) {
        if (hp[i].n_out < 256) BMv = 128;
        if (hp[i].n_out < 256 || hp[i].n_in < 256 || (hp[i].ld_dw & 3) || (((uintptr_t)hp[i].dw) & 15)) pp = false;
    }
    int order[G16_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    if (pp)                                       // deepest reductions first: the short ones fill the tail of the launch
        for (int i = 1; i < n; ++i) {
            const int v = order[i];
            int j = i - 1;
            while (j >= 0 && hp[order[j]].rows < hp[v].rows) { order[j + 1] = order[j]; --j; }
            order[j + 1] = v;
        }
    int items = 0, slabs = 0, stiles = 0, seen = 0;          // seen = tiles dealt so far (in order)
    for (int ii = 0; ii < n; ++ii) {
        const int i = ii;
        const HostWgrad16Problem& h = hp[order[ii]];
        SVPC_REQUIRE(svpc_gemm_glds_supported(0, 0, h.ld_dz, h.ld_x, h.n_out, h.n_in, h.rows) &&
                         ((((uintptr_t)h.dz) | ((uintptr_t)h.x)) & 15) == 0 && h.db == nullptr,
                     "gemm_group_wgrad_bf16: n_out % 8, n_in % 8, 16-byte aligned bf16 rows; no bias output");
        G16Prob& q = g.p[i];
        q.dz = (const __bf16*)h.dz; q.x = (const __bf16*)h.x; q.dw = h.dw; q.n_out = h.n_out; q.n_in = h.n_in; q.rows = h.rows;
        q.ld_dz = h.ld_dz; q.ld_x = h.ld_x; q.ld_dw = h.ld_dw; q.tiles_n = ceil_div(h.n_in, pp ? 256 : GL_BN);
        const int tp = ceil_div(h.n_out, BMv) * q.tiles_n;
        if (!pp) { q.tile0 = tiles; tiles += tp; continue; }
        // tiles of this problem that still belong to the first round run whole; later ones are cut when they are deep
        const int units = ceil_div(h.rows, GL_BK);
        int split = (split_env > 0 && workspace) ? units / split_env : 1;
        if (split > 8) split = 8;
        if (split < 2) split = 1;
        int whole = tp;
        if (split > 1) whole = seen >= cus ? 0 : (cus - seen < tp ? cus - seen : tp);
        if (split > 1 && (size_t)(slabs + (tp - whole) * split) * 65536 * sizeof(float) > workspace_bytes) { split = 1; whole = tp; }
        if (whole == tp) split = 1;
        q.whole = whole; q.split = split; q.slab0 = slabs; q.stile0 = stiles; q.tile0 = items;
        items += whole + (tp - whole) * split;
        if (split > 1) { slabs += (tp - whole) * split; stiles += tp - whole; }
        seen += tp;
    }
    if (pp) {
        g.total = items;
        hipLaunchKernelGGL(gemm_group_wgrad16_pp_kernel, dim3(items), dim3(512), 0, stream, g, workspace);
        int rc = svpc_check_launch("gemm_group_wgrad_bf16");
        if (rc || stiles == 0) return rc;
        hipLaunchKernelGGL(wgrad16_fixup_kernel, dim3(stiles * 16), dim3(256), 0, stream, g, (const float*)workspace);
        return svpc_check_launch("gemm_group_wgrad_bf16 fix-up");
    }
    g.total = tiles;
    if (BMv == 256) hipLaunchKernelGGL(gemm_group_wgrad16_kernel<256>, dim3(tiles), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL(gemm_group_wgrad16_kernel<128>, dim3(tiles), dim3(512), 0, stream, g);
    return svpc_check_launch("gemm_group_wgrad_bf16");
}
int svpc_gemm_group_wgrad_bf16(const void* problems, int n, hipStream_t stream) {
    return svpc_gemm_group_wgrad_bf16_ws(problems, n, nullptr, 0, stream);
}

// A, B bf16; C bf16 (c_dt = 1) or fp32 (c_dt = 0); Z (optional pre-activation copy) has C's type.
int svpc_gemm_glds_rg(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, const void* R,
                      const void* G, int gact, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed,
                      int accumulate, float* workspace, size_t workspace_bytes, hipStream_t stream);
int svpc_gemm_glds_r(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, const void* R,
                     int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                     float* workspace, size_t workspace_bytes, hipStream_t stream) {
    return svpc_gemm_glds_rg(A, lda, a_kc, B, ldb, b_kc, C, c_dt, ldc, Z, R, nullptr, 0, M, N, K, bias, act, p_drop, site, seed, accumulate,
                             workspace, workspace_bytes, stream);
}
int svpc_gemm_glds(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, int M, int N,
                   int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate, float* workspace,
                   size_t workspace_bytes, hipStream_t stream) {
    return svpc_gemm_glds_r(A, lda, a_kc, B, ldb, b_kc, C, c_dt, ldc, Z, nullptr, M, N, K, bias, act, p_drop, site, seed, accumulate, workspace,
                            workspace_bytes, stream);
}
// R (optional): addend of C's type and leading dimension, C = epi(A·B) + R — the residual-path gradient joining a dgrad.
// G (optional, of C's type and leading dimension): C = epi(A·B) ⊙ gact'(G) (+ R) — what the forward of activation `gact` kept (z for
// GELU, y for ReLU / sigmoid): the dgrad whose output is that activation's output gradient applies the activation backward itself.
int svpc_gemm_glds_rg(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, const void* R,
                      const void* G, int gact, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed,
                      int accumulate, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    SVPC_REQUIRE(svpc_gemm_glds_supported(a_kc, b_kc, lda, ldb, M, N, K) && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0,
                 "gemm_glds: unsupported shape (see svpc_gemm_glds_supported) or operands not 16-byte aligned");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "gemm: dropout needs a seed pointer");
    SVPC_REQUIRE(R == nullptr || (c_dt == 1 && p_drop <= 0.f && !accumulate && (N & 7) == 0 && (ldc & 7) == 0 &&
                                  ((((uintptr_t)C) | ((uintptr_t)Z) | ((uintptr_t)bias) | ((uintptr_t)R)) & 15) == 0),
                 "gemm_glds: an addend R needs a bf16 output with N % 8 == 0, 16-byte aligned rows, no dropout / accumulate");
    SVPC_REQUIRE(G == nullptr || (c_dt == 1 && p_drop <= 0.f && !accumulate && (N & 7) == 0 && (ldc & 7) == 0 && gact >= ACT_NONE &&
                                  gact <= ACT_SIGMOID && ((((uintptr_t)C) | ((uintptr_t)Z) | ((uintptr_t)bias) | ((uintptr_t)G)) & 15) == 0),
                 "gemm_glds: an activation-backward factor G needs a bf16 output with N % 8 == 0, 16-byte aligned rows, no dropout / accumulate");
    Epi epi{bias, act, p_drop, site, seed, accumulate, (float*)Z, R, G, gact};
    // 256-row tiles when 128-row tiles would need more than one round of resident workgroups and the taller tile fills its rounds
    // better (3 × 48 KiB workgroups per CU vs 2 × 72 KiB)
    static int bm_env = -1;
    if (bm_env < 0) { const char* e = getenv("SVPC_GLDS_BM"); bm_env = e ? atoi(e) : 0; }
    const int t128 = ceil_div(M, 128) * ceil_div(N, GL_BN), t256 = ceil_div(M, 256) * ceil_div(N, GL_BN);
    const int slots128 = 256 * 3, slots256 = 256 * 2;
    const double eff128 = (double)t128 / ((double)ceil_div(t128, slots128) * slots128);
    const double eff256 = (double)t256 / ((double)ceil_div(t256, slots256) * slots256);
    int BMv = (t128 > slots128 && eff256 >= eff128) ? 256 : 128;
    static int wgrad_bm = -1;
    if (wgrad_bm < 0) { const char* e = getenv("SVPC_GLDS_WGRAD_BM"); wgrad_bm = e ? atoi(e) : 128; }
    if (!a_kc && !b_kc && M >= 256 && wgrad_bm == 256) BMv = 256;      // wgrad: split-K supplies the parallelism
    if (bm_env == 128 || bm_env == 256) BMv = bm_env;
    // 256×256 tiles (wave tile 128×64, 4-stage ring of 32 KiB stages = 128 KiB, one workgroup per CU): half the LDS reads and
    // half the L2→LDS bytes per MFMA of the 256×128 tile; taken when such tiles alone fill most of the chip (≥ 150 of them)
    static int big_env = -1;
    if (big_env < 0) { const char* e = getenv("SVPC_GLDS_BIG"); big_env = e ? atoi(e) : 1; }
    const int t_big = ceil_div(M, 256) * ceil_div(N, 256);
    static int big_min = -1;
    if (big_min < 0) { const char* e = getenv("SVPC_GLDS_BIG_MIN"); big_min = e ? atoi(e) : 150; }   // measured on 13 / 10 / 8-video batches: 183 tiles win, ≤ 141 lose
    const bool big = big_env && (a_kc || b_kc) && t_big >= big_min;
    const int BNv = big ? 256 : GL_BN;
    if (big) BMv = 256;
    const int tiles_m = ceil_div(M, BMv), tiles_n = ceil_div(N, BNv), tiles = tiles_m * tiles_n;
    int splitk = 1;
    static int split_below = -1;
    if (split_below < 0) { const char* e = getenv("SVPC_GLDS_SPLIT_BELOW"); split_below = e ? atoi(e) : 150; }   // ≥150 tiles already fill most CUs: a split would only add the reduce launch (measured)
    static int split_target = -1;
    if (split_target < 0) { const char* e = getenv("SVPC_GLDS_SPLIT_TARGET"); split_target = e ? atoi(e) : 512; }   // two workgroups per CU (measured best of 128…1024 when the step replays as one linear chain)
    if (K >= 512 && tiles < split_below && R == nullptr && G == nullptr) {
        splitk = ceil_div(split_target, tiles);
        const int max_by_k = K / 256;
        if (splitk > max_by_k) splitk = max_by_k;
        if (splitk > 64) splitk = 64;
        while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > workspace_bytes) --splitk;
        if (splitk < 1) splitk = 1;
    }
    int k_chunk = ceil_div(ceil_div(K, splitk), GL_BK) * GL_BK;
    splitk = ceil_div(K, k_chunk);
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    static int ppdbg = -1;
    if (ppdbg < 0) { const char* e = getenv("SVPC_PP_DBG"); ppdbg = e ? atoi(e) : 0; }
    static int p8_env = -1;
    if (p8_env < 0) { const char* e = getenv("SVPC_P8"); p8_env = e ? atoi(e) : 1; }
    if (big && p8_env && glds_p8_supported(a_kc, b_kc, c_dt, lda, ldb, ldc, M, N, K, epi, A, B, C))
        return glds_p8_launch(A, lda, B, ldb, C, ldc, M, N, K, epi, remap, stream);
    dim3 grid(tiles * splitk), block(512);
    // 3 stages = 48 KiB → 3 workgroups per CU (measured best of 3 / 4)
#define GL_LAUNCH1(AK, BKC, TC, NSV, BMV)                                                                                            \
    hipLaunchKernelGGL((gemm_glds_kernel<AK, BKC, TC, NSV, BMV>), grid, block, 0, stream, (const __bf16*)A, lda, (const __bf16*)B, ldb,   \
                       (TC*)C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk, workspace, remap)
#define GL_LAUNCH(AK, BKC, TC)                                                                                                       \
    do {                                                                                                                             \
        if (big)                                                                                                                     \
            hipLaunchKernelGGL((gemm_glds_pp_kernel<AK, BKC, TC>), grid, block, 0, stream, (const __bf16*)A, lda, (const __bf16*)B,  \
                               ldb, (TC*)C, ldc, M, N, K, epi, tiles_m, tiles_n, remap, nullptr);                                    \
        else if (BMv == 256) GL_LAUNCH1(AK, BKC, TC, 3, 256);                                                                          \
        else GL_LAUNCH1(AK, BKC, TC, 3, 128);                                                                                        \
    } while (0)
    if (big && (ppdbg & 8) && c_dt == 1 && a_kc && b_kc && workspace) {      // development aid: stamped instance, stamps → workspace
        hipLaunchKernelGGL((gemm_glds_pp_kernel<true, true, __bf16, true>), grid, block, 0, stream, (const __bf16*)A, lda, (const __bf16*)B,
                           ldb, (__bf16*)C, ldc, M, N, K, epi, tiles_m, tiles_n, remap, (unsigned long long*)workspace);
        return svpc_check_launch("gemm_glds pp stamped");
    }
    if (c_dt == 1) {
        if (a_kc && b_kc) GL_LAUNCH(true, true, __bf16);
        else if (a_kc && !b_kc) GL_LAUNCH(true, false, __bf16);
        else if (!a_kc && b_kc) GL_LAUNCH(false, true, __bf16);
        else GL_LAUNCH(false, false, __bf16);
    } else {
        if (a_kc && b_kc) GL_LAUNCH(true, true, float);
        else if (a_kc && !b_kc) GL_LAUNCH(true, false, float);
        else if (!a_kc && b_kc) GL_LAUNCH(false, true, float);
        else GL_LAUNCH(false, false, float);
    }
#undef GL_LAUNCH
#undef GL_LAUNCH1
    int rc = svpc_check_launch("gemm_glds");
    if (rc) return rc;
    if (splitk > 1) {
        if (c_dt == 0) launch_splitk_reduce<float>(workspace, splitk, (float*)C, ldc, M, N, epi, stream);
        else launch_splitk_reduce<__bf16>(workspace, splitk, (__bf16*)C, ldc, M, N, epi, stream);
        rc = svpc_check_launch("gemm_glds splitk reduce");
    }
    return rc;
}

}  // extern "C"
