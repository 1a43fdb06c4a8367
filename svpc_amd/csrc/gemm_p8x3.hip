// Three-term split-bf16 GEMM ("bf16x3") for the forward projections of the activation streams in the ≤1e-4-parity throughput mode:
//
//   C = act( A·Bᵀ + bias ),  A = A_hi + A_lo,  B = B_hi + B_lo  (each an exact sum of two bf16 planes, 16-17 significant bits)
//   A·Bᵀ ≈ A_lo·B_hiᵀ + A_hi·B_hiᵀ + A_hi·B_loᵀ          (the dropped A_lo·B_lo term is ≤ 2⁻¹⁶ relative)
//
// The structure, LDS images, barriers and prefetch schedule are those of gemm_p8.hip (256×256 tiles, 8 phases per pair of staged buffers,
// both operands direct-to-LDS, v_mfma_f32_16x16x32_bf16; see the comment there) with ONE difference in what a staged buffer holds: the
// two 32-wide "k blocks" of a half-tile image are the HI and the LO plane of the same 32-deep k-slice (not two consecutive k blocks of
// one plane).  A buffer then carries everything the three products of its slice need — A_hi, A_lo, B_hi, B_lo fetched ONCE, their
// fragments read from LDS ONCE — and every phase issues 24 MFMAs (lo·hi, hi·hi, hi·lo per output block) on the 12 fragments that fed 16:
// the kernel is bound by the LDS fill rate of a CU (≈40–50 GB/s direct-to-LDS from L2 / Infinity Cache: MI355X_MICROARCH.md "gather
// into LDS"), so 1.5× the matrix work per staged byte is what pays (round 3, first form: a 3K-deep contraction of three plane pairs,
// every hi tile staged twice — 6 tiles per 64-deep slice instead of 4).  fp32 accumulation of bf16 products is exact per product, so the
// result carries ≈ fp32 accuracy at 3× the bf16 MFMA work — against 16× for the f32 MFMA (MI355X_MICROARCH.md: f32-input MFMA runs at
// 1/16 of the bf16 rate).
//
// Storage ("split16", see layernorm.hip): a row of A holds its hi plane at columns [0, K) and its lo plane at [a_lo, a_lo + K); the
// weight shadow keeps two planes of identical layout b_lo elements apart; C is written the same way (hi at column c, lo at c_lo + c).
// The hi plane alone is the RNE-rounded bf16 tensor the bf16 backward kernels (dgrad / wgrad / LayerNorm / attention) read in place.
// The optional pre-activation copy Z (what the GELU backward differentiates) is a plain bf16 matrix.
//
// reference ops replaced: every nn.Linear forward of the clip encoder (src/rtransformer/model.py:195-197, :230, :259, :281, :551).
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * x3_gptr;
typedef void __attribute__((address_space(3))) * x3_lptr;

constexpr int X3_BK = 64;            // width of a staged half-tile image: 2 planes × 32 k
constexpr int X3_KS = 32;            // k-slice per staged buffer
constexpr int X3_HALF = 128 * X3_BK * 2;      // 16 KiB: one staged half-tile
constexpr int X3_BUF = 4 * X3_HALF;           // 64 KiB: SA0 SA1 SB0 SB1 of one k-tile

__device__ __forceinline__ uint32_t x3_pack2(float lo, float hi) {
    union { __bf16 h[2]; uint32_t u; } pk;
    pk.h[0] = (__bf16)lo; pk.h[1] = (__bf16)hi;
    return pk.u;
}
// residual of the bf16 rounding, itself rounded to bf16: v ≈ hi + lo to 2⁻¹⁷
__device__ __forceinline__ float x3_lo(float v) { return v - (float)(__bf16)v; }

// erf by Abramowitz-Stegun 7.1.26 (|error| ≤ 1.5e-7 absolute on erf, i.e. ≈1e-7 relative on gelu — below the 2⁻¹⁷ of the stored pair)
__device__ __forceinline__ float x3_gelu(float x) {
    const float u = x * 0.70710678118654752f, au = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-au * au);
    const float erf_abs = fmaf(-p * t, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, u));
}
template <int ACT>
__device__ __forceinline__ float x3_act(float z) {
    if (ACT == ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ACT_GELU) return x3_gelu(z);
    return z;
}

// one epilogue pass: acc (+bias) → MODE 0: bf16(z) (pre-activation copy) | 1: hi plane of act(z) | 2: lo plane of act(z)
// → wave-private LDS image [128 rows][128 B] → whole-line stores (as gemm_p8.hip's p8_store_pass)
template <int ACT, int MODE>
__device__ __forceinline__ void x3_store_pass(const floatx4 (&acc)[8][4], __bf16* __restrict__ C, int ldc, const float4 (&bb)[4], int row0,
                                              int col0, int M, int N, int lane, char* __restrict__ wl) {
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float z0 = acc[i][j][0] + bb[j].x, z1 = acc[i][j][1] + bb[j].y, z2 = acc[i][j][2] + bb[j].z, z3 = acc[i][j][3] + bb[j].w;
            if (MODE != 0) { z0 = x3_act<ACT>(z0); z1 = x3_act<ACT>(z1); z2 = x3_act<ACT>(z2); z3 = x3_act<ACT>(z3); }
            if (MODE == 2) { z0 = x3_lo(z0); z1 = x3_lo(z1); z2 = x3_lo(z2); z3 = x3_lo(z3); }
            uint2 v;
            v.x = x3_pack2(z0, z1); v.y = x3_pack2(z2, z3);
            const int c16 = j * 2 + (q >> 1);           // 16-byte chunk of the row; this lane's 8 bytes are its half (q & 1)
            *reinterpret_cast<uint2*>(wl + r * 128 + ((c16 ^ (r & 7)) << 4) + ((q & 1) << 3)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);              // one row block at a time: keeps the register footprint of the tail small
    }
    // same wave wrote and reads: LDS operations of a wave complete in order.  One 32-bit byte offset per lane (host check: < 4 GiB)
    const int chunk = lane & 7, cc = col0 + 8 * chunk, r0 = lane >> 3;
    uint32_t off = ((uint32_t)(row0 + r0) * (uint32_t)ldc + (uint32_t)cc) * 2u;
    const uint32_t step = 16u * (uint32_t)ldc;                         // 8 rows
    const bool col_ok = cc + 8 <= N;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + r0;
        const uint4 v = *reinterpret_cast<const uint4*>(wl + r * 128 + ((chunk ^ (r & 7)) << 4));
        if (col_ok && row0 + r < M) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(C) + off) = v;
        off += step;
        if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four lines in flight
    }
}

template <int ACT, bool HASZ>
__global__ __launch_bounds__(512) void gemm_p8x3_kernel(const __bf16* __restrict__ A, int lda, int a_lo, const __bf16* __restrict__ B, int ldb,
                                                        long long b_lo, __bf16* __restrict__ C, int ldc, int c_lo, __bf16* __restrict__ Z,
                                                        int ldz, const float* __restrict__ bias, int M, int N, int K, int tiles_m,
                                                        int tiles_n, int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * X3_BUF];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K / X3_KS;                            // staged buffers = 32-deep k-slices (hi and lo plane of both operands)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's group = its 128-row half; wc = its 64-column strip

    // ---- staging (as gemm_p8.hip): this wave fills the subtiles (row block `wave`, k blocks 0 and 1) of every half-tile
    const int sr = lane >> 2, sc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const __bf16* ga[2];
    const __bf16* gb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ga[h] = A + (size_t)min(m0 + 128 * h + 16 * wave + sr, M - 1) * lda + 8 * sc;
        gb[h] = B + (size_t)min(n0 + 128 * h + 16 * wave + sr, N - 1) * ldb + 8 * sc;
    }
    char* const my = smem + wave * 2048;
    // buffer t = k-slice t (32 deep): "k block" 0 of every half-tile image is the slice of the hi plane, "k block" 1 the slice of the
    // lo plane (element offsets from the hi-plane row pointers).  The half-tile `which` (0 SA0, 1 SA1, 2 SB0, 3 SB1): two 1-KiB pieces per wave
#define X3_STAGE(t, which)                                                                                                 \
    do {                                                                                                                   \
        char* dst_ = my + ((t) & 1) * X3_BUF + (which) * X3_HALF;                                                          \
        const __bf16* src_ = ((which) < 2 ? ga[(which) & 1] : gb[(which) & 1]) + (size_t)(t) * X3_KS;                      \
        __builtin_amdgcn_global_load_lds((x3_gptr)(src_), (x3_lptr)(dst_), 16, 0, 0);                                      \
        __builtin_amdgcn_global_load_lds((x3_gptr)(src_ + ((which) < 2 ? (size_t)a_lo : (size_t)b_lo)), (x3_lptr)(dst_ + 1024), 16, 0, 0); \
    } while (0)
    // every DMA of k-tile t+1 issued by this wave has landed (the two B halves of tile t+2, issued after them, may stay in flight)
#define X3_WAIT(t)                                                                                                         \
    do {                                                                                                                   \
        if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    } while (0)
#define X3_SYNC()                                                                                                          \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)

    // ---- fragment reads: block `blk` (16 rows) and k block kb of a half-tile image; lane: row lane&15, logical chunk lane>>4
    const int fr_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
#define X3_FRAG(img, blk, kb) (*reinterpret_cast<const bf16x8*>((img) + (((blk) * 2 + (kb)) << 10) + fr_off))

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: buffer 0 whole, the B halves of buffer 1 (K ≥ 64: there always is a slice 1)
#pragma unroll
    for (int w = 0; w < 4; ++w) X3_STAGE(0, w);
    X3_STAGE(1, 2); X3_STAGE(1, 3);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    X3_SYNC();
    if (wr == 1) X3_SYNC();                              // group 1 runs one interval behind

    const int cb0 = (wc & 1) * 4;                        // the wave's first 16-column block inside its B half-tile
    for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t & 1) * X3_BUF + wr * X3_HALF;
        const char* sb = smem + (t & 1) * X3_BUF + (2 + (wc >> 1)) * X3_HALF;
        bf16x8 afr[2][4], b0[2][2], b1[2][2];
        // ---- phase 0: rows 0-63 × columns 0-31 of the wave tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[kb][j] = X3_FRAG(sb, cb0 + j, kb);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = X3_FRAG(sa, i, kb);
        }
        if (t + 1 < nk) X3_STAGE(t + 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X3_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)            // lo·hi, hi·hi, hi·lo: plane 1 = lo
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[term == 2 ? 1 : 0][j], afr[term == 0 ? 1 : 0][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        X3_SYNC();
        // ---- phase 1: rows 0-63 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[kb][j] = X3_FRAG(sb, cb0 + 2 + j, kb);
        if (t + 1 < nk) X3_STAGE(t + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X3_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)            // lo·hi, hi·hi, hi·lo: plane 1 = lo
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[term == 2 ? 1 : 0][j], afr[term == 0 ? 1 : 0][i], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        X3_SYNC();
        // ---- phase 2: rows 64-127 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = X3_FRAG(sa, 4 + i, kb);
        if (t + 2 < nk) X3_STAGE(t + 2, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X3_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)            // lo·hi, hi·hi, hi·lo: plane 1 = lo
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[term == 2 ? 1 : 0][j], afr[term == 0 ? 1 : 0][i], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        X3_SYNC();
        // ---- phase 3: rows 64-127 × columns 0-31 (fragments already in registers)
        if (t + 2 < nk) X3_STAGE(t + 2, 3);
        if (wr == 1 && t + 1 < nk) X3_WAIT(t);
        X3_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < 3; ++term)            // lo·hi, hi·hi, hi·lo: plane 1 = lo
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[term == 2 ? 1 : 0][j], afr[term == 0 ? 1 : 0][i], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0 && t + 1 < nk) X3_WAIT(t);
        X3_SYNC();
    }
    if (wr == 0) X3_SYNC();                              // both groups pass the same number of barriers: 2 + 8·nk
#undef X3_STAGE
#undef X3_WAIT
#undef X3_SYNC
#undef X3_FRAG

    // every wave is past its last LDS read and every LDS-DMA has landed: the ring is free, 16 KiB per wave
    char* wl = smem + wave * 16384;
    const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
    float4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col0 + j * 16 + 4 * (lane >> 4);
        bb[j] = (bias && c + 4 <= N) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (HASZ) x3_store_pass<ACT, 0>(acc, Z, ldz, bb, row0, col0, M, N, lane, wl);
    x3_store_pass<ACT, 2>(acc, C + c_lo, ldc, bb, row0, col0, M, N, lane, wl);
    x3_store_pass<ACT, 1>(acc, C, ldc, bb, row0, col0, M, N, lane, wl);
}

extern "C" {

// 1 if (shape, layout) runs on this kernel
int svpc_gemm_p8x3_supported(int lda, int a_lo, int ldb, int ldc, int c_lo, int ldz, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 2 * X3_KS || K % X3_KS != 0 || (N & 7) != 0) return 0;
    if ((lda & 7) || (a_lo & 7) || (ldb & 7) || (ldc & 7) || (c_lo & 7) || (ldz & 7)) return 0;
    if (a_lo < K || lda < a_lo + K || c_lo < N || ldc < c_lo + N || ldb < K) return 0;
    if ((unsigned long long)M * (unsigned long long)ldc * 2ull >= (1ull << 32)) return 0;
    if ((unsigned long long)M * (unsigned long long)(ldz > 0 ? ldz : 1) * 2ull >= (1ull << 32)) return 0;
    return 1;
}

// C (split) = act(A (split) · B (split)ᵀ + bias); A [M][lda] with planes at columns 0 / a_lo; B_hi [N][ldb] and B_lo = B_hi + b_lo
// elements (same layout); C [M][ldc] planes at columns 0 / c_lo; Z (optional, only with an activation): plain bf16 [M][ldz]
// pre-activation (without an activation it would equal C's hi plane).
int svpc_gemm_p8x3(const void* A, int lda, int a_lo, const void* B, int ldb, long long b_lo, void* C, int ldc, int c_lo, void* Z, int ldz,
                   int M, int N, int K, const float* bias, int act, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (Z == nullptr) ldz = 0;
    SVPC_REQUIRE(svpc_gemm_p8x3_supported(lda, a_lo, ldb, ldc, c_lo, ldz, M, N, K) == 1 && (b_lo & 7) == 0 &&
                     ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)Z) | ((uintptr_t)bias)) & 15) == 0 &&
                     (act == ACT_RELU || act == ACT_GELU || (act == ACT_NONE && Z == nullptr)) && (Z == nullptr || ldz >= N),
                 "gemm_p8x3: needs K % 32 == 0, K >= 64, N % 8 == 0, 16-byte aligned split rows (lo plane behind the hi plane), act in {none, relu, gelu}");
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 256);
#define X3_GO(ACTV, ZV)                                                                                                          \
    hipLaunchKernelGGL((gemm_p8x3_kernel<ACTV, ZV>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, a_lo,  \
                       (const __bf16*)B, ldb, b_lo, (__bf16*)C, ldc, c_lo, (__bf16*)Z, ldz, bias, M, N, K, tiles_m, tiles_n, remap)
    const bool z = Z != nullptr;
    if (act == ACT_GELU) { if (z) X3_GO(ACT_GELU, true); else X3_GO(ACT_GELU, false); }
    else if (act == ACT_RELU) { if (z) X3_GO(ACT_RELU, true); else X3_GO(ACT_RELU, false); }
    else X3_GO(ACT_NONE, false);
#undef X3_GO
    return svpc_check_launch("gemm_p8x3");
}

}  // extern "C"
