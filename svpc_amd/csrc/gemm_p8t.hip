// bf16 GEMM with a K-STRIDED B operand on the 8-phase 256×256×64 template ("p8t") — the input gradients (dgrad) of the bf16 activation
// streams' projections (reference: the backward of every nn.Linear of the clip encoder, src/rtransformer/model.py:195-197, :230, :259,
// :281, :551):
//
//   C[M,N] = ( A[M,K] · B[K,N] ) ⊙ act'(G[M,N]) + R[M,N]        A = dz (k-contiguous rows), B = the weight matrix W[K = out][N = in] as it is stored
//
// G (optional): what the forward of the activation in front of this projection's input kept (z for GELU / ReLU) — the dgrad applies the
// activation's backward in its epilogue; R (optional): the residual-path gradient a LayerNorm backward parked for this tensor.  Both bf16,
// C's layout.  Same contract as svpc_gemm_glds_rg's stream dgrads, which ran on the round-1 32-deep ping-pong kernel (≈0.26 of peak).
//
// Structure = gemm_p8.hip (two wave groups one barrier interval apart, 8 phases per pair of k-tiles, both operands direct-to-LDS, one
// counted vmcnt per k-tile, v_mfma_f32_16x16x32_bf16, weights in the A slot) with ONE difference: the B half-tile image is
// [64 k-rows][128 columns] (the weight rows as they lie in memory, 1 KiB = 4 k-rows per DMA wave-instruction) and its MFMA fragments are
// fetched with ds_read_b64_tr_b16: within a group of 16 lanes the instruction reads a 4-row × 16-column block and hands lane li the four
// rows of column li — two of them give a lane the 8 consecutive k of its column, exactly the 16×16×32 operand layout.  Bank conflicts:
// rows are 256 bytes (= all 64 banks), so the 16-byte chunk c of k-row r is kept at slot c ^ 2·((r & 3) | ((r >> 1) & 4)) — the eight rows
// a half-wave touches land in eight disjoint 8-bank windows (swizzle applied on the per-lane DMA source address and on the read address).
// Epilogue: fp32 sums through a wave-private LDS image in two 32-column passes, act'(G), + R, one rounding to bf16, 16-byte stores.
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short4t __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * p8t_gptr;
typedef void __attribute__((address_space(3))) * p8t_lptr;

constexpr int P8T_BK = 64;
constexpr int P8T_HALF = 128 * P8T_BK * 2;      // 16 KiB: one staged half-tile
constexpr int P8T_BUF = 4 * P8T_HALF;           // 64 KiB: SA0 SA1 SB0 SB1 of one k-tile

__device__ __forceinline__ bf16x8 p8t_frag_tr(const char* __restrict__ a) {      // rows r..r+3 (this call) and r+4..r+7 of a lane's column
    typedef short4t __attribute__((address_space(3))) * lds_ptr;
    const short4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const short4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * 256));
    union { short s[8]; bf16x8 v; } u;
    u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
    u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
    return u.v;
}

// one 32-column half of the wave's 128×64 block: fp32 sums → LDS image [128 rows][128 B] (16-byte chunk c of row r at c ^ (r & 7)) →
// a lane takes 8 consecutive columns of a row: ⊙ act'(G), + R, bf16, one 16-byte store
template <int GACT, bool HASR>
__device__ __forceinline__ void p8t_store_half(const floatx4 (&acc)[8][4], int jh, __bf16* __restrict__ C, const __bf16* __restrict__ G,
                                               const __bf16* __restrict__ R, int ldc, int row0, int col0, int M, int N, int lane,
                                               char* __restrict__ wl) {
    const int l15 = lane & 15, q = lane >> 4;
    const int c8 = lane & 3, cc = col0 + 32 * jh + 8 * c8, r0 = lane >> 2;      // 8 columns = chunks 2·c8, 2·c8 + 1
    const bool col_ok = cc + 8 <= N;
    // the G / R pieces of all 8 row groups are requested FIRST and land under the LDS staging: written as load → use → store per row
    // group the compiler has to keep every load behind the previous store (16 dependent round trips per tile: + 10 µs for R, + 18 for G)
    bf16x8 gv[GACT != ACT_NONE ? 8 : 1], rv[HASR ? 8 : 1];
    // an interior block (the common case) runs without per-row predicates: with them every store sits in its own exec-masked region
    // and the compiler drains the memory counter in front of each (8 store round trips per pass instead of 1)
    const bool interior = row0 + 128 <= M && col0 + 32 * jh + 32 <= N;      // wave-uniform
#define P8T_EPI_LOADS(GUARD)                                                                                               \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                                     \
        const int r = it * 16 + r0;                                                                                        \
        if (!(GUARD) || (col_ok && row0 + r < M)) {                                                                        \
            const size_t o = (size_t)(row0 + r) * ldc + cc;                                                                \
            if (GACT != ACT_NONE) gv[it] = *reinterpret_cast<const bf16x8*>(G + o);                                        \
            if (HASR) rv[it] = *reinterpret_cast<const bf16x8*>(R + o);                                                    \
        }                                                                                                                  \
    }
#define P8T_EPI_STORES(GUARD)                                                                                              \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                                     \
        const int r = it * 16 + r0;                                                                                        \
        const floatx4 v0 = *reinterpret_cast<const floatx4*>(wl + r * 128 + (((2 * c8) ^ (r & 7)) << 4));                  \
        const floatx4 v1 = *reinterpret_cast<const floatx4*>(wl + r * 128 + (((2 * c8 + 1) ^ (r & 7)) << 4));              \
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};                                             \
        if (!(GUARD) || (col_ok && row0 + r < M)) {                                                                        \
            const size_t o = (size_t)(row0 + r) * ldc + cc;                                                                \
            if (GACT != ACT_NONE) {                                                                                        \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] *= act_grad_from_aux((float)gv[it][j], GACT, true);     \
            }                                                                                                              \
            if (HASR) {                                                                                                    \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] += (float)rv[it][j];                                    \
            }                                                                                                              \
            bf16x8 ov;                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) ov[j] = (__bf16)v[j];                                            \
            *reinterpret_cast<bf16x8*>(C + o) = ov;                                                                        \
        }                                                                                                                  \
    }
    if (interior) { P8T_EPI_LOADS(false) } else { P8T_EPI_LOADS(true) }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + l15;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int c16 = jj * 4 + q;                 // 16-byte chunk (4 fp32 columns) of the 32-column row
            *reinterpret_cast<floatx4*>(wl + r * 128 + ((c16 ^ (r & 7)) << 4)) = acc[i][2 * jh + jj];
        }
    }
    if (interior) { P8T_EPI_STORES(false) } else { P8T_EPI_STORES(true) }
#undef P8T_EPI_LOADS
#undef P8T_EPI_STORES
}

template <int GACT, bool HASR>
__global__ __launch_bounds__(512) void gemm_p8t_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                       __bf16* __restrict__ C, int ldc, const __bf16* __restrict__ G,
                                                       const __bf16* __restrict__ R, int M, int N, int K, int tiles_m, int tiles_n, int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * P8T_BUF];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K / P8T_BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's group = its 128-row half; wc = its 64-column strip

    // ---- staging.  A (k-contiguous rows): as gemm_p8.hip — this wave fills the subtiles (row block `wave`, k blocks 0 and 1) of both
    // A half-tiles.  B (k-strided): pieces 2·wave and 2·wave + 1 of both B half-tiles; piece pi = k-rows 4·pi … 4·pi + 3, lane l ↔
    // (k-row 4·pi + (l >> 4), slot l & 15), holding logical chunk slot ^ f(k-row)
    const int sr = lane >> 2, sc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const __bf16* ga[2];
    const __bf16* gb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ga[h] = A + (size_t)min(m0 + 128 * h + 16 * wave + sr, M - 1) * lda + 8 * sc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kr = 8 * wave + 4 * u + (lane >> 4);
            const int fx = 2 * ((kr & 3) | ((kr >> 1) & 4));
            const int col = min(n0 + 128 * h + 8 * ((lane & 15) ^ fx), N - 8);      // columns past N are clamped (their sums are never stored)
            gb[h][u] = B + (size_t)kr * ldb + col;
        }
    }
    const size_t stepB = (size_t)P8T_BK * ldb;
    char* const my = smem + wave * 2048;
    // the half-tile `which` (0 SA0, 1 SA1, 2 SB0, 3 SB1) of k-tile t: two 1-KiB pieces per wave
#define P8T_STAGE(t, which)                                                                                                \
    do {                                                                                                                   \
        char* dst_ = my + ((t) & 1) * P8T_BUF + (which) * P8T_HALF;                                                        \
        if ((which) < 2) {                                                                                                 \
            const __bf16* src_ = ga[(which) & 1] + (size_t)(t) * P8T_BK;                                                   \
            __builtin_amdgcn_global_load_lds((p8t_gptr)(src_), (p8t_lptr)(dst_), 16, 0, 0);                                \
            __builtin_amdgcn_global_load_lds((p8t_gptr)(src_ + 32), (p8t_lptr)(dst_ + 1024), 16, 0, 0);                    \
        } else {                                                                                                           \
            __builtin_amdgcn_global_load_lds((p8t_gptr)(gb[(which) & 1][0] + (size_t)(t) * stepB), (p8t_lptr)(dst_), 16, 0, 0);        \
            __builtin_amdgcn_global_load_lds((p8t_gptr)(gb[(which) & 1][1] + (size_t)(t) * stepB), (p8t_lptr)(dst_ + 1024), 16, 0, 0); \
        }                                                                                                                  \
    } while (0)
#define P8T_WAIT(t)                                                                                                        \
    do {                                                                                                                   \
        if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    } while (0)
#define P8T_SYNC()                                                                                                         \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)

    // ---- fragment reads.  A: block `blk` (16 rows), k block kb of a half-tile image; lane: row lane&15, logical chunk lane>>4
    const int fr_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
#define P8T_FRAGA(img, blk, kb) (*reinterpret_cast<const bf16x8*>((img) + (((blk) * 2 + (kb)) << 10) + fr_off))
    // B: 16-column block cb (0..7) and k block kb of a [64][128] image; lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) reads k-rows
    // 32·kb + 8g + q (+4) at logical chunk 2·cb + (p>>1), byte (p&1)·8; both rows share f = 2·(q + 4·(g&1))
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tfx = 2 * (tq + 4 * (tg & 1));
    const int tb_off = (8 * tg + tq) * 256 + ((tp & 1) << 3);
    const int tp1 = tp >> 1;
#define P8T_FRAGB(img, cb, kb) p8t_frag_tr((img) + (kb) * (32 * 256) + tb_off + (((((2 * (cb)) ^ tfx)) | tp1) << 4))

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: k-tile 0 whole, the B halves of k-tile 1
#pragma unroll
    for (int w = 0; w < 4; ++w) P8T_STAGE(0, w);
    if (nk > 1) { P8T_STAGE(1, 2); P8T_STAGE(1, 3); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8T_SYNC();
    if (wr == 1) P8T_SYNC();                              // group 1 runs one interval behind

    const int cb0 = (wc & 1) * 4;                        // the wave's first 16-column block inside its B half-tile
    for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t & 1) * P8T_BUF + wr * P8T_HALF;
        const char* sb = smem + (t & 1) * P8T_BUF + (2 + (wc >> 1)) * P8T_HALF;
        bf16x8 afr[2][4], b0[2][2], b1[2][2];
        // ---- phase 0: rows 0-63 × columns 0-31 of the wave tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[kb][j] = P8T_FRAGB(sb, cb0 + j, kb);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8T_FRAGA(sa, i, kb);
        }
        if (t + 1 < nk) P8T_STAGE(t + 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8T_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8T_SYNC();
        // ---- phase 1: rows 0-63 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[kb][j] = P8T_FRAGB(sb, cb0 + 2 + j, kb);
        if (t + 1 < nk) P8T_STAGE(t + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8T_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8T_SYNC();
        // ---- phase 2: rows 64-127 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8T_FRAGA(sa, 4 + i, kb);
        if (t + 2 < nk) P8T_STAGE(t + 2, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8T_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8T_SYNC();
        // ---- phase 3: rows 64-127 × columns 0-31 (fragments already in registers)
        if (t + 2 < nk) P8T_STAGE(t + 2, 3);
        if (wr == 1 && t + 1 < nk) P8T_WAIT(t);
        P8T_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0 && t + 1 < nk) P8T_WAIT(t);
        P8T_SYNC();
    }
    if (wr == 0) P8T_SYNC();                              // both groups pass the same number of barriers: 2 + 8·nk
#undef P8T_STAGE
#undef P8T_WAIT
#undef P8T_SYNC
#undef P8T_FRAGA
#undef P8T_FRAGB

    // every wave is past its last LDS read and every LDS-DMA has landed: the ring is free, 16 KiB per wave
    char* wl = smem + wave * 16384;
    const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
    p8t_store_half<GACT, HASR>(acc, 0, C, G, R, ldc, row0, col0, M, N, lane, wl);
    p8t_store_half<GACT, HASR>(acc, 1, C, G, R, ldc, row0, col0, M, N, lane, wl);
}

extern "C" {

// 1 if this (shape, layout) runs on the p8t kernel: A bf16 [M][lda] k-contiguous, B bf16 [K][ldb] (k-strided), C / G / R bf16 [M][ldc]
int svpc_gemm_p8t_supported(int lda, int ldb, int ldc, int M, int N, int K) {
    return (M > 0 && N >= 8 && (N & 7) == 0 && K >= P8T_BK && K % P8T_BK == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 7) == 0 && lda >= K &&
            ldb >= N && ldc >= N) ? 1 : 0;
}

// C = (A·B) ⊙ gact'(G) + R.  gact: SVPC_ACT_NONE / RELU / GELU (G = the pre-activation or activated tensor the forward kept; the
// derivative as svpc_act_bwd, GELU by the A&S erf of svpc_gemm_glds_rg's epilogue); G, R optional (NULL).
int svpc_gemm_p8t(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* G, int gact, const void* R, int M, int N, int K,
                  hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (G == nullptr) gact = ACT_NONE;
    SVPC_REQUIRE(svpc_gemm_p8t_supported(lda, ldb, ldc, M, N, K) == 1 &&
                     ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)G) | ((uintptr_t)R)) & 15) == 0 &&
                     (gact == ACT_NONE || gact == ACT_RELU || gact == ACT_GELU),
                 "gemm_p8t: needs K % 64 == 0, N % 8 == 0, 16-byte aligned bf16 rows, gact in {none, relu, gelu}");
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 256);
#define P8T_GO(GA, HR)                                                                                                           \
    hipLaunchKernelGGL((gemm_p8t_kernel<GA, HR>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, (const __bf16*)B, \
                       ldb, (__bf16*)C, ldc, (const __bf16*)G, (const __bf16*)R, M, N, K, tiles_m, tiles_n, remap)
    const bool hr = R != nullptr;
    if (gact == ACT_GELU) { if (hr) P8T_GO(ACT_GELU, true); else P8T_GO(ACT_GELU, false); }
    else if (gact == ACT_RELU) { if (hr) P8T_GO(ACT_RELU, true); else P8T_GO(ACT_RELU, false); }
    else { if (hr) P8T_GO(ACT_NONE, true); else P8T_GO(ACT_NONE, false); }
#undef P8T_GO
    return svpc_check_launch("gemm_p8t");
}

}  // extern "C"
