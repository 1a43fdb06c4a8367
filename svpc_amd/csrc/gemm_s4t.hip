// bf16 GEMM with a K-STRIDED B operand on 128×128 tiles ("s4t") — the input gradients (dgrad) of the DECODER's projections
// (4,224 sentence rows, 576 memory rows; reference: the backward of the nn.Linear layers of BertDecoderLayerNoMemoryUntied,
// src/rtransformer/model.py:620-663):
//
//   C[M,N] = A[M,K] · B[K,N] + R[M,N]        A = dz (k-contiguous rows), B = the weight matrix W[K = out][N = in] as it is stored,
//                                            R (optional, bf16, C's layout) = the residual-path gradient parked for this tensor
//
// The small-M sibling of gemm_p8t.hip: at M = 4,224 a 256×256 tiling yields 51 workgroups for 256 CUs; 128×128 tiles give 198.
// Tiling and pipeline of gemm_s4x3.hip (8 waves, wave tile 32×64 = 2×4 v_mfma_f32_16x16x32_bf16 tiles, weights in the MFMA's A slot, NS
// stages of 32 KiB with a counted vmcnt, one barrier per k-tile); the A image is the st_16x32 subtile layout of gemm_p8.hip, the B image
// is gemm_p8t.hip's [64 k-rows][128 columns] (the weight rows as they lie in memory, 1 KiB = 4 k-rows per DMA wave-instruction, 16-byte
// chunk c of k-row r kept at slot c ^ 2·((r & 3) | ((r >> 1) & 4))) read through ds_read_b64_tr_b16.
// Epilogue: fp32 sums through a wave-private LDS image, + R, one rounding to bf16, whole 128-byte lines.
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short4s __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * s4t_gptr;
typedef void __attribute__((address_space(3))) * s4t_lptr;

constexpr int S4T_BK = 64;
constexpr int S4T_HALF = 128 * S4T_BK * 2;      // 16 KiB: the A tile image, the B tile image
constexpr int S4T_STAGE = 2 * S4T_HALF;

__device__ __forceinline__ bf16x8 s4t_frag_tr(const char* __restrict__ a) {      // rows r..r+3 (this call) and r+4..r+7 of a lane's column
    typedef short4s __attribute__((address_space(3))) * lds_ptr;
    const short4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const short4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * 256));
    union { short s[8]; bf16x8 v; } u;
    u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
    u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
    return u.v;
}

template <bool HASR, int NS>
__global__ __launch_bounds__(512, NS == 2 ? 2 : 1) void gemm_s4t_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                                        __bf16* __restrict__ C, int ldc, const __bf16* __restrict__ R, int M,
                                                                        int N, int K, int tiles_m, int tiles_n, int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[NS * S4T_STAGE];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int nk = K / S4T_BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;             // the wave's 32-row strip (of four) / 64-column half

    // ---- staging.  A: this wave fills row block `wave` (16 rows, k blocks 0 and 1) of the A image (st_16x32: LDS slot (row lane>>2, chunk
    // slot lane&3) of a subtile holds logical 16-byte chunk (lane&3) ^ 2·(row >= 8)).  B: pieces 2·wave and 2·wave + 1 of the B image;
    // piece pi = k-rows 4·pi … 4·pi + 3, lane l ↔ (k-row 4·pi + (l >> 4), slot l & 15) holding logical chunk slot ^ f(k-row)
    const int sr = lane >> 2, sc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const __bf16* const ga = A + (size_t)min(m0 + 16 * wave + sr, M - 1) * lda + 8 * sc;
    const __bf16* gb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kr = 8 * wave + 4 * u + (lane >> 4);
        const int fx = 2 * ((kr & 3) | ((kr >> 1) & 4));
        const int col = min(n0 + 8 * ((lane & 15) ^ fx), N - 8);      // columns past N are clamped (their sums are never stored)
        gb[u] = B + (size_t)kr * ldb + col;
    }
    const size_t stepB = (size_t)S4T_BK * ldb;
#define S4T_ISSUE(t)                                                                                                              \
    do {                                                                                                                          \
        char* st_ = smem + ((t) % NS) * S4T_STAGE;                                                                                \
        const __bf16* sa_ = ga + (size_t)(t) * S4T_BK;                                                                            \
        __builtin_amdgcn_global_load_lds((s4t_gptr)(sa_), (s4t_lptr)(st_ + ((wave * 2) << 10)), 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds((s4t_gptr)(sa_ + 32), (s4t_lptr)(st_ + ((wave * 2 + 1) << 10)), 16, 0, 0);               \
        __builtin_amdgcn_global_load_lds((s4t_gptr)(gb[0] + (size_t)(t) * stepB), (s4t_lptr)(st_ + S4T_HALF + wave * 2048), 16, 0, 0);        \
        __builtin_amdgcn_global_load_lds((s4t_gptr)(gb[1] + (size_t)(t) * stepB), (s4t_lptr)(st_ + S4T_HALF + wave * 2048 + 1024), 16, 0, 0); \
    } while (0)
    // ---- fragment reads.  A: block `blk` (16 rows), k block kb; lane: row lane&15, logical chunk lane>>4
    const int fr_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
#define S4T_FRAGA(img, blk, kb) (*reinterpret_cast<const bf16x8*>((img) + (((blk) * 2 + (kb)) << 10) + fr_off))
    // B: 16-column block cb (0..7) and k block kb of the [64][128] image; lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) reads k-rows
    // 32·kb + 8g + q (+4) at logical chunk 2·cb + (p>>1), byte (p&1)·8; both rows share f = 2·(q + 4·(g&1))   (gemm_p8t.hip)
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tfx = 2 * (tq + 4 * (tg & 1));
    const int tb_off = (8 * tg + tq) * 256 + ((tp & 1) << 3);
    const int tp1 = tp >> 1;
#define S4T_FRAGB(img, cb, kb) s4t_frag_tr((img) + (kb) * (32 * 256) + tb_off + (((((2 * (cb)) ^ tfx)) | tp1) << 4))

    floatx4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) S4T_ISSUE(t);
    for (int t = 0; t < nk; ++t) {
        if (NS == 4 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (NS >= 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of k-tile t have landed
        __builtin_amdgcn_s_barrier();                            // … everyone's have; everyone is done reading k-tile t-1
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) S4T_ISSUE(t + NS - 1);              // into the stage k-tile t-1 used
        const char* sa = smem + (t % NS) * S4T_STAGE;
        const char* sb = sa + S4T_HALF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 bfr[4], afr[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = S4T_FRAGB(sb, wc * 4 + j, kb);
#pragma unroll
            for (int i = 0; i < 2; ++i) afr[i] = S4T_FRAGA(sa, wr * 2 + i, kb);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], afr[i], acc[i][j], 0, 0, 0);
        }
    }
#undef S4T_ISSUE
#undef S4T_FRAGA
#undef S4T_FRAGB
    __syncthreads();                                             // every wave is past its last LDS read: the stages are free
    // ---- epilogue: the wave's 32×64 fp32 block → wave-private image [32 rows][256 B] (16-byte chunk c of row r at c ^ (r & 7)) → a lane
    // takes 8 consecutive columns of a row: + R, bf16, one 16-byte store (8 lanes = one 128-byte line)
    char* wl = smem + wave * 8192;
    const int row0 = m0 + wr * 32, col0 = n0 + wc * 64;
    const int l15 = lane & 15, q = lane >> 4;
    const int c8 = lane & 7, cc = col0 + 8 * c8, r0 = lane >> 3;
    const bool col_ok = cc + 8 <= N;
    bf16x8 rv[HASR ? 4 : 1];                                     // the R pieces of all four row groups are requested first (see gemm_p8t.hip)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + r0;
        if (HASR && col_ok && row0 + r < M) rv[it] = *reinterpret_cast<const bf16x8*>(R + (size_t)(row0 + r) * ldc + cc);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c16 = j * 4 + q;                           // 16-byte chunk (4 fp32 columns) of the 64-column row
            *reinterpret_cast<floatx4*>(wl + r * 256 + ((c16 ^ (r & 7)) << 4)) = acc[i][j];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + r0;
        const floatx4 v0 = *reinterpret_cast<const floatx4*>(wl + r * 256 + (((2 * c8) ^ (r & 7)) << 4));
        const floatx4 v1 = *reinterpret_cast<const floatx4*>(wl + r * 256 + (((2 * c8 + 1) ^ (r & 7)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (col_ok && row0 + r < M) {
            const size_t o = (size_t)(row0 + r) * ldc + cc;
            if (HASR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)rv[it][j];
            }
            bf16x8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (__bf16)v[j];
            *reinterpret_cast<bf16x8*>(C + o) = ov;
        }
    }
}

extern "C" {

// 1 if (shape, layout) runs on this kernel
int svpc_gemm_s4t_supported(int lda, int ldb, int ldc, int M, int N, int K) {
    if (M <= 0 || N < 8 || K < S4T_BK || K % S4T_BK != 0 || (N & 7) != 0) return 0;
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || lda < K || ldb < N || ldc < N) return 0;
    return 1;
}

// C[M,N] (bf16) = A[M,K] (bf16, k-contiguous rows) · B[K,N] (bf16, k-strided: row k holds the N columns) + R (optional, bf16, C's layout);
// all pointers 16-byte aligned.  Replaces svpc_gemm_glds_rg's (a_kc = 1, b_kc = 0) form for the decoder's dgrads.
int svpc_gemm_s4t(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* R, int M, int N, int K, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    SVPC_REQUIRE(svpc_gemm_s4t_supported(lda, ldb, ldc, M, N, K) == 1 &&
                     ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)R)) & 15) == 0,
                 "gemm_s4t: needs K % 64 == 0, N % 8 == 0, 16-byte aligned bf16 rows");
    static int remap = -1, deep_max = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    if (deep_max < 0) { const char* e = getenv("SVPC_S4T_DEEP_MAX"); deep_max = e ? atoi(e) : 256; }
    const int tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128);
    const bool deep = tiles_m * tiles_n <= deep_max;     // at most one workgroup per CU anyway: four stages
#define S4T_GO(RV, NSV)                                                                                                           \
    hipLaunchKernelGGL((gemm_s4t_kernel<RV, NSV>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, (const __bf16*)B, \
                       ldb, (__bf16*)C, ldc, (const __bf16*)R, M, N, K, tiles_m, tiles_n, remap)
    if (R != nullptr) { if (deep) S4T_GO(true, 4); else S4T_GO(true, 2); }
    else { if (deep) S4T_GO(false, 4); else S4T_GO(false, 2); }
#undef S4T_GO
    return svpc_check_launch("gemm_s4t");
}

}  // extern "C"
