// bf16 × bf16 → bf16 GEMM for the forward projections of the bf16 activation streams ("p8"): 256×256×64 tiles, 8 phases per pair of
// k-tiles, both operands k-contiguous and DIRECT-TO-LDS (global_load_lds_dwordx4), v_mfma_f32_16x16x32_bf16.
//
//   C[M,N] = act( A[M,K]·B[N,K]ᵀ + bias )  (+ optional pre-activation copy Z),  K % 64 == 0, any M, N % 8 == 0
//
// Structure (cdna_hip_programming.md §5 "The 256² 8-phase template", rebuilt here for this library's operand layouts and epilogues):
//   * 8 waves = two groups (waves 0-3 / 4-7, one of each per SIMD) running ONE barrier interval apart: while a group issues the 16
//     MFMAs of a phase (a 64×32 quadrant of its 128×64 wave tile over the 64-deep k-tile), the other reads the fragments of its
//     next phase from LDS and issues its share of the LDS-DMA prefetch — a SIMD's matrix pipe always has exactly one wave feeding it.
//   * LDS: two 64-KiB buffers, each four staged half-tiles (A rows 0-127 / 128-255, B columns 0-127 / 128-255 of one k-tile), a
//     half-tile = 16 subtiles of [16 rows][32 k] bf16 (1 KiB = one wave-instruction of the DMA, lane-linear).  ds_read_b128 bank
//     conflicts are removed by the st_16x32 swizzle (16-byte chunk c of rows 8-15 kept at c ^ 2), applied on the per-lane SOURCE
//     address of the DMA and on the read address (never on the LDS destination).
//   * Prefetch: phase p of k-tile t stages half-tile p of what comes next — A halves of tile t+1 in phases 0,1 (their buffer's last
//     A read was phase 2 of tile t-1), B halves of tile t+2 in phases 2,3 (this buffer's last B read was phase 1 of tile t; the
//     bh = 0 fragments needed again in phase 3 stay in registers).  ONE counted wait per k-tile (vmcnt(4): the two B halves of tile
//     t+2 stay in flight), placed one barrier before the first read of tile t+1 by either group; raw s_barrier only.
//   * Operands swapped in the MFMA (weights in the A slot): a lane holds 4 consecutive columns of ONE output row, the epilogue
//     packs them to bf16 and passes them through a wave-private, XOR-swizzled LDS image so that every global store instruction
//     writes 8 whole 128-byte lines (the ring is free by then).
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * p8_gptr;
typedef void __attribute__((address_space(3))) * p8_lptr;

constexpr int P8_BK = 64;
constexpr int P8_HALF = 128 * P8_BK * 2;      // 16 KiB: one staged half-tile
constexpr int P8_BUF = 4 * P8_HALF;           // 64 KiB: SA0 SA1 SB0 SB1 of one k-tile

__device__ __forceinline__ uint32_t p8_pack2(float lo, float hi) {
    union { __bf16 h[2]; uint32_t u; } pk;
    pk.h[0] = (__bf16)lo; pk.h[1] = (__bf16)hi;
    return pk.u;
}

// erf for the GELU epilogue of a bf16 output: Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 + the 1-ulp rcp/exp: three orders below the
// 2^-9 rounding of the stored value), branch-free, ~14 instructions per element instead of the ~60 of the exact library routine —
// the tail of a 256x256 tile evaluates it 65,536 times per workgroup with nothing left to overlap it.  The pre-activation copy Z
// (what the backward differentiates) is the exact sum; the fp32 parity mode never comes here (gemm.hip, exact erff).
__device__ __forceinline__ float p8_gelu(float x) {
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
__device__ __forceinline__ float p8_act(float z) {
    if (ACT == ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ACT_GELU) return p8_gelu(z);
    return z;
}

// one epilogue pass: acc (+bias, activation unless PRE) → bf16 → wave-private LDS image [128 rows][128 B] → whole-line stores
template <int ACT, bool PRE>
__device__ __forceinline__ void p8_store_pass(const floatx4 (&acc)[8][4], __bf16* __restrict__ C, int ldc, const float4 (&bb)[4], int row0,
                                              int col0, int M, int N, int lane, char* __restrict__ wl) {
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z0 = acc[i][j][0] + bb[j].x, z1 = acc[i][j][1] + bb[j].y, z2 = acc[i][j][2] + bb[j].z, z3 = acc[i][j][3] + bb[j].w;
            uint2 v;
            if (PRE) { v.x = p8_pack2(z0, z1); v.y = p8_pack2(z2, z3); }
            else { v.x = p8_pack2(p8_act<ACT>(z0), p8_act<ACT>(z1)); v.y = p8_pack2(p8_act<ACT>(z2), p8_act<ACT>(z3)); }
            const int c16 = j * 2 + (q >> 1);           // 16-byte chunk of the row; this lane's 8 bytes are its half (q & 1)
            *reinterpret_cast<uint2*>(wl + r * 128 + ((c16 ^ (r & 7)) << 4) + ((q & 1) << 3)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);              // one row block at a time: keeps the register footprint of the tail small
    }
    // same wave wrote and reads: LDS operations of a wave complete in order.  Global addresses as ONE 32-bit byte offset per lane
    // from the uniform base (host check: the output spans < 4 GiB) stepped by a uniform stride — 16 precomputed 64-bit
    // addresses would not fit beside the 128 accumulator registers the Z pass still needs.
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
__device__ __forceinline__ void p8_store(const floatx4 (&acc)[8][4], __bf16* __restrict__ C, int ldc, __bf16* __restrict__ Z,
                                         const float* __restrict__ bias, int row0, int col0, int M, int N, int lane, char* __restrict__ wl) {
    float4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col0 + j * 16 + 4 * (lane >> 4);
        bb[j] = (bias && c + 4 <= N) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // pre-activation copy first: the activation pass then consumes the sums for the last time and its registers free up as it goes
    if (HASZ) p8_store_pass<ACT, true>(acc, Z, ldc, bb, row0, col0, M, N, lane, wl);
    p8_store_pass<ACT, false>(acc, C, ldc, bb, row0, col0, M, N, lane, wl);
}

template <int ACT, bool HASZ>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                                      __bf16* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_m,
                                                      int tiles_n, int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * P8_BUF];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K / P8_BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's group = its 128-row half; wc = its 64-column strip

    // ---- staging: this wave fills the subtiles (row block `wave`, k blocks 0 and 1) of every half-tile.  LDS slot (row r = lane>>2,
    // chunk slot lane&3) of a subtile holds logical 16-byte chunk (lane&3) ^ 2·(r >= 8).
    const int sr = lane >> 2, sc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const __bf16* ga[2];
    const __bf16* gb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ga[h] = A + (size_t)min(m0 + 128 * h + 16 * wave + sr, M - 1) * lda + 8 * sc;
        gb[h] = B + (size_t)min(n0 + 128 * h + 16 * wave + sr, N - 1) * ldb + 8 * sc;
    }
    char* const my = smem + wave * 2048;
    // the half-tile `which` (0 SA0, 1 SA1, 2 SB0, 3 SB1) of k-tile t: two 1-KiB pieces per wave
#define P8_STAGE(t, which)                                                                                                 \
    do {                                                                                                                   \
        char* dst_ = my + ((t) & 1) * P8_BUF + (which) * P8_HALF;                                                          \
        const __bf16* src_ = ((which) < 2 ? ga[(which) & 1] : gb[(which) & 1]) + (size_t)(t) * P8_BK;                      \
        /* (marking the activation loads non-temporal measured 10-25 % slower: the tiles of a tile row share them through L2) */ \
        __builtin_amdgcn_global_load_lds((p8_gptr)(src_), (p8_lptr)(dst_), 16, 0, 0);                                      \
        __builtin_amdgcn_global_load_lds((p8_gptr)(src_ + 32), (p8_lptr)(dst_ + 1024), 16, 0, 0);                          \
    } while (0)
    // every DMA of k-tile t+1 issued by this wave has landed (the two B halves of tile t+2, issued after them, may stay in flight)
#define P8_WAIT(t)                                                                                                         \
    do {                                                                                                                   \
        if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    } while (0)
#define P8_SYNC()                                                                                                          \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)

    // ---- fragment reads: block `blk` (16 rows) and k block kb of a half-tile image; lane: row lane&15, logical chunk lane>>4
    const int fr_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
#define P8_FRAG(img, blk, kb) (*reinterpret_cast<const bf16x8*>((img) + (((blk) * 2 + (kb)) << 10) + fr_off))

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: k-tile 0 whole, the B halves of k-tile 1
#pragma unroll
    for (int w = 0; w < 4; ++w) P8_STAGE(0, w);
    if (nk > 1) { P8_STAGE(1, 2); P8_STAGE(1, 3); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_SYNC();
    if (wr == 1) P8_SYNC();                              // group 1 runs one interval behind

    const int cb0 = (wc & 1) * 4;                        // the wave's first 16-column block inside its B half-tile
    for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t & 1) * P8_BUF + wr * P8_HALF;
        const char* sb = smem + (t & 1) * P8_BUF + (2 + (wc >> 1)) * P8_HALF;
        bf16x8 afr[2][4], b0[2][2], b1[2][2];
        // ---- phase 0: rows 0-63 × columns 0-31 of the wave tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[kb][j] = P8_FRAG(sb, cb0 + j, kb);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8_FRAG(sa, i, kb);
        }
        if (t + 1 < nk) P8_STAGE(t + 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8_SYNC();
        // ---- phase 1: rows 0-63 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[kb][j] = P8_FRAG(sb, cb0 + 2 + j, kb);
        if (t + 1 < nk) P8_STAGE(t + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8_SYNC();
        // ---- phase 2: rows 64-127 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8_FRAG(sa, 4 + i, kb);
        if (t + 2 < nk) P8_STAGE(t + 2, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8_SYNC();
        // ---- phase 3: rows 64-127 × columns 0-31 (fragments already in registers)
        if (t + 2 < nk) P8_STAGE(t + 2, 3);
        if (wr == 1 && t + 1 < nk) P8_WAIT(t);
        P8_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0 && t + 1 < nk) P8_WAIT(t);
        P8_SYNC();
    }
    if (wr == 0) P8_SYNC();                              // both groups pass the same number of barriers: 2 + 8·nk
#undef P8_STAGE
#undef P8_WAIT
#undef P8_SYNC
#undef P8_FRAG

    // every wave is past its last LDS read and every LDS-DMA has landed: the ring is free, 16 KiB per wave
    char* wl = smem + wave * 16384;
    __bf16* Zb = reinterpret_cast<__bf16*>(epi.Z);
    const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
    p8_store<ACT, HASZ>(acc, C, ldc, Zb, epi.bias, row0, col0, M, N, lane, wl);
}

// 1 if this (shape, epilogue) runs on the p8 kernel; the launcher of gemm_glds.hip asks before choosing its own 256×256 form
bool glds_p8_supported(int a_kc, int b_kc, int c_dt, int lda, int ldb, int ldc, int M, int N, int K, const Epi& epi, const void* A,
                       const void* B, const void* C) {
    return a_kc && b_kc && c_dt == 1 && K >= P8_BK && K % P8_BK == 0 && (N & 7) == 0 && (ldc & 7) == 0 && (lda & 7) == 0 &&
           (ldb & 7) == 0 && epi.p_drop <= 0.f && !epi.accumulate && epi.R == nullptr && epi.G == nullptr && epi.act != ACT_SIGMOID &&
           (unsigned long long)M * (unsigned long long)ldc * 2ull < (1ull << 32) &&
           ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)epi.Z) | ((uintptr_t)epi.bias)) & 15) == 0;
}
int glds_p8_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const Epi& epi, int remap,
                   hipStream_t stream) {
    const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 256);
    // one instantiation per (activation, pre-activation copy): each epilogue is register-allocated on its own
#define P8_GO(ACTV, ZV)                                                                                                         \
    hipLaunchKernelGGL((gemm_p8_kernel<ACTV, ZV>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, (const __bf16*)B, \
                       ldb, (__bf16*)C, ldc, M, N, K, epi, tiles_m, tiles_n, remap)
    const bool z = epi.Z != nullptr;
    if (epi.act == ACT_GELU) { if (z) P8_GO(ACT_GELU, true); else P8_GO(ACT_GELU, false); }
    else if (epi.act == ACT_RELU) { if (z) P8_GO(ACT_RELU, true); else P8_GO(ACT_RELU, false); }
    else { if (z) P8_GO(ACT_NONE, true); else P8_GO(ACT_NONE, false); }
#undef P8_GO
    return svpc_check_launch("gemm_p8");
}

extern "C" int svpc_gemm_p8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* Z, int M, int N, int K, const float* bias,
                            int act, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    Epi epi{bias, act, 0.f, 0u, nullptr, 0, (float*)Z, nullptr};
    SVPC_REQUIRE(glds_p8_supported(1, 1, 1, lda, ldb, ldc, M, N, K, epi, A, B, C),
                 "gemm_p8: needs K % 64 == 0, N % 8 == 0, 16-byte aligned bf16 rows, act in {none, relu, gelu}");
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    return glds_p8_launch(A, lda, B, ldb, C, ldc, M, N, K, epi, remap, stream);
}
