// Three-term split-bf16 GEMM on 128×128 tiles ("s4x3") — the small-M sibling of gemm_p8x3.hip for the decoder's projections in the
// bf16x3 mode (4,224 sentence rows, 576 memory rows: reference src/rtransformer/model.py:620-663, the Linear layers of
// BertDecoderLayerNoMemoryUntied).  At M = 4,224 a 256×256 tiling yields 51 (N = 768) … 153 (N = 2304) workgroups for 256 CUs and a
// launch lasts as long as ONE tile's k-loop (57–62 µs measured); 128×128 tiles give 198 … 594 workgroups of a quarter of the work.
//
//   C (split) = act( A (split) · B (split)ᵀ + bias ),  A·Bᵀ ≈ A_lo·B_hiᵀ + A_hi·B_hiᵀ + A_hi·B_loᵀ
//
// 8 waves, wave tile 32×64 = 2×4 v_mfma_f32_16x16x32_bf16 tiles (two waves per SIMD cover each other's LDS-read latency: 31 → 28 µs at
// M = 4,224, N = 768), weights in the MFMA's A slot (a lane holds 4 consecutive columns of one output row, as gemm_p8x3.hip); both
// operands direct-to-LDS in the st_16x32 subtile layout of gemm_p8.hip (one 1-KiB wave-instruction per [16 rows][32 k] subtile, swizzle
// on the DMA source address and on the read address).  A staged 32-KiB buffer holds the hi AND the lo plane of one 32-deep k-slice of
// both operands (as gemm_p8x3.hip): 12 fragment reads feed 24 MFMAs.  NS = 2 stages with two workgroups per CU, or NS = 4 when the grid
// leaves at most one workgroup per CU anyway (three slices in flight).  Epilogue as gemm_p8x3.hip (bias, ReLU / GELU, hi / lo planes
// through a wave-private LDS image, whole-line stores; optional plain-bf16 pre-activation copy Z).
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * s4_gptr;
typedef void __attribute__((address_space(3))) * s4_lptr;

constexpr int S4_BK = 64;            // width of a staged tile image: 2 planes × 32 k
constexpr int S4_KS = 32;            // k-slice per staged buffer
constexpr int S4_HALF = 128 * S4_BK * 2;      // 16 KiB: 128 rows × 64 k
constexpr int S4_STAGE = 2 * S4_HALF;         // A tile + B tile

__device__ __forceinline__ uint32_t s4_pack2(float lo, float hi) {
    union { __bf16 h[2]; uint32_t u; } pk;
    pk.h[0] = (__bf16)lo; pk.h[1] = (__bf16)hi;
    return pk.u;
}
__device__ __forceinline__ float s4_lo(float v) { return v - (float)(__bf16)v; }
__device__ __forceinline__ float s4_gelu(float x) {      // erf by Abramowitz-Stegun 7.1.26 (|error| ≤ 1.5e-7), as gemm_p8x3.hip
    const float u = x * 0.70710678118654752f, au = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-au * au);
    return 0.5f * x * (1.0f + copysignf(fmaf(-p * t, e, 1.0f), u));
}
template <int ACT>
__device__ __forceinline__ float s4_act(float z) {
    if (ACT == ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ACT_GELU) return s4_gelu(z);
    return z;
}

// one epilogue pass over the wave's 32×64 block: MODE 0 bf16(z) | 1 hi plane of act(z) | 2 lo plane → wave-private LDS image
// [32 rows][128 B] (16-byte chunk c of row r kept at c ^ (r & 7)) → whole 128-byte lines
template <int ACT, int MODE>
__device__ __forceinline__ void s4_store_pass(const floatx4 (&acc)[2][4], __bf16* __restrict__ C, int ldc, const float4 (&bb)[4], int row0,
                                              int col0, int M, int N, int lane, char* __restrict__ wl) {
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float z0 = acc[i][j][0] + bb[j].x, z1 = acc[i][j][1] + bb[j].y, z2 = acc[i][j][2] + bb[j].z, z3 = acc[i][j][3] + bb[j].w;
            if (MODE != 0) { z0 = s4_act<ACT>(z0); z1 = s4_act<ACT>(z1); z2 = s4_act<ACT>(z2); z3 = s4_act<ACT>(z3); }
            if (MODE == 2) { z0 = s4_lo(z0); z1 = s4_lo(z1); z2 = s4_lo(z2); z3 = s4_lo(z3); }
            uint2 v;
            v.x = s4_pack2(z0, z1); v.y = s4_pack2(z2, z3);
            const int c16 = j * 2 + (q >> 1);
            *reinterpret_cast<uint2*>(wl + r * 128 + ((c16 ^ (r & 7)) << 4) + ((q & 1) << 3)) = v;
        }
    }
    const int chunk = lane & 7, cc = col0 + 8 * chunk, r0 = lane >> 3;
    uint32_t off = ((uint32_t)(row0 + r0) * (uint32_t)ldc + (uint32_t)cc) * 2u;
    const uint32_t step = 16u * (uint32_t)ldc;                         // 8 rows
    const bool col_ok = cc + 8 <= N;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + r0;
        const uint4 v = *reinterpret_cast<const uint4*>(wl + r * 128 + ((chunk ^ (r & 7)) << 4));
        if (col_ok && row0 + r < M) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(C) + off) = v;
        off += step;
    }
}

template <int ACT, bool HASZ, int NS>
__global__ __launch_bounds__(512, NS == 2 ? 2 : 1) void gemm_s4x3_kernel(const __bf16* __restrict__ A, int lda, int a_lo, const __bf16* __restrict__ B, int ldb,
                                                           long long b_lo, __bf16* __restrict__ C, int ldc, int c_lo, __bf16* __restrict__ Z,
                                                           int ldz, const float* __restrict__ bias, int M, int N, int K, int tiles_m,
                                                           int tiles_n, int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[NS * S4_STAGE];
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n) : (int)blockIdx.x;
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int nk = K / S4_KS;                            // staged buffers = 32-deep k-slices, hi and lo plane of both operands (gemm_p8x3.hip)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;             // the wave's 32-row strip (of four) / 64-column half

    // ---- staging: a tile image = 8 row blocks (16 rows) × 2 k blocks (32 k) of 1-KiB subtiles, subtile (rb, kb) at ((rb·2 + kb) << 10);
    // this wave fills row blocks wave and wave + 4 (both k blocks) of the A and of the B image: 8 DMA instructions per k-tile.
    // LDS slot (row r = lane>>2, chunk slot lane&3) of a subtile holds logical 16-byte chunk (lane&3) ^ 2·(r >= 8)  (st_16x32)
    const int sr = lane >> 2, sc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const __bf16* const ga = A + (size_t)min(m0 + 16 * wave + sr, M - 1) * lda + 8 * sc;
    const __bf16* const gb = B + (size_t)min(n0 + 16 * wave + sr, N - 1) * ldb + 8 * sc;
    // buffer t = k-slice t: "k block" 0 of a tile image is the slice of the hi plane, "k block" 1 the slice of the lo plane
#define S4_ISSUE(t)                                                                                                               \
    do {                                                                                                                          \
        char* da_ = smem + ((t) % NS) * S4_STAGE + ((wave * 2) << 10);                                                            \
        const size_t o_ = (size_t)(t) * S4_KS;                                                                                    \
        __builtin_amdgcn_global_load_lds((s4_gptr)(ga + o_), (s4_lptr)(da_), 16, 0, 0);                                           \
        __builtin_amdgcn_global_load_lds((s4_gptr)(ga + o_ + a_lo), (s4_lptr)(da_ + 1024), 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((s4_gptr)(gb + o_), (s4_lptr)(da_ + S4_HALF), 16, 0, 0);                                 \
        __builtin_amdgcn_global_load_lds((s4_gptr)(gb + o_ + b_lo), (s4_lptr)(da_ + S4_HALF + 1024), 16, 0, 0);                   \
    } while (0)
    // fragment: block `blk` (16 rows) and k block kb of an image; lane: row lane&15, logical chunk lane>>4
    const int fr_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);
#define S4_FRAG(img, blk, kb) (*reinterpret_cast<const bf16x8*>((img) + (((blk) * 2 + (kb)) << 10) + fr_off))

    floatx4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // NS stages: k-tiles t+1 … t+NS-2 are in flight while k-tile t is awaited (4 DMA instructions per wave and k-tile: the counted
    // wait lets the later ones stay outstanding); NS = 2 is the plain double buffer
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) S4_ISSUE(t);
    for (int t = 0; t < nk; ++t) {
        if (NS == 4 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (NS >= 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of k-tile t have landed
        __builtin_amdgcn_s_barrier();                            // … everyone's have; everyone is done reading k-tile t-1
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) S4_ISSUE(t + NS - 1);               // into the stage k-tile t-1 used; lands under the MFMAs of the next k-tiles
        const char* sa = smem + (t % NS) * S4_STAGE;
        const char* sb = sa + S4_HALF;
        bf16x8 bfr[2][4], afr[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[pl][j] = S4_FRAG(sb, wc * 4 + j, pl);
#pragma unroll
            for (int i = 0; i < 2; ++i) afr[pl][i] = S4_FRAG(sa, wr * 2 + i, pl);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)            // lo·hi, hi·hi, hi·lo: plane 1 = lo
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[term == 2 ? 1 : 0][j], afr[term == 0 ? 1 : 0][i], acc[i][j], 0, 0, 0);
    }
#undef S4_ISSUE
#undef S4_FRAG
    __syncthreads();                                             // every wave is past its last LDS read: the stages are free
    char* wl = smem + wave * 4096;                               // 4 KiB per wave: [32 rows][128 B]
    const int row0 = m0 + wr * 32, col0 = n0 + wc * 64;
    float4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = col0 + j * 16 + 4 * (lane >> 4);
        bb[j] = (bias && c + 4 <= N) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (HASZ) s4_store_pass<ACT, 0>(acc, Z, ldz, bb, row0, col0, M, N, lane, wl);
    s4_store_pass<ACT, 2>(acc, C + c_lo, ldc, bb, row0, col0, M, N, lane, wl);
    s4_store_pass<ACT, 1>(acc, C, ldc, bb, row0, col0, M, N, lane, wl);
}

extern "C" {

// same contract and requirements as svpc_gemm_p8x3 (gemm_p8x3.hip); preferred when ceil(M/256)·ceil(N/256) leaves most CUs idle
int svpc_gemm_s4x3(const void* A, int lda, int a_lo, const void* B, int ldb, long long b_lo, void* C, int ldc, int c_lo, void* Z, int ldz,
                   int M, int N, int K, const float* bias, int act, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (Z == nullptr) ldz = 0;
    SVPC_REQUIRE(K >= 2 * S4_KS && K % S4_KS == 0 && (N & 7) == 0 && !((lda | a_lo | ldb | ldc | c_lo | ldz) & 7) && a_lo >= K && lda >= a_lo + K &&
                     c_lo >= N && ldc >= c_lo + N && ldb >= K && (b_lo & 7) == 0 &&
                     (unsigned long long)M * (unsigned long long)ldc * 2ull < (1ull << 32) &&
                     ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)Z) | ((uintptr_t)bias)) & 15) == 0 &&
                     (act == ACT_RELU || act == ACT_GELU || (act == ACT_NONE && Z == nullptr)) && (Z == nullptr || ldz >= N),
                 "gemm_s4x3: needs K % 32 == 0, K >= 64, N % 8 == 0, 16-byte aligned split rows (lo plane behind the hi plane), act in {none, relu, gelu}");
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    const int tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128);
    // at most one workgroup per CU anyway (the decoder's N = 768 projections: 198): four stages, three k-tiles in flight
    static int deep_max = -1;
    if (deep_max < 0) { const char* e = getenv("SVPC_S4X3_DEEP_MAX"); deep_max = e ? atoi(e) : 256; }
    const bool deep = tiles_m * tiles_n <= deep_max;
#define S4_GO(ACTV, ZV)                                                                                                           \
    do {                                                                                                                          \
        if (deep) hipLaunchKernelGGL((gemm_s4x3_kernel<ACTV, ZV, 4>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, a_lo, \
                       (const __bf16*)B, ldb, b_lo, (__bf16*)C, ldc, c_lo, (__bf16*)Z, ldz, bias, M, N, K, tiles_m, tiles_n, remap); \
        else hipLaunchKernelGGL((gemm_s4x3_kernel<ACTV, ZV, 2>), dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, lda, a_lo, \
                       (const __bf16*)B, ldb, b_lo, (__bf16*)C, ldc, c_lo, (__bf16*)Z, ldz, bias, M, N, K, tiles_m, tiles_n, remap); \
    } while (0)
    const bool z = Z != nullptr;
    if (act == ACT_GELU) { if (z) S4_GO(ACT_GELU, true); else S4_GO(ACT_GELU, false); }
    else if (act == ACT_RELU) { if (z) S4_GO(ACT_RELU, true); else S4_GO(ACT_RELU, false); }
    else S4_GO(ACT_NONE, false);
#undef S4_GO
    return svpc_check_launch("gemm_s4x3");
}

}  // extern "C"
