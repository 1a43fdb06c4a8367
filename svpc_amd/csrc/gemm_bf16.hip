// Mixed-precision GEMM for the dense contractions of the hot path: bf16 operands on the CDNA4 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate, dense peak ≈2.5 PFLOP/s) with the same fused epilogue and the same
// layout contract as gemm.hip:
//
//   C[M,N] = epi( sum_k A(m,k) · B(n,k) ),  A: a_kc ? [M][lda] : [K][lda],  B: b_kc ? [N][ldb] : [K][ldb]
//
// Operands may sit in HBM as fp32 (converted with v_cvt_pk_bf16_f32 while they are staged) or as bf16.
// Tile 128×128 (or 64×64), BK = 32, 256 threads = 4 waves in 2×2, each wave (BM/2)×(BN/2) as 32×32 MFMA tiles.
// Two LDS images, chosen per operand by where its reduction index lives in memory:
//   k-contiguous operand  → image [rows][32] bf16, 80-byte rows: the 8-element MFMA fragment is ONE conflict-free
//                            ds_read_b128 per lane (rows r and r+16 of a lane group land on disjoint bank quads);
//   k-strided operand     → image [32][cols] bf16 exactly as it lies in memory (coalesced 16-byte global loads, 8-byte LDS
//     (wgrad / dgrad)        stores), rows padded by 64 B; the fragment is TWO ds_read_b64_tr_b16 — the hardware transpose
//                            read of gfx950 — so no transposed copy of an activation or weight is ever made.
// Global → register prefetch of k-tile t+1 runs under the MFMAs of tile t (two LDS buffers, one barrier per k-tile);
// XCD-aware workgroup remap; split-K slabs reduced in a fixed order for the long-K / few-tile products.
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int GBK = 32;            // k-tile depth
constexpr int RS_K = GBK * 2 + 16; // row stride (bytes) of the k-contiguous image
template <int BM> struct ImgM { static constexpr int RS = BM * 2 + 64; };   // row stride (bytes) of the k-strided image

__device__ __forceinline__ bf16x4 cvt4(float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}

// stages one BM × GBK operand tile: global (fp32) → registers → LDS (bf16)
template <int BM, bool KC>
struct Stage {
    static constexpr int NU = (BM * GBK / 4) / 256;   // float4 units per thread: 4 (BM=128) or 2 (BM=64)
    float4 reg[NU];
    const float* ptr[NU];                             // interior-tile fast path: per-unit source pointers

    // interior tiles (no edge in m or k): pointers are set once, each k-tile is NU unguarded 16-byte loads
    __device__ __forceinline__ void init_full(const float* __restrict__ P, int ld, int m0, int k0) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + 256 * i;
            if (KC) ptr[i] = P + (size_t)(m0 + (u >> 3)) * ld + k0 + 4 * (u & 7);
            else { constexpr int UPR = BM / 4; ptr[i] = P + (size_t)(k0 + u / UPR) * ld + m0 + 4 * (u % UPR); }
        }
    }
    __device__ __forceinline__ void load_full(int ld) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            reg[i] = *reinterpret_cast<const float4*>(ptr[i]);
            ptr[i] += KC ? GBK : (size_t)GBK * ld;
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int m0, int k0, int Mdim, int Kend, bool vec_ok) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int row = u >> 3, k = k0 + 4 * (u & 7), m = m0 + row;
                if (m < Mdim) {
                    const float* p = P + (size_t)m * ld + k;
                    if (vec_ok && k + 3 < Kend) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (k < Kend) v.x = p[0];
                        if (k + 1 < Kend) v.y = p[1];
                        if (k + 2 < Kend) v.z = p[2];
                        if (k + 3 < Kend) v.w = p[3];
                    }
                }
            } else {
                constexpr int UPR = BM / 4;               // units per k-row
                const int krow = u / UPR, m = m0 + 4 * (u % UPR), k = k0 + krow;
                if (k < Kend) {
                    const float* p = P + (size_t)k * ld + m;
                    if (vec_ok && m + 3 < Mdim) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (m < Mdim) v.x = p[0];
                        if (m + 1 < Mdim) v.y = p[1];
                        if (m + 2 < Mdim) v.z = p[2];
                        if (m + 3 < Mdim) v.w = p[3];
                    }
                }
            }
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ img) const {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + 256 * i;
            const bf16x4 b = cvt4(reg[i]);
            if (KC) {
                const int row = u >> 3, kq = u & 7;
                *reinterpret_cast<bf16x4*>(img + row * RS_K + kq * 8) = b;
            } else {
                constexpr int UPR = BM / 4;
                const int krow = u / UPR, c4 = u % UPR;
                *reinterpret_cast<bf16x4*>(img + krow * ImgM<BM>::RS + c4 * 8) = b;
            }
        }
    }
};

// MFMA 32x32x16 operand fragment of the 32 rows starting at `row0` for k-step `ks` (16 deep) of the staged tile.
// Lane l = (r = l & 31, h = l >> 5) must hold elements k = 8h .. 8h+7 of row r.
template <int BM, bool KC>
__device__ __forceinline__ bf16x8 fragment(const char* __restrict__ img, int row0, int ks, int lane) {
    if (KC) {
        return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * RS_K + ks * 32 + (lane >> 5) * 16);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
        // 4-row × 16-column block; lane i receives column i with rows 0..3 in its 4 elements.
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g >> 1;
        const char* a = img + (ks * 16 + 8 * h + q) * ImgM<BM>::RS + (row0 + 16 * (g & 1) + 4 * p) * 2;
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * ImgM<BM>::RS));
        union { short s[8]; bf16x8 v; } u;
        u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
        u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
        return u.v;
    }
}

template <int BM, bool KC> struct ImgBytes { static constexpr int value = KC ? BM * RS_K : GBK * ImgM<BM>::RS; };

template <int BM, int BN, bool A_KC, bool B_KC, bool FULL>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        float* __restrict__ C, int ldc, int M, int N, int K, Epi epi,
                                                        int tiles_m, int tiles_n, int splitk, int k_chunk,
                                                        float* __restrict__ slabs, int a_vec, int b_vec, int remap) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int ABYTES = ImgBytes<BM, A_KC>::value, BBYTES = ImgBytes<BN, B_KC>::value;
    __shared__ __attribute__((aligned(16))) char smem[2 * (ABYTES + BBYTES)];
    constexpr int BUF = ABYTES + BBYTES;   // buffer b: A image at b*BUF, B image at b*BUF + ABYTES

    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n * splitk) : (int)blockIdx.x;
    const int ks_id = wg / (tiles_m * tiles_n);
    const int tile = wg - ks_id * (tiles_m * tiles_n);
    int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    if (remap == 2) { tn = tile / tiles_m; tm = tile - tn * tiles_m; }   // m fastest: neighbours share the B panel
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = ks_id * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Stage<BM, A_KC> sa;
    Stage<BN, B_KC> sb;
    const int nk = (k_end - k_begin + GBK - 1) / GBK;
    if (FULL) { sa.init_full(A, lda, m0, k_begin); sb.init_full(B, ldb, n0, k_begin); }
    if (nk > 0) {
        if (FULL) { sa.load_full(lda); sb.load_full(ldb); }
        else {
            sa.load(A, lda, m0, k_begin, M, k_end, a_vec);
            sb.load(B, ldb, n0, k_begin, N, k_end, b_vec);
        }
        sa.store(smem);
        sb.store(smem + ABYTES);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            if (FULL) { sa.load_full(lda); sb.load_full(ldb); }
            else {
                sa.load(A, lda, m0, k_begin + (kt + 1) * GBK, M, k_end, a_vec);
                sb.load(B, ldb, n0, k_begin + (kt + 1) * GBK, N, k_end, b_vec);
            }
        }
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = fragment<BM, A_KC>(smem + cur * BUF, wr * (BM / 2) + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = fragment<BN, B_KC>(smem + cur * BUF + ABYTES, wc * (BN / 2) + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(smem + (cur ^ 1) * BUF);
            sb.store(smem + (cur ^ 1) * BUF + ABYTES);
        }
        __syncthreads();
    }

    const u64 seed = (epi.p_drop > 0.f && splitk == 1) ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row < M && col < N) {
                    if (splitk == 1) epilogue_store(acc[i][j][e], row, col, C, ldc, epi, seed, inv_keep);
                    else slabs[((size_t)ks_id * M + row) * N + col] = acc[i][j][e];
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ slabs, int splitk, float* __restrict__ C,
                                                                 int ldc, int M, int N, Epi epi) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i - (size_t)row * N);
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += slabs[(size_t)k * M * N + i];
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    epilogue_store(s, row, col, C, ldc, epi, seed, inv_keep);
}

template <int BM, int BN>
static void launch_gemm_bf16(bool a_kc, bool b_kc, dim3 grid, hipStream_t s, const float* A, int lda, const float* B, int ldb,
                             float* C, int ldc, int M, int N, int K, Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk,
                             float* slabs, int a_vec, int b_vec) {
    static int remap = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    // interior-only problems (every tile full in m, n and k, 16-byte aligned rows) take the unguarded staging path
    const bool full = (M % BM == 0) && (N % BN == 0) && (K % k_chunk == 0) && (k_chunk % GBK == 0) && a_vec && b_vec;
#define SVPC_GEMM_LAUNCH(AK, BKC)                                                                                                  \
    do {                                                                                                                           \
        if (full)                                                                                                                  \
            hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, \
                               epi, tiles_m, tiles_n, splitk, k_chunk, slabs, a_vec, b_vec, remap);                               \
        else                                                                                                                       \
            hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, \
                               epi, tiles_m, tiles_n, splitk, k_chunk, slabs, a_vec, b_vec, remap);                               \
    } while (0)
    if (a_kc && b_kc) SVPC_GEMM_LAUNCH(true, true);
    else if (a_kc && !b_kc) SVPC_GEMM_LAUNCH(true, false);
    else if (!a_kc && b_kc) SVPC_GEMM_LAUNCH(false, true);
    else SVPC_GEMM_LAUNCH(false, false);
#undef SVPC_GEMM_LAUNCH
}

extern "C" {

// Same contract as svpc_gemm_f32; operands are rounded to bf16 (RNE) on their way into LDS, products accumulate in fp32.
int svpc_gemm_bf16(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N,
                   int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                   float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "gemm: dropout needs a seed pointer");
    Epi epi{bias, act, p_drop, site, seed, accumulate, Z};
    const bool big = (M >= 96 && N >= 96);
    const int BMN = big ? 128 : 64;
    const int tiles_m = ceil_div(M, BMN), tiles_n = ceil_div(N, BMN);
    const int tiles = tiles_m * tiles_n;
    int splitk = 1;
    if (K >= 512 && tiles < 256) {
        splitk = ceil_div(512, tiles);
        const int max_by_k = K / 256;
        if (splitk > max_by_k) splitk = max_by_k;
        if (splitk > 64) splitk = 64;
        while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > workspace_bytes) --splitk;
        if (splitk < 1) splitk = 1;
    }
    int k_chunk = ceil_div(ceil_div(K, splitk), GBK) * GBK;
    splitk = ceil_div(K, k_chunk);
    if (K == 0) { splitk = 1; k_chunk = GBK; }
    const int a_vec = (lda % 4 == 0) && ((((uintptr_t)A) & 15) == 0);
    const int b_vec = (ldb % 4 == 0) && ((((uintptr_t)B) & 15) == 0);
    dim3 grid(tiles * splitk);
    if (big) launch_gemm_bf16<128, 128>(a_kc, b_kc, grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk,
                                        k_chunk, workspace, a_vec, b_vec);
    else launch_gemm_bf16<64, 64>(a_kc, b_kc, grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk,
                                  workspace, a_vec, b_vec);
    int rc = svpc_check_launch("gemm_bf16");
    if (rc) return rc;
    if (splitk > 1) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, workspace, splitk, C, ldc,
                           M, N, epi);
        rc = svpc_check_launch("gemm_bf16 splitk reduce");
    }
    return rc;
}

}  // extern "C"
