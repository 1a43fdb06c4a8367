// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD) with fused
// epilogue (bias, ReLU / exact-erf GELU / sigmoid, dropout, accumulate) — the dense contractions of the hot path:
// video embedding F→D (model.py:551), Q/K/V and output projections (:195-197, :230), FFN (:259, :281), LM head
// (:706, :738), simulator / LSTM projections (:755-770, :865) and all of their backward products.
//
//   C[M,N] = epi( sum_k A(m,k) · B(n,k) )
//   A: a_kc ? [M][lda] (k contiguous) : [K][lda] (m contiguous)      B: b_kc ? [N][ldb] : [K][ldb]
//   forward  y = x Wᵀ        : A=x  (kc)  B=W  (kc)
//   dgrad    dx = dy W       : A=dy (kc)  B=W  (n = in-features contiguous → b_kc = 0)
//   wgrad    dW = dyᵀ x      : A=dy (m=out contiguous, a_kc = 0)  B=x (n=in contiguous, b_kc = 0)
//
// Tile BM×BN (128×128 or 64×64), BK = 16, 256 threads = 4 waves in 2×2, each wave (BM/2)×(BN/2) as 32×32 MFMA
// tiles.  LDS tiles are k-major ([BK][BM+4]) so a fragment read is one conflict-free ds_read_b32 per lane;
// global→register prefetch of tile t+1 overlaps the MFMAs of tile t (two LDS buffers, one barrier per k-tile).
// Workgroup ids are remapped so that the tiles sharing an A row-panel run on one XCD (private L2).
// Long-K / few-tile problems (the wgrad products) are split along K into fp32 slabs reduced in a fixed order
// (deterministic, no atomics).  Bound: fp32 MFMA peak 157 TFLOP/s.
#include "gemm_common.h"

constexpr int BK = 16;

template <int BM, bool KC>
struct TileLoader {
    // a BM × BK operand tile → registers → LDS [BK][BM + 4]
    static constexpr int LD = BM + 4;
    static constexpr int NV = (BM * BK / 4) / 256;  // float4 per thread (2 for 128, 1 for 64)
    float4 reg[NV];

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int m0, int k0, int Mdim, int Kdim, bool vec_ok) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                const int row = (t >> 2) + i * 64;
                const int m = m0 + row, k = k0 + 4 * (t & 3);
                if (m < Mdim) {
                    const float* p = P + (size_t)m * ld + k;
                    if (vec_ok && k + 3 < Kdim) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (k < Kdim) v.x = p[0];
                        if (k + 1 < Kdim) v.y = p[1];
                        if (k + 2 < Kdim) v.z = p[2];
                        if (k + 3 < Kdim) v.w = p[3];
                    }
                }
            } else {
                constexpr int V4 = BM / 4;            // float4 per k-row
                const int krow = (t / V4) + i * (256 / V4);
                const int k = k0 + krow, m = m0 + 4 * (t % V4);
                if (k < Kdim) {
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
    __device__ __forceinline__ void store(float* __restrict__ S) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (KC) {
                const int row = (t >> 2) + i * 64;
                const int kk = 4 * (t & 3);
                S[(kk + 0) * LD + row] = reg[i].x;
                S[(kk + 1) * LD + row] = reg[i].y;
                S[(kk + 2) * LD + row] = reg[i].z;
                S[(kk + 3) * LD + row] = reg[i].w;
            } else {
                constexpr int V4 = BM / 4;
                const int krow = (t / V4) + i * (256 / V4);
                *reinterpret_cast<float4*>(&S[krow * LD + 4 * (t % V4)]) = reg[i];
            }
        }
    }
};

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N, int K, Epi epi,
                                                       int tiles_m, int tiles_n, int splitk, int k_chunk,
                                                       float* __restrict__ slabs, int a_vec, int b_vec) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = BM / 64, TN = BN / 64;   // 32×32 MFMA tiles per wave
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    // XCD-aware remap of the linear workgroup id (bijective for any grid size)
    const int nwg = tiles_m * tiles_n * splitk;
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int ks = wg / (tiles_m * tiles_n);
    const int tile = wg - ks * (tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = ks * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    TileLoader<BM, A_KC> la;
    TileLoader<BN, B_KC> lb;
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        la.load(A, lda, m0, k_begin, M, k_end, a_vec);
        lb.load(B, ldb, n0, k_begin, N, k_end, b_vec);
        la.store(As[0]);
        lb.store(Bs[0]);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(A, lda, m0, k_begin + (kt + 1) * BK, M, k_end, a_vec);
            lb.load(B, ldb, n0, k_begin + (kt + 1) * BK, N, k_end, b_vec);
        }
        const float* as = As[cur] + wr * (BM / 2) + l31;
        const float* bs = Bs[cur] + wc * (BN / 2) + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(kk + lhi) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(kk + lhi) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    // C/D layout of the 32×32 MFMA: col = lane & 31, row = (reg & 3) + 8·(reg >> 2) + 4·(lane >> 5)
    const u64 seed = (epi.p_drop > 0.f && splitk == 1) ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
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
                    else slabs[((size_t)ks * M + row) * N + col] = acc[i][j][e];
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splitk, float* __restrict__ C,
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
static void launch_gemm(bool a_kc, bool b_kc, dim3 grid, hipStream_t s, const float* A, int lda, const float* B, int ldb, float* C,
                        int ldc, int M, int N, int K, Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk, float* slabs,
                        int a_vec, int b_vec) {
#define SVPC_GEMM_LAUNCH(AK, BKC)                                                                                          \
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BKC>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, epi, \
                       tiles_m, tiles_n, splitk, k_chunk, slabs, a_vec, b_vec)
    if (a_kc && b_kc) SVPC_GEMM_LAUNCH(true, true);
    else if (a_kc && !b_kc) SVPC_GEMM_LAUNCH(true, false);
    else if (!a_kc && b_kc) SVPC_GEMM_LAUNCH(false, true);
    else SVPC_GEMM_LAUNCH(false, false);
#undef SVPC_GEMM_LAUNCH
}

extern "C" {

// Z (optional) receives the pre-activation values.  workspace is used only when the problem is split along K.
int svpc_gemm_f32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N,
                  int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                  float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "gemm: dropout needs a seed pointer");
    Epi epi{bias, act, p_drop, site, seed, accumulate, Z};
    // 128×128 tiles whenever both output dims can fill one; long-K problems with few tiles are split along K so
    // that ≥ ~2 workgroups per CU are in flight (wgrad: 36 tiles × K = 19,200 → 15 slabs).
    const bool big = (M >= 96 && N >= 96);
    const int BMN = big ? 128 : 64;
    const int tiles_m = ceil_div(M, BMN), tiles_n = ceil_div(N, BMN);
    const int tiles = tiles_m * tiles_n;
    int splitk = 1;
    if (K >= 512 && tiles < 256) {
        splitk = ceil_div(512, tiles);
        const int max_by_k = K / 128;
        if (splitk > max_by_k) splitk = max_by_k;
        if (splitk > 64) splitk = 64;
        while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > workspace_bytes) --splitk;
        if (splitk < 1) splitk = 1;
    }
    int k_chunk = ceil_div(ceil_div(K, splitk), BK) * BK;
    splitk = ceil_div(K, k_chunk);
    if (K == 0) { splitk = 1; k_chunk = BK; }
    const int a_vec = (lda % 4 == 0) && ((((uintptr_t)A) & 15) == 0);
    const int b_vec = (ldb % 4 == 0) && ((((uintptr_t)B) & 15) == 0);
    dim3 grid(tiles * splitk);
    if (big) launch_gemm<128, 128>(a_kc, b_kc, grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk,
                                   workspace, a_vec, b_vec);
    else launch_gemm<64, 64>(a_kc, b_kc, grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk,
                             workspace, a_vec, b_vec);
    int rc = svpc_check_launch("gemm_f32");
    if (rc) return rc;
    if (splitk > 1) {
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, workspace, splitk, C, ldc,
                           M, N, epi);
        rc = svpc_check_launch("gemm splitk reduce");
    }
    return rc;
}

}  // extern "C"
