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

#ifndef SVPC_GBK
#define SVPC_GBK 32
#endif
constexpr int GBK = SVPC_GBK;      // k-tile depth
constexpr int RS_K = GBK * 2 + 16; // row stride (bytes) of the k-contiguous image
template <int BM> struct ImgM { static constexpr int RS = BM * 2 + 64; };   // row stride (bytes) of the k-strided image

__device__ __forceinline__ bf16x4 cvt4(float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}

// stages one BM × GBK operand tile: global (fp32) → registers → LDS (bf16)
template <int BM, bool KC, int NT, typename T>
struct Stage {
    static constexpr bool F32 = sizeof(T) == 4;
    static constexpr int EPU = F32 ? 4 : 8;           // elements per 16-byte unit
    static constexpr int NU = (BM * GBK / EPU) / NT;  // 16-byte units per thread
    uint4 reg[NU];                                    // raw 16 bytes (4 fp32 or 8 bf16)
    const T* ptr[NU];                                 // interior-tile fast path: per-unit source pointers

    // interior tiles (no edge in m or k): pointers are set once, each k-tile is NU unguarded 16-byte loads
    __device__ __forceinline__ void init_full(const T* __restrict__ P, int ld, int m0, int k0) {
        constexpr int UK = GBK / EPU;                 // units per row of the k-contiguous image
        constexpr int UPR = BM / EPU;                 // units per k-row of the k-strided image
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + NT * i;
            if (KC) ptr[i] = P + (size_t)(m0 + u / UK) * ld + k0 + EPU * (u % UK);
            else ptr[i] = P + (size_t)(k0 + u / UPR) * ld + m0 + EPU * (u % UPR);
        }
    }
    __device__ __forceinline__ void load_full(int ld) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            reg[i] = *reinterpret_cast<const uint4*>(ptr[i]);
            ptr[i] += KC ? GBK : (size_t)GBK * ld;
        }
    }

    // general (edge-guarded) path: fp32 sources only
    __device__ __forceinline__ void load(const T* __restrict__ Pt, int ld, int m0, int k0, int Mdim, int Kend, bool vec_ok) {
        const float* __restrict__ P = reinterpret_cast<const float*>(Pt);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + NT * i;
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
            reg[i] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ img) const {
        constexpr int UK = GBK / EPU, UPR = BM / EPU;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = threadIdx.x + NT * i;
            char* dst = KC ? img + (u / UK) * RS_K + (u % UK) * (EPU * 2) : img + (u / UPR) * ImgM<BM>::RS + (u % UPR) * (EPU * 2);
            if (F32) {
                bf16x4 b;
                b[0] = (__bf16)__uint_as_float(reg[i].x); b[1] = (__bf16)__uint_as_float(reg[i].y);
                b[2] = (__bf16)__uint_as_float(reg[i].z); b[3] = (__bf16)__uint_as_float(reg[i].w);
                *reinterpret_cast<bf16x4*>(dst) = b;
            } else {
                *reinterpret_cast<uint4*>(dst) = reg[i];
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

// NWN = waves along n (2 → 4 waves of 64×64, 4 → 8 waves of 64×32 for BM = BN = 128): more resident waves per SIMD hide
// the staging-load latency that bounds this one-tile-deep pipeline
template <int BM, int BN, bool A_KC, bool B_KC, bool FULL, int NWN, typename TA, typename TB, typename TC>
__global__ __launch_bounds__(128 * NWN) void gemm_bf16_kernel(const TA* __restrict__ A, int lda, const TB* __restrict__ B, int ldb,
                                                        TC* __restrict__ C, int ldc, int M, int N, int K, Epi epi,
                                                        int tiles_m, int tiles_n, int splitk, int k_chunk,
                                                        float* __restrict__ slabs, int a_vec, int b_vec, int remap) {
    constexpr int NT = 128 * NWN;
    constexpr int TM = BM / 64, TN = BN / (32 * NWN);
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
    const int wr = wave / NWN, wc = wave % NWN;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Stage<BM, A_KC, NT, TA> sa;
    Stage<BN, B_KC, NT, TB> sb;
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
            for (int j = 0; j < TN; ++j) b[j] = fragment<BN, B_KC>(smem + cur * BUF + ABYTES, wc * (BN / NWN) + j * 32, ks, lane);
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
    if (sizeof(TC) == 2 && FULL && splitk == 1 && epi.p_drop <= 0.f && !epi.accumulate) {
        // bf16 output, interior tile: sub-dword stores are slow, so neighbouring lanes swap one value and every lane stores
        // two adjacent columns of one row as a single dword (even lanes take rows e, odd lanes rows e+1)
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wc * (BN / NWN) + j * 32 + l31;
                const float bias = epi.bias ? epi.bias[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const float z0 = acc[i][j][e] + bias, z1 = acc[i][j][e + 1] + bias;
                    const float y0 = apply_act(z0, epi.act), y1 = apply_act(z1, epi.act);
                    // even lane keeps (row e: own y0, partner y0); odd lane keeps (row e+1: partner y1, own y1)
                    const float py = __shfl_xor(odd ? y0 : y1, 1, 64);
                    const float pz = __shfl_xor(odd ? z0 : z1, 1, 64);
                    const int row = m0 + wr * (BM / 2) + i * 32 + ((e + (odd ? 1 : 0)) & 3) + 8 * (e >> 2) + 4 * lhi;
                    const size_t o = (size_t)row * ldc + (col & ~1);
                    union { __bf16 h[2]; uint32_t u; } pk;
                    pk.h[0] = (__bf16)(odd ? py : y0); pk.h[1] = (__bf16)(odd ? y1 : py);
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<__bf16*>(C) + o) = pk.u;
                    if (epi.Z) {
                        pk.h[0] = (__bf16)(odd ? pz : z0); pk.h[1] = (__bf16)(odd ? z1 : pz);
                        *reinterpret_cast<uint32_t*>(reinterpret_cast<__bf16*>(epi.Z) + o) = pk.u;
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * (BN / NWN) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row < M && col < N) {
                    if (splitk == 1) epilogue_store_t<TC>(acc[i][j][e], row, col, C, ldc, epi, seed, inv_keep);
                    else slabs[((size_t)ks_id * M + row) * N + col] = acc[i][j][e];
                }
            }
        }
}

template <int BM, int BN, int NWN, bool AK, bool BKC, bool FULL, typename TA, typename TB, typename TC>
static void launch_one(dim3 grid, hipStream_t s, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                       Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk, float* slabs, int a_vec, int b_vec, int remap) {
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, FULL, NWN, TA, TB, TC>), grid, dim3(128 * NWN), 0, s, (const TA*)A, lda,
                       (const TB*)B, ldb, (TC*)C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk, slabs, a_vec, b_vec, remap);
}

// fp32 operands in HBM: every layout, every shape (edge-guarded path when a tile is ragged)
template <int BM, int BN, int NWN>
static void launch_f32(bool a_kc, bool b_kc, bool full, dim3 grid, hipStream_t s, const void* A, int lda, const void* B, int ldb, void* C,
                       int ldc, int M, int N, int K, Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk, float* slabs, int a_vec,
                       int b_vec, int remap) {
#define SVPC_L(AK, BKC, FU) launch_one<BM, BN, NWN, AK, BKC, FU, float, float, float>(grid, s, A, lda, B, ldb, C, ldc, M, N, K, epi, \
                                                                                     tiles_m, tiles_n, splitk, k_chunk, slabs, a_vec, b_vec, remap)
    if (full) {
        if (a_kc && b_kc) SVPC_L(true, true, true); else if (a_kc) SVPC_L(true, false, true);
        else if (b_kc) SVPC_L(false, true, true); else SVPC_L(false, false, true);
    } else {
        if (a_kc && b_kc) SVPC_L(true, true, false); else if (a_kc) SVPC_L(true, false, false);
        else if (b_kc) SVPC_L(false, true, false); else SVPC_L(false, false, false);
    }
#undef SVPC_L
}

extern "C" {

// dtype codes: 0 = fp32, 1 = bf16.  Supported operand/output combinations:
//   (A f32, B f32, C f32)   any shape, any layout
//   (A bf16, B f32, C bf16) forward / dgrad of a bf16 activation stream (layouts NT, NN), interior-only shapes
//   (A bf16, B bf16, C f32) wgrad of a bf16 activation stream (layout TN), interior-only shapes
int svpc_gemm_mx(const void* A, int a_dt, int lda, int a_kc, const void* B, int b_dt, int ldb, int b_kc, void* C, int c_dt, int ldc,
                 void* Z, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                 float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "gemm: dropout needs a seed pointer");
    Epi epi{bias, act, p_drop, site, seed, accumulate, (float*)Z};
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
    if (a_dt || b_dt || c_dt) {
        // the bf16-storage variants only exist in the unguarded form: every K slice must be whole (K % 32 == 0 is required)
        while (splitk > 1 && K % k_chunk != 0) { --splitk; k_chunk = ceil_div(ceil_div(K, splitk), GBK) * GBK; }
    }
    splitk = ceil_div(K, k_chunk);
    if (K == 0) { splitk = 1; k_chunk = GBK; }
    const int a_el = a_dt ? 8 : 4, b_el = b_dt ? 8 : 4;   // elements per 16 bytes
    const int a_vec = (lda % a_el == 0) && ((((uintptr_t)A) & 15) == 0);
    const int b_vec = (ldb % b_el == 0) && ((((uintptr_t)B) & 15) == 0);
    const bool full = (M % BMN == 0) && (N % BMN == 0) && (K % k_chunk == 0) && (k_chunk % GBK == 0) && a_vec && b_vec;
    static int remap = -1, nwn_env = -1;
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    if (nwn_env < 0) { const char* e = getenv("SVPC_GEMM_NWN"); nwn_env = e ? atoi(e) : 0; }
    // 8 waves (64×32 per wave) hide the staging latency better when A is k-contiguous and the grid is not huge;
    // the transposed-read (wgrad) and very wide problems run better with 4 waves of 64×64
    int nwn = (a_kc && tiles <= 1200) ? 4 : 2;
    if (nwn_env == 2 || nwn_env == 4) nwn = nwn_env;
    dim3 grid(tiles * splitk);
#define SVPC_ARGS grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk, workspace, a_vec, b_vec, remap
    if (a_dt == 0 && b_dt == 0 && c_dt == 0) {
        if (big && nwn == 4) launch_f32<128, 128, 4>(a_kc, b_kc, full, SVPC_ARGS);
        else if (big) launch_f32<128, 128, 2>(a_kc, b_kc, full, SVPC_ARGS);
        else launch_f32<64, 64, 2>(a_kc, b_kc, full, SVPC_ARGS);
    } else {
        SVPC_REQUIRE(big && full, "gemm_mx: bf16 operands need interior-only shapes (multiples of 128 × 128 × 32, 16-byte rows)");
        // bf16-operand variants always use the 8-wave form (one 16-byte staging unit per thread and operand)
        if (a_dt == 1 && b_dt == 0 && c_dt == 1 && a_kc && b_kc) {
            launch_one<128, 128, 4, true, true, true, __bf16, float, __bf16>(SVPC_ARGS);
        } else if (a_dt == 1 && b_dt == 0 && c_dt == 1 && a_kc && !b_kc) {
            launch_one<128, 128, 4, true, false, true, __bf16, float, __bf16>(SVPC_ARGS);
        } else if (a_dt == 1 && b_dt == 1 && c_dt == 0 && !a_kc && !b_kc) {
            launch_one<128, 128, 4, false, false, true, __bf16, __bf16, float>(SVPC_ARGS);
        } else {
            svpc_set_error("gemm_mx: unsupported dtype/layout combination");
            return -1;
        }
    }
#undef SVPC_ARGS
    int rc = svpc_check_launch("gemm_mx");
    if (rc) return rc;
    if (splitk > 1) {
        if (c_dt == 0) launch_splitk_reduce<float>(workspace, splitk, (float*)C, ldc, M, N, epi, stream);
        else launch_splitk_reduce<__bf16>(workspace, splitk, (__bf16*)C, ldc, M, N, epi, stream);
        rc = svpc_check_launch("gemm_mx splitk reduce");
    }
    return rc;
}

// fp32-in / fp32-out form (same contract as svpc_gemm_f32): operands are rounded to bf16 (RNE) on their way into LDS
int svpc_gemm_bf16(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N,
                   int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                   float* workspace, size_t workspace_bytes, hipStream_t stream) {
    return svpc_gemm_mx(A, 0, lda, a_kc, B, 0, ldb, b_kc, C, 0, ldc, Z, M, N, K, bias, act, p_drop, site, seed, accumulate, workspace,
                        workspace_bytes, stream);
}

}  // extern "C"
