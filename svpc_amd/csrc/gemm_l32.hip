// fp32-operand GEMM on the bf16 matrix cores with DIRECT-TO-LDS staging, for the latency-bound half of the train step
// (decoder / text rows, step-level rows, simulator and LSTM projections, their dgrad and wgrad): these launches have
// 12–800 tiles and 6–130 k-tiles, so what bounds them is the memory round trip per k-tile, not bandwidth or MFMA rate.
//
//   C[M,N] = epi( sum_k A(m,k) · B(n,k) ),  A: a_kc ? [M][lda] : [K][lda],  B: b_kc ? [N][ldb] : [K][ldb]   (fp32 in HBM)
//
// The fp32 tiles go global → LDS with global_load_lds_dwordx4 into a ring of NS stages, NS-1 k-tiles in flight behind a counted
// vmcnt (no staging registers); they are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when a wave builds its MFMA fragments, which
// is the same rounding the register-staged kernel (gemm_bf16.hip) applies on its way into LDS — results are bit-identical.
// Edges: rows past M / N are clamped to the last row (their products are never stored); K must be a multiple of 32.
//
// LDS images of one operand tile (R rows × 32 k, fp32):
//   k-contiguous: [R][128 B], 16-byte chunk c of row r stored at c ^ ((r>>1)&7)   → 2 × ds_read_b128 per fragment, conflict-free
//   k-strided:    [32 k-rows][R·4 B], linear                                     → 8 × ds_read_b32 per fragment, conflict-free
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const void __attribute__((address_space(1))) * l32_gptr;
typedef void __attribute__((address_space(3))) * l32_lptr;

constexpr int L32_BK = 32;
__device__ __attribute__((aligned(16))) const float l32_zeros[4] = {0.f, 0.f, 0.f, 0.f};     // source of k-rows past K (wgrad tails)

template <int N> __device__ __forceinline__ void l32_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Per-lane source pointer of wave-instruction `j` (1 KiB) of an operand tile whose first row is m0 and first k is k0.
template <int R, bool KC>
__device__ __forceinline__ const float* l32_src(const float* P, int ld, int m0, int rows, int k0, int j, int lane) {
    if (KC) {
        const int row = 8 * j + (lane >> 3), pos = lane & 7;
        const int c = pos ^ ((row >> 1) & 7);
        const int gr = min(m0 + row, rows - 1);
        return P + (size_t)gr * ld + k0 + 4 * c;
    } else {
        constexpr int CPR = R / 4;                      // 16-byte chunks per k-row
        const int krow = j * (64 / CPR) + lane / CPR, c = lane % CPR;
        const int gm = min(m0 + 4 * c, rows - 4);       // rows % 4 == 0 (checked on the host)
        return P + (size_t)(k0 + krow) * ld + gm;
    }
}

// MFMA 32x32x16 operand fragment: row (row0 + lane&31), k = 16·ks + 8·(lane>>5) + 0..7, rounded to bf16
template <int R, bool KC>
__device__ __forceinline__ bf16x8 l32_fragment(const char* __restrict__ img, int row0, int ks, int lane) {
    float f[8];
    if (KC) {
        const int row = row0 + (lane & 31), sw = (row >> 1) & 7;
        const int c0 = 4 * ks + 2 * (lane >> 5);
        const float4 lo = *reinterpret_cast<const float4*>(img + row * 128 + ((c0 ^ sw) << 4));
        const float4 hi = *reinterpret_cast<const float4*>(img + row * 128 + (((c0 + 1) ^ sw) << 4));
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    } else {
        const float* p = reinterpret_cast<const float*>(img) + (16 * ks + 8 * (lane >> 5)) * R + row0 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p[j * R];
    }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
    return v;
}
// bf16x3 arithmetic (the ≤1e-4-parity throughput mode, see gemm_p8x3.hip): an fp32 operand value is split into hi = bf16(v) and
// lo = bf16(v - hi) when the fragments are built, and a product becomes three MFMAs  lo·hi + hi·lo + hi·hi  (fp32 accumulate; the
// dropped lo·lo term is ≤ 2⁻¹⁶ relative).  These launches wait on memory round trips, not on the matrix pipe.
__device__ __forceinline__ void l32_split8(const float* f, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        hi[j] = h;
        lo[j] = (__bf16)(f[j] - (float)h);
    }
}
template <int R, bool KC>
__device__ __forceinline__ void l32_fragment2(const char* __restrict__ img, int row0, int ks, int lane, bf16x8& hi, bf16x8& lo) {
    float f[8];
    if (KC) {
        const int row = row0 + (lane & 31), sw = (row >> 1) & 7;
        const int c0 = 4 * ks + 2 * (lane >> 5);
        const float4 a = *reinterpret_cast<const float4*>(img + row * 128 + ((c0 ^ sw) << 4));
        const float4 b = *reinterpret_cast<const float4*>(img + row * 128 + (((c0 + 1) ^ sw) << 4));
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const float* p = reinterpret_cast<const float*>(img) + (16 * ks + 8 * (lane >> 5)) * R + row0 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p[j * R];
    }
    l32_split8(f, hi, lo);
}
#define L32_MFMA3(ah, al, bh, bl, acc)                                           \
    do {                                                                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);     \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);     \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);     \
    } while (0)


template <int BM, int BN, bool A_KC, bool B_KC, int NS, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_l32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_m,
                                                       int tiles_n, int splitk, int k_chunk, float* __restrict__ slabs, int remap) {
    extern __shared__ __attribute__((aligned(1024))) char l32_smem[];
    constexpr int A_BYTES = BM * L32_BK * 4, B_BYTES = BN * L32_BK * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = BM / 32, B_PER_WAVE = BN / 32;          // wave-instructions per wave and k-tile (4 waves)
    constexpr int P = A_PER_WAVE + B_PER_WAVE;
    constexpr int TM = BM / 64, TN = BN / 64;                          // 32×32 MFMA tiles per wave (waves 2 × 2)
    const int wg = remap ? xcd_remap(blockIdx.x, tiles_m * tiles_n * splitk) : (int)blockIdx.x;
    const int ks_id = wg / (tiles_m * tiles_n);
    const int tile = wg - ks_id * (tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = ks_id * k_chunk;
    const int k_end = min(K, k_begin + k_chunk);
    const int nk = (k_end - k_begin + L32_BK - 1) / L32_BK;
    // K % 32 != 0 (the 300-wide word vectors; both operands k-contiguous, K % 4 == 0 — checked on the host): the 16-byte chunks of
    // the last k-tile that lie past K are sourced from a block of zeros
    const int k_tail = (k_end - k_begin) - (nk - 1) * L32_BK;      // valid k of the last tile (32 when there is no tail)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const float* ga[A_PER_WAVE];
    const float* gb[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) ga[i] = l32_src<BM, A_KC>(A, lda, m0, M, k_begin, wave * A_PER_WAVE + i, lane);
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) gb[i] = l32_src<BN, B_KC>(B, ldb, n0, N, k_begin, wave * B_PER_WAVE + i, lane);
    const size_t stepA = A_KC ? (size_t)L32_BK : (size_t)L32_BK * lda;
    const size_t stepB = B_KC ? (size_t)L32_BK : (size_t)L32_BK * ldb;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // (k-contiguous images) the k offset of this lane's chunk inside a k-tile: 4·(slot ^ swizzle(row)), piece j ↔ rows 8j … 8j+7
#define L32_KOFS(j) (4 * ((lane & 7) ^ (((8 * (j) + (lane >> 3)) >> 1) & 7)))
#define L32_ISSUE(t)                                                                                                              \
    do {                                                                                                                          \
        char* st = l32_smem + ((t) % NS) * STAGE;                                                                                 \
        const bool tail_ = A_KC && B_KC && k_tail < L32_BK && (t) == nk - 1;                                                      \
        _Pragma("unroll") for (int i = 0; i < A_PER_WAVE; ++i) {                                                                  \
            const float* s_ = (tail_ && L32_KOFS(wave * A_PER_WAVE + i) >= k_tail) ? l32_zeros : ga[i];                          \
            __builtin_amdgcn_global_load_lds((l32_gptr)s_, (l32_lptr)(st + (wave * A_PER_WAVE + i) * 1024), 16, 0, 0);           \
            ga[i] += stepA;                                                                                                       \
        }                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < B_PER_WAVE; ++i) {                                                                  \
            const float* s_ = (tail_ && L32_KOFS(wave * B_PER_WAVE + i) >= k_tail) ? l32_zeros : gb[i];                          \
            __builtin_amdgcn_global_load_lds((l32_gptr)s_, (l32_lptr)(st + A_BYTES + (wave * B_PER_WAVE + i) * 1024), 16, 0, 0);  \
            gb[i] += stepB;                                                                                                       \
        }                                                                                                                         \
    } while (0)

    for (int t = 0; t < NS - 1 && t < nk; ++t) L32_ISSUE(t);

    for (int t = 0; t < nk; ++t) {
        // tile t has landed once at most P·(tiles issued after t) of this wave's loads are still outstanding
        const int after = min(NS - 2, nk - 1 - t);
        switch (after) {
            case 0: l32_wait_vmcnt<0>(); break;
            case 1: l32_wait_vmcnt<P>(); break;
            case 2: l32_wait_vmcnt<2 * P>(); break;
            case 3: l32_wait_vmcnt<(NS > 4 ? 3 * P : 0)>(); break;
            case 4: l32_wait_vmcnt<(NS > 5 ? 4 * P : 0)>(); break;
            case 5: l32_wait_vmcnt<(NS > 6 ? 5 * P : 0)>(); break;
            default: l32_wait_vmcnt<(NS > 7 ? 6 * P : 0)>(); break;
        }
        __builtin_amdgcn_s_barrier();               // every wave's part of tile t is in LDS; every wave is done with tile t-1
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) L32_ISSUE(t + NS - 1); // refills the stage tile t-1 used
        const char* sa = l32_smem + (t % NS) * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < L32_BK / 16; ++ks) {
            if constexpr (X3) {
                bf16x8 bh[TN], bl[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) l32_fragment2<BN, B_KC>(sb, wc * (BN / 2) + j * 32, ks, lane, bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    bf16x8 ah, al;
                    l32_fragment2<BM, A_KC>(sa, wr * (BM / 2) + i * 32, ks, lane, ah, al);
#pragma unroll
                    for (int j = 0; j < TN; ++j) L32_MFMA3(ah, al, bh[j], bl[j], acc[i][j]);
                }
            } else {
                bf16x8 bf[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = l32_fragment<BN, B_KC>(sb, wc * (BN / 2) + j * 32, ks, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bf16x8 af = l32_fragment<BM, A_KC>(sa, wr * (BM / 2) + i * 32, ks, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
#undef L32_ISSUE
#undef L32_KOFS

    const u64 seed = (epi.p_drop > 0.f && splitk == 1) ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * (BN / 2) + j * 32 + l31;
            if (splitk == 1) {                            // (wave-uniform) the 16 elements of a lane with every epilogue load in flight at once
                float vs[16]; int rs[16], cs[16]; bool oks[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                    vs[e] = acc[i][j][e]; oks[e] = row < M && col < N; rs[e] = min(row, M - 1); cs[e] = min(col, N - 1);
                }
                epilogue_store_n<16>(vs, rs, cs, oks, C, ldc, epi, seed, inv_keep);
                continue;
            }
            if (col >= N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row >= M) continue;
                slabs[((size_t)ks_id * M + row) * N + col] = acc[i][j][e];
            }
        }
}

// ---- wave-split-K form for grids of a few dozen tiles: ONE 64×64 tile per workgroup, but each of the 4 waves owns every 4th
// k-tile of it — its own 2-stage LDS ring (32 KiB), its own loads, no workgroup barrier inside the k-loop — and computes the whole
// 64×64 tile (2×2 MFMA tiles) over its k-tiles; the four partial tiles are summed through LDS in wave order (deterministic) by all
// 256 threads, which then run the epilogue with coalesced rows.  A 24-deep k-loop becomes 6 steps per wave with four times the
// bytes in flight per CU — these launches are bound by the memory round trip per k-tile, nothing else.
// `asum` (optional): += Σ_k A(m, k) for the tile's 64 rows m — the bias gradient of a wgrad (A = dz, k-strided) — taken from the
// fp32 LDS image (not the bf16-rounded fragments) by the workgroups of the first tile column.
// PRE (the grouped weight gradients: a plain accumulate epilogue): the 16 values of C a thread will add to are requested BEFORE the
// operand tiles, so the read half of the read-modify-write travels with the operands instead of after the reduction.
template <bool A_KC, bool B_KC, int NS = 2, bool X3 = false, bool PRE = false>
__device__ __forceinline__ void l32w_tile(char* l32_smem, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          float* __restrict__ C, int ldc, int M, int N, int K, const Epi& epi, int tm, int tn,
                                          float* __restrict__ asum) {
    constexpr int T = 64, OP = T * L32_BK * 4, STAGE = 2 * OP, PIECES = OP / 1024;      // 8 pieces of 1 KiB per operand tile; NS stages per wave
    const int m0 = tm * T, n0 = tn * T;
    float cpre[PRE ? T * T / 256 : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int u = 0; u < T * T / 256; ++u) {
            const int idx = threadIdx.x + 256 * u;
            const int row = m0 + (idx >> 6), col = n0 + (idx & 63);
            cpre[u] = (row < M && col < N) ? C[(size_t)row * ldc + col] : 0.f;
        }
    }
    const bool want_asum = !A_KC && asum != nullptr && tn == 0;
    float bsum = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // both operands k-strided (a weight gradient: K = the row count of the activations, arbitrary): the last k-tile may be partial,
    // its k-rows past K are fetched from a block of zeros; otherwise K is a multiple of the tile depth (host check)
    constexpr bool TAIL = !A_KC && !B_KC;
    const int nk_all = TAIL ? (K + L32_BK - 1) / L32_BK : K / L32_BK;
    const int nk = (nk_all - wave + 3) / 4;                 // k-tiles wave, wave+4, wave+8, …
    const int krow0 = wave * L32_BK + lane / (T / 4);       // k-row of this lane in piece 0 of the wave's first tile (k-strided operands)
    char* const ring = l32_smem + wave * (NS * STAGE);

    const float* ga[PIECES];
    const float* gb[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        ga[i] = l32_src<T, A_KC>(A, lda, m0, M, wave * L32_BK, i, lane);
        gb[i] = l32_src<T, B_KC>(B, ldb, n0, N, wave * L32_BK, i, lane);
    }
    const size_t stepA = 4 * (A_KC ? (size_t)L32_BK : (size_t)L32_BK * lda);      // elements: this wave's next k-tile is 4 tiles on
    const size_t stepB = 4 * (B_KC ? (size_t)L32_BK : (size_t)L32_BK * ldb);

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define L32W_ISSUE(t)                                                                                                       \
    do {                                                                                                                    \
        char* st = ring + ((t) % NS) * STAGE;                                                                               \
        _Pragma("unroll") for (int i = 0; i < PIECES; ++i) {                                                                \
            const bool z_ = TAIL && krow0 + (t) * 4 * L32_BK + i * (256 / T) >= K;     /* piece i = k-rows i·(256/T) … of the tile */ \
            __builtin_amdgcn_global_load_lds((l32_gptr)(z_ ? l32_zeros : ga[i]), (l32_lptr)(st + i * 1024), 16, 0, 0);      \
            __builtin_amdgcn_global_load_lds((l32_gptr)(z_ ? l32_zeros : gb[i]), (l32_lptr)(st + OP + i * 1024), 16, 0, 0); \
            ga[i] += stepA; gb[i] += stepB;                                                                                 \
        }                                                                                                                   \
    } while (0)

    if (NS == 2 && nk > 0) L32W_ISSUE(0);
    for (int t = 0; t < nk; ++t) {
        if (NS == 1) {                             // one stage per wave (half the LDS → two workgroups per CU): load, wait, multiply
            L32W_ISSUE(t);
            l32_wait_vmcnt<0>();
        } else if (t + 1 < nk) {
            L32W_ISSUE(t + 1);
            l32_wait_vmcnt<2 * PIECES>();          // tile t landed; tile t+1 (16 pieces) may still be in flight
        } else {
            l32_wait_vmcnt<0>();
        }
        __builtin_amdgcn_wave_barrier();
        const char* sa = ring + (t % NS) * STAGE;
        const char* sb = sa + OP;
        if (want_asum) {                                    // k-strided image: [32 k-rows][64 columns] fp32, lane = column
            const float* img = reinterpret_cast<const float*>(sa);
#pragma unroll
            for (int q = 0; q < L32_BK; ++q) bsum += img[q * T + lane];
        }
#pragma unroll
        for (int ks = 0; ks < L32_BK / 16; ++ks) {
            if constexpr (X3) {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) l32_fragment2<T, B_KC>(sb, j * 32, ks, lane, bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bf16x8 ah, al;
                    l32_fragment2<T, A_KC>(sa, i * 32, ks, lane, ah, al);
#pragma unroll
                    for (int j = 0; j < 2; ++j) L32_MFMA3(ah, al, bh[j], bl[j], acc[i][j]);
                }
            } else {
                bf16x8 bf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = l32_fragment<T, B_KC>(sb, j * 32, ks, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 af = l32_fragment<T, A_KC>(sa, i * 32, ks, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        // the stage just read is refilled two iterations later by THIS wave (after its own fragment reads): no hazard
        __builtin_amdgcn_wave_barrier();
    }
#undef L32W_ISSUE

    // partial tiles → LDS ([wave][64][64] fp32, reusing the rings), summed in wave order
    __syncthreads();
    float* part = reinterpret_cast<float*>(l32_smem) + wave * (T * T);
    float* bpart = reinterpret_cast<float*>(l32_smem) + 4 * T * T;       // [wave][64] row sums of A
    if (want_asum) bpart[wave * T + lane] = bsum;
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                part[r * T + j * 32 + l31] = acc[i][j][e];
            }
    __syncthreads();
    const float* p0 = reinterpret_cast<const float*>(l32_smem);
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    constexpr int NE = T * T / 256;
    float vs[NE]; int rs[NE], cs[NE]; bool oks[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int idx = threadIdx.x + 256 * u;
        const int r = idx >> 6, c = idx & 63;
        const float v = ((p0[idx] + p0[T * T + idx]) + p0[2 * T * T + idx]) + p0[3 * T * T + idx];
        const int row = m0 + r, col = n0 + c;
        if constexpr (PRE) { if (row < M && col < N) C[(size_t)row * ldc + col] = v + cpre[u]; }
        else { vs[u] = v; oks[u] = row < M && col < N; rs[u] = min(row, M - 1); cs[u] = min(col, N - 1); }
    }
    if constexpr (!PRE) epilogue_store_n<NE>(vs, rs, cs, oks, C, ldc, epi, seed, inv_keep);      // every load of the epilogue in flight at once
    if (want_asum && threadIdx.x < T && m0 + (int)threadIdx.x < M) {
        const int i = threadIdx.x;
        asum[m0 + i] += ((bpart[i] + bpart[T + i]) + bpart[2 * T + i]) + bpart[3 * T + i];
    }
}

template <bool A_KC, bool B_KC, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_l32w_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        float* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_m,
                                                        int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char l32_smem[];
    const int tile = blockIdx.x;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    l32w_tile<A_KC, B_KC, 2, X3>(l32_smem, A, lda, B, ldb, C, ldc, M, N, K, epi, tm, tn, nullptr);
}

// ---- grouped weight gradients: up to 48 independent problems  dW_p[n_out, n_in] += dz_pᵀ · x_p  (+ db_p[n_out] += Σ_rows dz_p) in ONE
// launch.  The text-side and step-level linears each have a handful of 64² tiles; launched one by one they cost ≈12 µs apiece
// (plus two more launches for the bias gradient) for a few µs of work.  The problem table travels by value in the kernel arguments.
struct GProb { const float* dz; const float* x; float* dw; float* db; int n_out, n_in, rows, ld_dz, ld_x, ld_dw, tile0, tiles_n; };
constexpr int GROUP_MAX = 48;
struct GArgs { int n; GProb p[GROUP_MAX]; };

template <int NS>
__global__ __launch_bounds__(256) void gemm_group_wgrad_kernel(GArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char l32_smem[];
    int pi = 0;
    while (pi + 1 < g.n && (int)blockIdx.x >= g.p[pi + 1].tile0) ++pi;
    const GProb& q = g.p[pi];
    const int tile = blockIdx.x - q.tile0;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, 1, nullptr};
    l32w_tile<false, false, NS, false, true>(l32_smem, q.dz, q.ld_dz, q.x, q.ld_x, q.dw, q.ld_dw, q.n_out, q.n_in, q.rows, epi, tm, tn, q.db);
}


template <bool A_KC, bool B_KC, bool X3 = false>
static int l32w_launch_one(dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                           int K, Epi epi, int tiles_m, int tiles_n) {
    constexpr int LDS = 4 * 2 * 2 * 64 * L32_BK * 4;      // 4 waves × 2 stages × (A + B) = 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_l32w_kernel<A_KC, B_KC, X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { svpc_set_error("gemm_l32w: cannot raise the dynamic LDS limit"); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_l32w_kernel<A_KC, B_KC, X3>), grid, dim3(256), LDS, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m,
                       tiles_n);
    return 0;
}

template <int BM, int BN, bool A_KC, bool B_KC, int NS, bool X3 = false>
static int l32_launch_one(dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                          int K, Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk, float* slabs, int remap) {
    constexpr int LDS = NS * (BM + BN) * L32_BK * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_l32_kernel<BM, BN, A_KC, B_KC, NS, X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { svpc_set_error("gemm_l32: cannot raise the dynamic LDS limit"); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_l32_kernel<BM, BN, A_KC, B_KC, NS, X3>), grid, dim3(256), LDS, stream, A, lda, B, ldb, C, ldc, M, N, K, epi,
                       tiles_m, tiles_n, splitk, k_chunk, slabs, remap);
    return 0;
}
template <int BM, int BN, int NS, bool X3 = false>
static int l32_launch(int a_kc, int b_kc, dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                      int M, int N, int K, Epi epi, int tiles_m, int tiles_n, int splitk, int k_chunk, float* slabs, int remap) {
#define L32_ARGS grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk, slabs, remap
    if (a_kc && b_kc) return l32_launch_one<BM, BN, true, true, NS, X3>(L32_ARGS);
    if (a_kc) return l32_launch_one<BM, BN, true, false, NS, X3>(L32_ARGS);
    if (b_kc) return l32_launch_one<BM, BN, false, true, NS, X3>(L32_ARGS);
    return l32_launch_one<BM, BN, false, false, NS, X3>(L32_ARGS);
#undef L32_ARGS
}

// ---- grouped small GEMMs of one layout (e.g. the two directions of the BiLSTM at one time step): C_p = (acc ? C_p : 0) + A_p·B_pᵀ
struct GemmProb { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc, tile0, tiles_n; };
constexpr int GEMM_GROUP_MAX = 32;
struct GemmGroupArgs { int n; int accumulate; GemmProb p[GEMM_GROUP_MAX]; };

template <bool A_KC, bool B_KC, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroupArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char l32_smem[];
    int pi = 0;
    while (pi + 1 < g.n && (int)blockIdx.x >= g.p[pi + 1].tile0) ++pi;
    const GemmProb& q = g.p[pi];
    const int tile = blockIdx.x - q.tile0;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, g.accumulate, nullptr};
    l32w_tile<A_KC, B_KC, 2, X3>(l32_smem, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, epi, tm, tn, nullptr);
}

struct HostGemmProblem { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc; };

template <bool A_KC, bool B_KC, bool X3 = false>
static int gemm_group_go(const GemmGroupArgs& g, int tiles, hipStream_t stream) {
    constexpr int LDS = 4 * 2 * 2 * 64 * L32_BK * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_kernel<A_KC, B_KC, X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { svpc_set_error("gemm_group: cannot raise the dynamic LDS limit"); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_group_kernel<A_KC, B_KC, X3>), dim3(tiles), dim3(256), LDS, stream, g);
    return svpc_check_launch("gemm_group");
}

// ---- "skinny" form for the step-level problems (M ≤ 256 rows: 192 step vectors × 768…2304 features).  These launches are pure
// latency: a 64² tile with an LDS ring walks 6+ dependent k-tiles per wave.  Here a workgroup owns ONE 32×32 output tile, its 4 waves
// take every 4th 16-deep k-step, and each wave loads its MFMA fragments STRAIGHT from global memory into registers — up to 12
// k-steps (192 VGPRs of fp32 operands) are requested before the first is consumed, so K = 768 costs one memory round trip.
// Partial tiles meet in LDS and are summed in wave order (deterministic); the whole workgroup runs the epilogue.
// A k-contiguous [M][lda]; B k-contiguous [N][ldb] (B_KC) or k-strided [K][ldb]; K % 16 == 0; rows past an edge are clamped.
// the k-loop of one 32×32 tile: this wave's share of the 16-deep k-steps, partial tile → part[wave] (the caller syncs and sums).
// arow / brow: this lane's operand rows (A row lane&31, B column lane&31) already advanced by 8·(lane>>5) k-elements.
template <bool B_KC, int NW, bool X3 = false>
__device__ __forceinline__ void skinny_partials(float (*part)[16][64], const float* __restrict__ arow, const float* __restrict__ brow,
                                                int ldb, int K) {
    constexpr int BATCH = NW == 8 ? 10 : 12;          // (12 spills 21 registers in the 512-thread form)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsteps = K >> 4;                                  // 16-deep k-steps; this wave takes wave, wave+NW, …
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int s0 = wave; s0 < nsteps; s0 += NW * BATCH) {
        float4 av[BATCH][2];
        float bv[BATCH][8];
        // (no predicate around the loads: a k-step past the end re-reads the last one and is not multiplied.  Inside `if (s < nsteps)` the
        // loads of every k-step sat in a region of their own with a drain of the memory counter behind it — the "up to 12 k-steps
        // requested before the first is consumed" of the comment above were 12 memory round trips in a row: tools/isa_audit.py)
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int s = min(s0 + NW * u, nsteps - 1);
            const float* ap = arow + 16 * s;
            av[u][0] = *reinterpret_cast<const float4*>(ap);
            av[u][1] = *reinterpret_cast<const float4*>(ap + 4);
            if (B_KC) {
                const float* bp = brow + 16 * s;
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                bv[u][0] = b0.x; bv[u][1] = b0.y; bv[u][2] = b0.z; bv[u][3] = b0.w;
                bv[u][4] = b1.x; bv[u][5] = b1.y; bv[u][6] = b1.z; bv[u][7] = b1.w;
            } else {
                const float* bp = brow + (size_t)(16 * s) * ldb;
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[u][j] = bp[(size_t)j * ldb];
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int s = s0 + NW * u;
            if (s < nsteps) {
                if constexpr (X3) {
                    const float a8[8] = {av[u][0].x, av[u][0].y, av[u][0].z, av[u][0].w, av[u][1].x, av[u][1].y, av[u][1].z, av[u][1].w};
                    bf16x8 ah, al, bh, bl;
                    l32_split8(a8, ah, al);
                    l32_split8(bv[u], bh, bl);
                    L32_MFMA3(ah, al, bh, bl, acc);
                } else {
                    bf16x8 af, bf;
                    af[0] = (__bf16)av[u][0].x; af[1] = (__bf16)av[u][0].y; af[2] = (__bf16)av[u][0].z; af[3] = (__bf16)av[u][0].w;
                    af[4] = (__bf16)av[u][1].x; af[5] = (__bf16)av[u][1].y; af[6] = (__bf16)av[u][1].z; af[7] = (__bf16)av[u][1].w;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[j] = (__bf16)bv[u][j];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave][e][lane] = acc[e];
    __syncthreads();
}
template <bool B_KC, int NW, bool X3 = false>
__device__ __forceinline__ void skinny_tile(float (*part)[16][64], const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                            float* __restrict__ C, int ldc, int M, int N, int K, const Epi& epi, int tm, int tn,
                                            int ts = 32) {
    // ts: the tile's useful edge (32, or 24: M = 192 × N = 768 is 8 × 32 = 256 tiles of 24², one per CU, each pulling 147 KB instead of
    // the 196 KB of 144 tiles of 32² — these launches are bound by what ONE CU pulls; lanes 24..31 re-read row 23)
    const int lane = threadIdx.x & 63;
    const int m0 = tm * ts, n0 = tn * ts;
    const int r = min(lane & 31, ts - 1), h = lane >> 5;
    const float* __restrict__ arow = A + (size_t)min(m0 + r, M - 1) * lda + 8 * h;
    const int bcol = min(n0 + r, N - 1);
    const float* __restrict__ brow = B_KC ? B + (size_t)bcol * ldb + 8 * h : B + (size_t)(8 * h) * ldb + bcol;
    skinny_partials<B_KC, NW, X3>(part, arow, brow, ldb, K);
    const u64 seed = epi.p_drop > 0.f ? epi.seed[0] : 0ull;
    const float inv_keep = epi.p_drop > 0.f ? 1.0f / (1.0f - epi.p_drop) : 1.0f;
    const int col = threadIdx.x & 31;
    constexpr int NE = 16 / NW;
    float vs[NE]; int rs[NE], cs[NE]; bool oks[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int row = (threadIdx.x >> 5) + 2 * NW * i;        // accumulator element e of lane l holds row (e&3) + 8(e>>2) + 4(l>>5), column l&31
        const int e = (row & 3) + 4 * (row >> 3), l = col + 32 * ((row >> 2) & 1);
        float v = part[0][e][l];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += part[w][e][l];        // wave order: deterministic
        vs[i] = v;
        oks[i] = row < ts && col < ts && m0 + row < M && n0 + col < N;
        rs[i] = min(m0 + row, M - 1); cs[i] = min(n0 + col, N - 1);
    }
    epilogue_store_n<NE>(vs, rs, cs, oks, C, ldc, epi, seed, inv_keep);      // every load of the epilogue in flight at once
}
template <bool B_KC, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, int M, int N, int K, Epi epi, int tiles_n, int ts) {
    __shared__ float part[4][16][64];
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    skinny_tile<B_KC, 4, X3>(part, A, lda, B, ldb, C, ldc, M, N, K, epi, tm, tn, ts);
}
// grouped form (A k-contiguous; B k-contiguous, or — B_KC = false — k-strided [K][ldb]: the LSTM's recurrent dgrad reads W_hh (4D, D) in
// place, no transposed copy per step), NW waves per 32×32 tile: 8 for the long reductions of the LSTM dgrad (K = 3072)
template <int NW, bool X3 = false, bool B_KC = true>
__global__ __launch_bounds__(64 * NW) void gemm_group_skinny_kernel(GemmGroupArgs g) {
    __shared__ float part[NW][16][64];
    int pi = 0;
    while (pi + 1 < g.n && (int)blockIdx.x >= g.p[pi + 1].tile0) ++pi;
    const GemmProb& q = g.p[pi];
    const int tile = blockIdx.x - q.tile0;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    Epi epi{nullptr, ACT_NONE, 0.f, 0u, nullptr, g.accumulate, nullptr};
    skinny_tile<B_KC, NW, X3>(part, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, epi, tm, tn);
}

// ---- one time step of BOTH LSTM directions, recurrent projection and cell in one launch (reference: nn.LSTM inside
// model.py:1022-1024; per step and direction gates = gx[row] + h_{t-1}·W_hhᵀ, then the cell).  A workgroup owns 8 hidden units of one
// direction: its 32 tile columns are those units' i | f | g | o gate rows of W_hh, so after the wave-split k-loop (K = D, one memory
// round trip) the four gate sums of a (video, unit) pair meet in LDS and the cell is evaluated there — no (N, 4D) gate buffer between
// a GEMM launch and a cell launch.  Inactive videos (sequence ended) carry their state.  Same arithmetic as gemm_group_skinny +
// lstm_pair_fwd (bf16 MFMA operands rounded from fp32, fp32 accumulate, wave-ordered sums).
struct LstmStep {
    const float* h_prev[2]; const float* c_prev[2]; const float* w_hh[2]; const float* gx[2]; const int* rows[2]; const float* active;
    float* h[2]; float* c[2]; float* gates[2]; int N, D;
};
template <bool X3>
__global__ __launch_bounds__(256) void lstm_pair_step_fwd_kernel(LstmStep a) {
    __shared__ float part[4][16][64];
    const int z = blockIdx.y, u0 = blockIdx.x * 8, m0 = blockIdx.z * 32;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int N = a.N, D = a.D;
    const float* arow = a.h_prev[z] + (size_t)min(m0 + r, N - 1) * D + 8 * hh;
    const float* brow = a.w_hh[z] + (size_t)((r >> 3) * D + u0 + (r & 7)) * D + 8 * hh;      // tile column r ↔ gate r>>3 of unit u0 + (r&7)
    // what the cell needs besides the gate sums — the input projections of its (video, unit), the old cell / hidden state, the activity
    // flag — is requested BEFORE the recurrent projection's k-loop and lands under it (two dependent round trips after it otherwise)
    const int row = threadIdx.x >> 3, j = threadIdx.x & 7, n = m0 + row;
    const int nc = min(n, N - 1), dpre = u0 + j;
    const float* gxp = a.gx[z] + (size_t)a.rows[z][nc] * 4 * D + dpre;
    const float gx0 = gxp[0], gx1 = gxp[D], gx2 = gxp[2 * D], gx3 = gxp[3 * D];
    const float cp_pre = a.c_prev[z][(size_t)nc * D + dpre], hp_pre = a.h_prev[z][(size_t)nc * D + dpre], act_pre = a.active[nc];
    skinny_partials<true, 4, X3>(part, arow, brow, D, D);
    if (n >= N) return;
    float gsum[4];
    const int e = (row & 3) + 4 * (row >> 3), lb = 32 * ((row >> 2) & 1);      // accumulator element / lane half that hold this row
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int l = g * 8 + j + lb;
        float v = part[0][e][l];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += part[w][e][l];
        gsum[g] = v;
    }
    const int d = u0 + j;
    const size_t i = (size_t)n * D + d, g0 = (size_t)n * 4 * D + d;
    const float gi = sigmoidf_(gx0 + gsum[0]);
    const float gf = sigmoidf_(gx1 + gsum[1]);
    const float gg = tanhf(gx2 + gsum[2]);
    const float go = sigmoidf_(gx3 + gsum[3]);
    float* ga = a.gates[z];
    ga[g0] = gi; ga[g0 + D] = gf; ga[g0 + 2 * D] = gg; ga[g0 + 3 * D] = go;
    const float cp = cp_pre;
    const float cn = gf * cp + gi * gg;
    const float hn = go * tanhf(cn);
    const float act = act_pre;
    a.c[z][i] = act * cn + (1.f - act) * cp;
    a.h[z][i] = act * hn + (1.f - act) * hp_pre;
}

extern "C" {

static int lstm_pair_step_fwd_x(const float* const* h_prev, const float* const* c_prev, const float* const* w_hh, const float* const* gx,
                                const int* const* rows, const float* active, float* const* h, float* const* c, float* const* gates, int N,
                                int D, int x3, hipStream_t stream) {
    if (N == 0) return 0;
    SVPC_REQUIRE(D % 16 == 0 && D >= 16, "lstm_pair_step_fwd: hidden size must be a multiple of 16");
    LstmStep a{};
    for (int z = 0; z < 2; ++z) {
        SVPC_REQUIRE(((((uintptr_t)h_prev[z]) | ((uintptr_t)w_hh[z])) & 15) == 0, "lstm_pair_step_fwd: 16-byte aligned state / weight rows");
        a.h_prev[z] = h_prev[z]; a.c_prev[z] = c_prev[z]; a.w_hh[z] = w_hh[z]; a.gx[z] = gx[z]; a.rows[z] = rows[z];
        a.h[z] = h[z]; a.c[z] = c[z]; a.gates[z] = gates[z];
    }
    a.active = active; a.N = N; a.D = D;
    if (x3) hipLaunchKernelGGL(lstm_pair_step_fwd_kernel<true>, dim3(D / 8, 2, ceil_div(N, 32)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(lstm_pair_step_fwd_kernel<false>, dim3(D / 8, 2, ceil_div(N, 32)), dim3(256), 0, stream, a);
    return svpc_check_launch("lstm_pair_step_fwd");
}
int svpc_lstm_pair_step_fwd(const float* const* h_prev, const float* const* c_prev, const float* const* w_hh, const float* const* gx,
                            const int* const* rows, const float* active, float* const* h, float* const* c, float* const* gates, int N,
                            int D, hipStream_t stream) {
    return lstm_pair_step_fwd_x(h_prev, c_prev, w_hh, gx, rows, active, h, c, gates, N, D, 0, stream);
}
// the same with bf16x3 products (≤1e-4-parity throughput mode)
int svpc_lstm_pair_step_fwd_x3(const float* const* h_prev, const float* const* c_prev, const float* const* w_hh, const float* const* gx,
                               const int* const* rows, const float* active, float* const* h, float* const* c, float* const* gates, int N,
                               int D, hipStream_t stream) {
    return lstm_pair_step_fwd_x(h_prev, c_prev, w_hh, gx, rows, active, h, c, gates, N, D, 1, stream);
}

// 1 if the fp32 direct-to-LDS kernel can run this (shape, layout): whole k-tiles, 16-byte aligned chunks, and for a k-strided
// operand a row count that is a multiple of 4 (its 16-byte chunks run along the rows)
int svpc_gemm_l32_supported(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || lda % 4 != 0 || ldb % 4 != 0) return 0;
    // a k tail (K % 32 != 0) only on the single-problem tiled form, both operands k-contiguous, whole 16-byte chunks
    if (K % L32_BK != 0 && !(a_kc && b_kc && K % 4 == 0 && K > L32_BK)) return 0;
    if (!a_kc && (M % 4 != 0)) return 0;
    if (!b_kc && (N % 4 != 0)) return 0;
    return 1;
}

// 1 if this kernel is also the FASTER choice (measured on MI355X, tools/small_gemms.py sweep): grids below ≈300 128²-tiles
// with a short k-loop.  Larger grids are bandwidth-bound, where the 8-wave register-staged kernel moves more bytes per CU;
// long k-loops on few tiles want split-K slabs, which that kernel already does.
int svpc_gemm_l32_preferred(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K) {
    if (!svpc_gemm_l32_supported(a_kc, b_kc, lda, ldb, M, N, K)) return 0;
    const int t128 = ceil_div(M, 128) * ceil_div(N, 128), t64 = ceil_div(M, 64) * ceil_div(N, 64);
    static int env_w = -1;
    if (env_w < 0) { const char* e = getenv("SVPC_L32_WAVESPLIT"); env_w = e ? atoi(e) : 1; }
    if (K % L32_BK != 0) return (t128 <= 300) ? 1 : 0;                         // k tail: the tiled form only
    if (env_w && t64 <= 256 && K >= 4 * L32_BK && t64 * 4 >= 32) return 1;      // wave-split-K form, any K
    return (t128 <= 300 && K < 2048) ? 1 : 0;
}

// dz/x/dw/db layout as in include/svpc_hip.h (svpc_wgrad_problem); every problem must satisfy svpc_gemm_l32_supported(0, 0, …)
struct HostWgradProblem { const float* dz; const float* x; float* dw; float* db; int n_out, n_in, rows, ld_dz, ld_x, ld_dw; };

int svpc_gemm_group_wgrad_max(void) { return GROUP_MAX; }

// fp32 operands, same layout flags for every problem (a_kc / b_kc as in svpc_gemm_l32), K % 32 == 0, ≤ 32 problems
static int gemm_group_x(const void* problems, int n, int a_kc, int b_kc, int accumulate, int x3, hipStream_t stream) {
    if (n == 0) return 0;
    SVPC_REQUIRE(n > 0 && n <= GEMM_GROUP_MAX, "gemm_group: 1..32 problems per launch");
    const HostGemmProblem* hp = reinterpret_cast<const HostGemmProblem*>(problems);
    GemmGroupArgs g{};
    g.n = n; g.accumulate = accumulate;
    int tiles = 0;
    static int env_skinny = -1;
    if (env_skinny < 0) { const char* e = getenv("SVPC_L32_SKINNY"); env_skinny = e ? atoi(e) : 1; }
    bool skinny = env_skinny && a_kc;               // every problem a few rows tall: 32×32 tiles with register-direct operands
    if (!b_kc && x3) skinny = false;                // (the k-strided B form exists for the one-term products of the backward)
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        if (hp[i].M > 256 || hp[i].N < 32 || (hp[i].K & 15) || (hp[i].lda & 3) || (hp[i].ldb & 3)) skinny = false;
        if (hp[i].K > kmax) kmax = hp[i].K;
    }
    const int T = skinny ? 32 : 64;
    for (int i = 0; i < n; ++i) {
        const HostGemmProblem& h = hp[i];
        SVPC_REQUIRE(svpc_gemm_l32_supported(a_kc, b_kc, h.lda, h.ldb, h.M, h.N, h.K) && h.K % L32_BK == 0 && ((((uintptr_t)h.A) | ((uintptr_t)h.B)) & 15) == 0,
                     "gemm_group: needs K % 32 == 0 and 16-byte aligned fp32 rows");
        GemmProb& q = g.p[i];
        q.A = h.A; q.B = h.B; q.C = h.C; q.M = h.M; q.N = h.N; q.K = h.K; q.lda = h.lda; q.ldb = h.ldb; q.ldc = h.ldc;
        q.tile0 = tiles; q.tiles_n = ceil_div(h.N, T);
        tiles += ceil_div(h.M, T) * q.tiles_n;
    }
    if (skinny && !b_kc) {
        if (kmax >= 1536) hipLaunchKernelGGL((gemm_group_skinny_kernel<8, false, false>), dim3(tiles), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((gemm_group_skinny_kernel<4, false, false>), dim3(tiles), dim3(256), 0, stream, g);
        return svpc_check_launch("gemm_group_skinny");
    }
    if (skinny) {
        if (x3) {
            if (kmax >= 1536) hipLaunchKernelGGL((gemm_group_skinny_kernel<8, true>), dim3(tiles), dim3(512), 0, stream, g);
            else hipLaunchKernelGGL((gemm_group_skinny_kernel<4, true>), dim3(tiles), dim3(256), 0, stream, g);
        } else if (kmax >= 1536) hipLaunchKernelGGL((gemm_group_skinny_kernel<8, false>), dim3(tiles), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((gemm_group_skinny_kernel<4, false>), dim3(tiles), dim3(256), 0, stream, g);
        return svpc_check_launch("gemm_group_skinny");
    }
    if (x3) {
        if (a_kc && b_kc) return gemm_group_go<true, true, true>(g, tiles, stream);
        if (a_kc) return gemm_group_go<true, false, true>(g, tiles, stream);
        if (b_kc) return gemm_group_go<false, true, true>(g, tiles, stream);
        return gemm_group_go<false, false, true>(g, tiles, stream);
    }
    if (a_kc && b_kc) return gemm_group_go<true, true>(g, tiles, stream);
    if (a_kc) return gemm_group_go<true, false>(g, tiles, stream);
    if (b_kc) return gemm_group_go<false, true>(g, tiles, stream);
    return gemm_group_go<false, false>(g, tiles, stream);
}
int svpc_gemm_group(const void* problems, int n, int a_kc, int b_kc, int accumulate, hipStream_t stream) {
    return gemm_group_x(problems, n, a_kc, b_kc, accumulate, 0, stream);
}
// the same with bf16x3 products (≤1e-4-parity throughput mode)
int svpc_gemm_group_x3(const void* problems, int n, int a_kc, int b_kc, int accumulate, hipStream_t stream) {
    return gemm_group_x(problems, n, a_kc, b_kc, accumulate, 1, stream);
}

int svpc_gemm_group_wgrad(const void* problems, int n, hipStream_t stream) {
    if (n == 0) return 0;
    SVPC_REQUIRE(n > 0 && n <= GROUP_MAX, "gemm_group_wgrad: 1..48 problems per launch");
    const HostWgradProblem* hp = reinterpret_cast<const HostWgradProblem*>(problems);
    GArgs g{};
    g.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const HostWgradProblem& h = hp[i];
        SVPC_REQUIRE(h.rows > 0 && h.n_out > 0 && h.n_in > 0 && h.n_out % 4 == 0 && h.n_in % 4 == 0 && h.ld_dz % 4 == 0 && h.ld_x % 4 == 0 &&
                         ((((uintptr_t)h.dz) | ((uintptr_t)h.x)) & 15) == 0,
                     "gemm_group_wgrad: n_out % 4, n_in % 4 and 16-byte aligned rows required (any row count)");
        GProb& q = g.p[i];
        q.dz = h.dz; q.x = h.x; q.dw = h.dw; q.db = h.db; q.n_out = h.n_out; q.n_in = h.n_in; q.rows = h.rows; q.ld_dz = h.ld_dz;
        q.ld_x = h.ld_x; q.ld_dw = h.ld_dw; q.tile0 = tiles; q.tiles_n = ceil_div(h.n_in, 64);
        tiles += ceil_div(h.n_out, 64) * q.tiles_n;
    }
    constexpr int LDS2 = 4 * 2 * 2 * 64 * L32_BK * 4, LDS1 = 4 * 64 * 64 * 4 + 1024;      // one stage per wave: the partial-tile area sets the size
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS2);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
        if (e != hipSuccess) { svpc_set_error("gemm_group_wgrad: cannot raise the dynamic LDS limit"); return (int)e; }
        attr_set = true;
    }
    // one stage per wave halves the LDS and lets two workgroups share a CU: these launches are thousands of 64² tiles with short
    // reductions (192 … 576 rows: ≤ 5 k-tiles per wave), one per CU at a time with the two-stage ring (measured: 1 < auto < 2)
    static int ns_env = -1;
    if (ns_env < 0) { const char* e = getenv("SVPC_GROUP_WGRAD_NS"); ns_env = e ? atoi(e) : 1; }
    const bool one_stage = ns_env != 2;
    if (one_stage) hipLaunchKernelGGL(gemm_group_wgrad_kernel<1>, dim3(tiles), dim3(256), LDS1, stream, g);
    else hipLaunchKernelGGL(gemm_group_wgrad_kernel<2>, dim3(tiles), dim3(256), LDS2, stream, g);
    return svpc_check_launch("gemm_group_wgrad");
}

// Same contract as svpc_gemm_mx with fp32 A, B, C (reference: every nn.Linear / matmul of model.py that is not on the
// clip-encoder bf16 stream — e.g. :620-663 decoder, :594-617 step-wise encoder, :742-823 simulator, :1017-1025 BiLSTM).
static int gemm_l32_x(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R,
                      const float* G, int gact, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed,
                      int accumulate, float* workspace, size_t workspace_bytes, int x3, hipStream_t stream);
int svpc_gemm_l32_r(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R, int M,
                    int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                    float* workspace, size_t workspace_bytes, hipStream_t stream) {
    return gemm_l32_x(A, lda, a_kc, B, ldb, b_kc, C, ldc, Z, R, nullptr, ACT_NONE, M, N, K, bias, act, p_drop, site, seed, accumulate, workspace,
                      workspace_bytes, 0, stream);
}
// … and with the activation-backward factor: C = (A·B) ⊙ gact'(G) + R, G (fp32, C's layout) = what the forward of activation `gact`
// kept (z for GELU, y for ReLU / sigmoid) — the fp32 twin of svpc_gemm_glds_rg: the dgrad of the projection that consumes an activated
// tensor writes the gradient of the pre-activation directly (step-wise encoder FFN, model.py:565-591); no bias / activation / dropout
int svpc_gemm_l32_rg(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, const float* R, const float* G,
                     int gact, int M, int N, int K, int accumulate, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    return gemm_l32_x(A, lda, a_kc, B, ldb, b_kc, C, ldc, nullptr, R, G, gact, M, N, K, nullptr, ACT_NONE, 0.f, 0u, nullptr, accumulate,
                      workspace, workspace_bytes, 0, stream);
}
// the same contract with bf16x3 products: every fp32 operand value enters as hi + lo bf16 terms, three MFMAs per product — the
// arithmetic of the ≤1e-4-parity throughput mode for every projection kept in fp32 storage (text side, step level, simulators, LSTM)
int svpc_gemm_l32_x3(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R, int M,
                     int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate,
                     float* workspace, size_t workspace_bytes, hipStream_t stream) {
    return gemm_l32_x(A, lda, a_kc, B, ldb, b_kc, C, ldc, Z, R, nullptr, ACT_NONE, M, N, K, bias, act, p_drop, site, seed, accumulate, workspace,
                      workspace_bytes, 1, stream);
}
int svpc_gemm_l32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N, int K,
                  const float* bias, int act, float p_drop, unsigned site, const u64* seed, int accumulate, float* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
    return svpc_gemm_l32_r(A, lda, a_kc, B, ldb, b_kc, C, ldc, Z, nullptr, M, N, K, bias, act, p_drop, site, seed, accumulate, workspace,
                           workspace_bytes, stream);
}
}  // extern "C"
// R (optional): fp32 addend with C's leading dimension, C = epi(A·B) + R — the residual-path gradient joining the dgrad of the
// projection that consumes the residual tensor (fp32 twin of svpc_gemm_glds_r)
static int gemm_l32_x(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R,
                      const float* G, int gact, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const u64* seed,
                      int accumulate, float* workspace, size_t workspace_bytes, int x3, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    SVPC_REQUIRE(svpc_gemm_l32_supported(a_kc, b_kc, lda, ldb, M, N, K) && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0,
                 "gemm_l32: needs K % 32 == 0 (or both operands k-contiguous and K % 4 == 0) and 16-byte aligned fp32 rows");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "gemm: dropout needs a seed pointer");
    Epi epi{bias, act, p_drop, site, seed, accumulate, Z, R, G, gact};
    static int env_tile = -1, env_split = -1, remap = -1;
    if (env_tile < 0) { const char* e = getenv("SVPC_L32_TILE"); env_tile = e ? atoi(e) : 0; }      // 128 | 64 (deep ring) | 65 (64, 4 stages)
    if (env_split < 0) { const char* e = getenv("SVPC_L32_SPLITK"); env_split = e ? atoi(e) : 0; }
    if (remap < 0) { const char* e = getenv("SVPC_GEMM_REMAP"); remap = e ? atoi(e) : 1; }
    static int env_skinny = -1;
    if (env_skinny < 0) { const char* e = getenv("SVPC_L32_SKINNY"); env_skinny = e ? atoi(e) : 1; }
    static int skinny_m = -1;
    if (skinny_m < 0) { const char* e = getenv("SVPC_L32_SKINNY_M"); skinny_m = e ? atoi(e) : 256; }
    if (env_skinny && a_kc && M <= skinny_m && N >= 32 && (K & 15) == 0 && (lda & 3) == 0 && (!b_kc || (ldb & 3) == 0)) {
        static int env_ts = -1;
        if (env_ts < 0) { const char* e = getenv("SVPC_L32_SKINNY_TS"); env_ts = e ? atoi(e) : 24; }
        // 24² tiles when they put at most one tile on a CU where 32² tiles leave CUs idle (M = 192, N = 768: 256 tiles instead of 144)
        const int t24 = ceil_div(M, 24) * ceil_div(N, 24), t32 = ceil_div(M, 32) * ceil_div(N, 32);
        const int ts = (env_ts == 24 && t24 <= 256 && t32 < t24 && t32 > 64) ? 24 : 32;
        const int tn_ = ceil_div(N, ts);
        dim3 grids(ceil_div(M, ts) * tn_);
        if (x3) {
            if (b_kc) hipLaunchKernelGGL((gemm_skinny_kernel<true, true>), grids, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tn_, ts);
            else hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), grids, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tn_, ts);
        } else if (b_kc) hipLaunchKernelGGL((gemm_skinny_kernel<true, false>), grids, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tn_, ts);
        else hipLaunchKernelGGL((gemm_skinny_kernel<false, false>), grids, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tn_, ts);
        return svpc_check_launch("gemm_skinny");
    }
    const int t128 = ceil_div(M, 128) * ceil_div(N, 128), t64 = ceil_div(M, 64) * ceil_div(N, 64);
    // 64² tiles with a 4-stage ring (two workgroups per CU) were the fastest form for every small grid in the sweep; 128² tiles
    // (one workgroup per CU) and the 8-stage ring stay selectable through SVPC_L32_TILE for experiments
    int mode = (M >= 96 && N >= 96 && t128 > 300) ? 128 : 65;
    // ≤ 256 tiles of 64² (one per CU at most) with at least 4 k-tiles: the wave-split-K form (no slabs, no reduce launch)
    static int env_w = -1;
    if (env_w < 0) { const char* e = getenv("SVPC_L32_WAVESPLIT"); env_w = e ? atoi(e) : 1; }
    if (env_w && t64 <= 256 && K >= 4 * L32_BK) mode = 66;
    if (env_tile == 128 || env_tile == 64 || env_tile == 65 || env_tile == 66) mode = env_tile;
    const bool k_tail = K % L32_BK != 0;           // the tiled form below takes it (one k-range per tile: no split-K, no wave split)
    if (k_tail && (mode == 66 || mode == 64)) mode = 65;
    if (mode == 66) {
        const int tm_ = ceil_div(M, 64), tn_ = ceil_div(N, 64);
        dim3 gridw(tm_ * tn_);
        int rcw;
        if (x3) {
            if (a_kc && b_kc) rcw = l32w_launch_one<true, true, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
            else if (a_kc) rcw = l32w_launch_one<true, false, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
            else if (b_kc) rcw = l32w_launch_one<false, true, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
            else rcw = l32w_launch_one<false, false, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
        } else if (a_kc && b_kc) rcw = l32w_launch_one<true, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
        else if (a_kc) rcw = l32w_launch_one<true, false>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
        else if (b_kc) rcw = l32w_launch_one<false, true>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
        else rcw = l32w_launch_one<false, false>(gridw, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tm_, tn_);
        if (rcw) return rcw;
        return svpc_check_launch("gemm_l32w");
    }
    const int BMN = mode == 128 ? 128 : 64;
    const int tiles_m = ceil_div(M, BMN), tiles_n = ceil_div(N, BMN), tiles = tiles_m * tiles_n;
    // split K only when the grid would leave most CUs idle AND every slice still has a long k-loop (a short loop is already
    // covered by the ring: all of its tiles are in flight at once, so splitting only adds the reduce launch)
    int splitk = 1;
    if (tiles <= 128 && K >= 2048) {
        splitk = 256 / tiles;
        const int max_by_k = K / 512;
        if (splitk > max_by_k) splitk = max_by_k;
        if (splitk > 32) splitk = 32;
    }
    if (env_split > 0) splitk = env_split;
    if (k_tail) splitk = 1;
    while (splitk > 1 && (size_t)splitk * M * N * sizeof(float) > workspace_bytes) --splitk;
    if (splitk < 1) splitk = 1;
    int k_chunk = ceil_div(ceil_div(K, splitk), L32_BK) * L32_BK;
    splitk = ceil_div(K, k_chunk);
    dim3 grid(tiles * splitk);
    int rc;
#define L32_ARGS a_kc, b_kc, grid, stream, A, lda, B, ldb, C, ldc, M, N, K, epi, tiles_m, tiles_n, splitk, k_chunk, workspace, remap
    if (x3) {
        if (mode == 128) rc = l32_launch<128, 128, 4, true>(L32_ARGS);
        else rc = l32_launch<64, 64, 4, true>(L32_ARGS);
    } else if (mode == 128) rc = l32_launch<128, 128, 4>(L32_ARGS);
    else if (mode == 64) rc = l32_launch<64, 64, 8>(L32_ARGS);
    else rc = l32_launch<64, 64, 4>(L32_ARGS);
#undef L32_ARGS
    if (rc) return rc;
    rc = svpc_check_launch("gemm_l32");
    if (rc) return rc;
    if (splitk > 1) {
        launch_splitk_reduce<float>(workspace, splitk, C, ldc, M, N, epi, stream);
        rc = svpc_check_launch("gemm_l32 splitk reduce");
    }
    return rc;
}
