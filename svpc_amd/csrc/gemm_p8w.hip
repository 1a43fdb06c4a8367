// Grouped weight gradients of the bf16 activation streams on the 8-phase 256×256×64 template ("p8w"):
//
//   dW_p[n_out, n_in] += dz_pᵀ · x_p      for up to 48 problems in ONE launch (reference: the weight half of every nn.Linear backward of
//                                          the clip encoder and the decoder, src/rtransformer/model.py:195-197, :230, :259, :281, :551, :620-663)
//
// Both operands are K-STRIDED (the contraction runs over the activation rows): each half-tile image is [64 rows][128 columns] as the
// rows lie in memory, and every MFMA fragment comes through ds_read_b64_tr_b16 with the conflict-free slot swizzle of gemm_p8t.hip.
// Schedule, barriers, prefetch and the counted vmcnt are gemm_p8.hip's (two wave groups one barrier interval apart, 16 MFMAs
// v_mfma_f32_16x16x32_bf16 per phase).  Row counts are arbitrary: k-rows past the end come from a block of zeros.
// Work items and balance as the round-1 ping-pong form (svpc_gemm_group_wgrad_bf16_ws): deep problems first; tiles dealt after the first
// round of the chip that are still deep are cut into k-parts that write 256×256 fp32 slabs, added in part order by a fix-up launch
// (deterministic, no atomics).  Epilogue: fp32 accumulate into the gradient arena (a lane holds 4 consecutive columns of one row).
#include "gemm_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short4w __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * p8w_gptr;
typedef void __attribute__((address_space(3))) * p8w_lptr;

constexpr int P8W_BK = 64;
constexpr int P8W_HALF = 128 * P8W_BK * 2;      // 16 KiB: [64 k-rows][128 columns]
constexpr int P8W_BUF = 4 * P8W_HALF;           // 64 KiB: A0 A1 B0 B1 of one k-tile
__device__ __attribute__((aligned(16))) const float p8w_zeros[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ bf16x8 p8w_frag_tr(const char* __restrict__ a) {
    typedef short4w __attribute__((address_space(3))) * lds_ptr;
    const short4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const short4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * 256));
    union { short s[8]; bf16x8 v; } u;
    u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
    u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
    return u.v;
}

// C[256 × 256 block at (0,0) of C] (+)= Σ_{k < K} A[k][m] · B[k][n];  A: [K][lda] (columns m < Mv valid), B: [K][ldb] (n < Nv valid)
template <bool ACCUM>
__device__ __forceinline__ void p8w_tile(char* __restrict__ smem, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ B, int ldb,
                                         float* __restrict__ C, int ldc, int Mv, int Nv, int K, float* __restrict__ bdst, int bmask, bool bstore) {
    const int nk = (K + P8W_BK - 1) / P8W_BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave's group = its 128-row half of C; wc = its 64-column strip

    // ---- staging: pieces 2·wave and 2·wave + 1 of every half-tile image; piece pi = k-rows 4·pi … 4·pi + 3, lane l ↔
    // (k-row 4·pi + (l >> 4), slot l & 15) holding logical 16-byte chunk slot ^ f(k-row), f(r) = 2·((r & 3) | ((r >> 1) & 4))
    const __bf16* ga[2][2];
    const __bf16* gb[2][2];
    int krow[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kr = 8 * wave + 4 * u + (lane >> 4);
        krow[u] = kr;
        const int fx = 2 * ((kr & 3) | ((kr >> 1) & 4));
        const int c8 = 8 * ((lane & 15) ^ fx);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ga[h][u] = A + (size_t)kr * lda + max(min(128 * h + c8, Mv - 8), 0);      // columns past the edge are clamped (never stored)
            gb[h][u] = B + (size_t)kr * ldb + max(min(128 * h + c8, Nv - 8), 0);
        }
    }
    const size_t stepA = (size_t)P8W_BK * lda, stepB = (size_t)P8W_BK * ldb;
    const __bf16* const zsrc = reinterpret_cast<const __bf16*>(p8w_zeros);
    char* const my = smem + wave * 2048;
    // the half-tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of k-tile t: two 1-KiB pieces per wave; k-rows ≥ K are zero-sourced
#define P8W_STAGE(t, which)                                                                                                \
    do {                                                                                                                   \
        char* dst_ = my + ((t) & 1) * P8W_BUF + (which) * P8W_HALF;                                                        \
        const bool z0_ = krow[0] + (t) * P8W_BK >= K, z1_ = krow[1] + (t) * P8W_BK >= K;                                   \
        const __bf16* s0_ = (which) < 2 ? ga[(which) & 1][0] + (size_t)(t) * stepA : gb[(which) & 1][0] + (size_t)(t) * stepB; \
        const __bf16* s1_ = (which) < 2 ? ga[(which) & 1][1] + (size_t)(t) * stepA : gb[(which) & 1][1] + (size_t)(t) * stepB; \
        __builtin_amdgcn_global_load_lds((p8w_gptr)(z0_ ? zsrc : s0_), (p8w_lptr)(dst_), 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds((p8w_gptr)(z1_ ? zsrc : s1_), (p8w_lptr)(dst_ + 1024), 16, 0, 0);                 \
    } while (0)
#define P8W_WAIT(t)                                                                                                        \
    do {                                                                                                                   \
        if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    } while (0)
#define P8W_SYNC()                                                                                                         \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
    // fragment: 16-column block cb (0..7) and k block kb of a [64][128] image (see gemm_p8t.hip)
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tfx = 2 * (tq + 4 * (tg & 1));
    const int tb_off = (8 * tg + tq) * 256 + ((tp & 1) << 3);
    const int tp1 = tp >> 1;
#define P8W_FRAG(img, cb, kb) p8w_frag_tr((img) + (kb) * (32 * 256) + tb_off + (((((2 * (cb)) ^ tfx)) | tp1) << 4))

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Bias gradient (bmask != 0: a problem with a bias): db[m] = Σ_k dz[k][m] is one more MFMA per dz fragment with an all-ones operand in
    // the x slot — every output column then holds the column sum of the fragment.  The wave with strip index wc takes the 16-column block
    // wc (bit 0 of bmask: during phase 0) and / or 4 + wc (bit 1: phase 2) of its half; the tile columns 0 and 1 of a problem share the
    // work (bit 0 / bit 1), so a tile pays 2 extra MFMAs per wave and k-tile, 8 accumulator registers.
    floatx4 bacc0 = floatx4{0.f, 0.f, 0.f, 0.f}, bacc1 = floatx4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#define P8W_BIAS(bacc, bit)                                                                                                   \
    do {                                                                                                                   \
        if (bmask & (bit)) {                                                                                               \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                             \
                switch (wc) {                                                                                              \
                    case 0: bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, afr[kb][0], bacc, 0, 0, 0); break;       \
                    case 1: bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, afr[kb][1], bacc, 0, 0, 0); break;       \
                    case 2: bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, afr[kb][2], bacc, 0, 0, 0); break;       \
                    default: bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, afr[kb][3], bacc, 0, 0, 0); break;      \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)

    if (nk > 0) {
        // ---- prologue: k-tile 0 whole, the B halves of k-tile 1
#pragma unroll
        for (int w = 0; w < 4; ++w) P8W_STAGE(0, w);
        if (nk > 1) { P8W_STAGE(1, 2); P8W_STAGE(1, 3); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8W_SYNC();
    if (wr == 1) P8W_SYNC();                              // group 1 runs one interval behind

    const int cb0 = (wc & 1) * 4;                        // the wave's first 16-column block inside its B half-tile
    for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t & 1) * P8W_BUF + wr * P8W_HALF;
        const char* sb = smem + (t & 1) * P8W_BUF + (2 + (wc >> 1)) * P8W_HALF;
        bf16x8 afr[2][4], b0[2][2], b1[2][2];
        // ---- phase 0: rows 0-63 × columns 0-31 of the wave tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[kb][j] = P8W_FRAG(sb, cb0 + j, kb);
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8W_FRAG(sa, i, kb);
        }
        if (t + 1 < nk) P8W_STAGE(t + 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8W_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[i][j], 0, 0, 0);
        P8W_BIAS(bacc0, 1);
        __builtin_amdgcn_s_setprio(0);
        P8W_SYNC();
        // ---- phase 1: rows 0-63 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[kb][j] = P8W_FRAG(sb, cb0 + 2 + j, kb);
        if (t + 1 < nk) P8W_STAGE(t + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8W_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8W_SYNC();
        // ---- phase 2: rows 64-127 × columns 32-63
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[kb][i] = P8W_FRAG(sa, 4 + i, kb);
        if (t + 2 < nk) P8W_STAGE(t + 2, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8W_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kb][j], afr[kb][i], acc[4 + i][2 + j], 0, 0, 0);
        P8W_BIAS(bacc1, 2);
        __builtin_amdgcn_s_setprio(0);
        P8W_SYNC();
        // ---- phase 3: rows 64-127 × columns 0-31 (fragments already in registers)
        if (t + 2 < nk) P8W_STAGE(t + 2, 3);
        if (wr == 1 && t + 1 < nk) P8W_WAIT(t);
        P8W_SYNC();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kb][j], afr[kb][i], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wr == 0 && t + 1 < nk) P8W_WAIT(t);
        P8W_SYNC();
    }
    if (wr == 0) P8W_SYNC();                              // both groups pass the same number of barriers: 2 + 8·nk
#undef P8W_STAGE
#undef P8W_WAIT
#undef P8W_SYNC
#undef P8W_FRAG
#undef P8W_BIAS

    // ---- epilogue: a lane holds columns 4q … 4q+3 of row l15 of every 16×16 block: one 16-byte read-modify-write each.  The reads of
    // TWO row blocks (8 × 16 bytes per lane) are issued together, then their sums are stored: written as one load → add → store per block
    // the compiler must keep every load behind the previous store (the addresses may alias for all it knows) — 32 dependent memory round
    // trips per tile instead of 4.
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i0 = 0; i0 < 8; i0 += 2) {
        floatx4 old[2][4];
        if (ACCUM) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = wr * 128 + (i0 + u) * 16 + l15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wc * 64 + j * 16 + 4 * q;
                    old[u][j] = (row < Mv && col + 4 <= Nv) ? *reinterpret_cast<const floatx4*>(C + (size_t)row * ldc + col)
                                                            : floatx4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = wr * 128 + (i0 + u) * 16 + l15;
            if (row >= Mv) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wc * 64 + j * 16 + 4 * q;
                if (col + 4 > Nv) continue;               // Nv % 8 == 0: a group of four is inside or outside as a whole
                floatx4* d = reinterpret_cast<floatx4*>(C + (size_t)row * ldc + col);
                if (ACCUM) *d = old[u][j] + acc[i0 + u][j];
                else *d = acc[i0 + u][j];
            }
        }
    }
    if (bmask && q == 0) {                                // lanes 0-15: column l15 of the wave's two blocks (every output column holds the sum)
        const int c0 = wr * 128 + wc * 16 + l15, c1 = c0 + 64;
        if (!bstore) {                                    // whole tile: the only writer of these bias-gradient entries
            if ((bmask & 1) && c0 < Mv) bdst[c0] += bacc0[0];
            if ((bmask & 2) && c1 < Mv) bdst[c1] += bacc1[0];
        } else {                                          // k-part: a 256-float strip behind the slabs, summed in part order by the fix-up
            if (bmask & 1) bdst[c0] = bacc0[0];
            if (bmask & 2) bdst[c1] = bacc1[0];
        }
    }
}

struct W8Prob {
    const __bf16* dz; const __bf16* x; float* dw; float* db; int n_out, n_in, rows, ld_dz, ld_x, ld_dw, tile0, tiles_n;
    int whole, split, slab0, stile0;      // leading tiles run whole (accumulating into dw), the others in `split` k-parts → slabs
};
constexpr int W8_MAX = 48;
struct W8Args { int n; int total; int nslabs; W8Prob p[W8_MAX]; };      // nslabs: the bias strips of the k-parts start behind that many slabs

// which of a tile's bias columns tile column tn computes: the blocks of phase 0 (columns c with (c & 64) == 0: bit 0) or of phase 2 (bit 1)
__device__ __forceinline__ int p8w_bias_mask(const W8Prob& q, int tn) {
    if (!q.db) return 0;
    if (q.tiles_n == 1) return 3;
    return tn == 0 ? 1 : (tn == 1 ? 2 : 0);
}

__global__ __launch_bounds__(512) void gemm_group_wgrad16_p8_kernel(W8Args g, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * P8W_BUF];
    const int wg = blockIdx.x;
    int pi = 0;
    while (pi + 1 < g.n && wg >= g.p[pi + 1].tile0) ++pi;
    const W8Prob& q = g.p[pi];
    const int item = wg - q.tile0;
    if (item < q.whole) {
        const int tm = item / q.tiles_n, tn = item - tm * q.tiles_n;
        const int m0 = tm * 256, n0 = tn * 256;
        p8w_tile<true>(smem, q.dz + m0, q.ld_dz, q.x + n0, q.ld_x, q.dw + (size_t)m0 * q.ld_dw + n0, q.ld_dw, min(256, q.n_out - m0),
                       min(256, q.n_in - n0), q.rows, q.db ? q.db + m0 : nullptr, p8w_bias_mask(q, tn), false);
        return;
    }
    const int st = (item - q.whole) / q.split, part = (item - q.whole) - st * q.split;
    const int tile = q.whole + st;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    const int units = (q.rows + P8W_BK - 1) / P8W_BK, upp = (units + q.split - 1) / q.split;
    const int k0 = min(q.rows, part * upp * P8W_BK), k1 = min(q.rows, k0 + upp * P8W_BK);
    const int m0 = tm * 256, n0 = tn * 256;
    const int sidx = q.slab0 + st * q.split + part;
    float* slab = slabs + (size_t)sidx * 65536;
    p8w_tile<false>(smem, q.dz + (size_t)k0 * q.ld_dz + m0, q.ld_dz, q.x + (size_t)k0 * q.ld_x + n0, q.ld_x, slab, 256,
                    min(256, q.n_out - m0), min(256, q.n_in - n0), k1 - k0, slabs + (size_t)g.nslabs * 65536 + (size_t)sidx * 256,
                    p8w_bias_mask(q, tn), true);
}
// dW += Σ_parts slab (part order): 16 workgroups per cut tile, 16 rows each.  Slab elements outside the valid block are never read.
__global__ __launch_bounds__(256) void wgrad16_p8_fixup_kernel(W8Args g, const float* __restrict__ slabs) {
    const int bt = blockIdx.x >> 4, chunk = blockIdx.x & 15;
    int pi = 0;
    while (pi + 1 < g.n && bt >= g.p[pi + 1].stile0) ++pi;
    const W8Prob& q = g.p[pi];
    const int st = bt - q.stile0, tile = q.whole + st;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    const float4* sl = reinterpret_cast<const float4*>(slabs + (size_t)(q.slab0 + st * q.split) * 65536);
    const int bm = p8w_bias_mask(q, tn);
    if (chunk == 0 && bm) {                               // bias gradient of a cut tile: the parts' strips, in part order
        const float* bs = slabs + (size_t)g.nslabs * 65536 + (size_t)(q.slab0 + st * q.split) * 256;
        const int c = tm * 256 + threadIdx.x;
        if (c < q.n_out && (bm & ((threadIdx.x & 64) ? 2 : 1))) {
            float v = bs[threadIdx.x];
            for (int p = 1; p < q.split; ++p) v += bs[(size_t)p * 256 + threadIdx.x];
            q.db[c] += v;
        }
    }
    // the old dW values of all four pieces are requested first, then each piece's parts together (written as load → add → load … → store
    // per piece every load waited for the one before it: tools/isa_audit.py)
    float4 oldv[4];
    bool okv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = chunk * 1024 + it * 256 + threadIdx.x;
        const int row = i >> 6, c4 = i & 63;
        const int gr = tm * 256 + row, gc = tn * 256 + 4 * c4;
        okv[it] = gr < q.n_out && gc < q.n_in;
        oldv[it] = okv[it] ? *reinterpret_cast<const float4*>(q.dw + (size_t)gr * q.ld_dw + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = chunk * 1024 + it * 256 + threadIdx.x;
        const int row = i >> 6, c4 = i & 63;
        const int gr = tm * 256 + row, gc = tn * 256 + 4 * c4;
        if (!okv[it]) continue;
        float4 pv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < q.split) pv[p] = sl[(size_t)p * 16384 + i];      // (split ≤ 8: wave-uniform)
        float4 v = pv[0];
#pragma unroll
        for (int p = 1; p < 8; ++p)
            if (p < q.split) { v.x += pv[p].x; v.y += pv[p].y; v.z += pv[p].z; v.w += pv[p].w; }
        float4 o = oldv[it];
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *reinterpret_cast<float4*>(q.dw + (size_t)gr * q.ld_dw + gc) = o;
    }
}

struct HostWgrad16ProblemP8 { const void* dz; const void* x; float* dw; float* db; int n_out, n_in, rows, ld_dz, ld_x, ld_dw; };

extern "C" {

// 1 if every problem of the table can run on this kernel: both widths ≥ 256 and multiples of 8, 16-byte aligned rows
int svpc_gemm_group_wgrad_bf16_p8_ok(const void* problems, int n) {
    if (n <= 0 || n > W8_MAX) return 0;
    const HostWgrad16ProblemP8* hp = reinterpret_cast<const HostWgrad16ProblemP8*>(problems);
    for (int i = 0; i < n; ++i) {
        const HostWgrad16ProblemP8& h = hp[i];
        if (h.n_out < 256 || h.n_in < 256 || (h.n_out & 7) || (h.n_in & 7) || (h.ld_dz & 7) || (h.ld_x & 7) || (h.ld_dw & 3) || h.rows <= 0 ||
            ((((uintptr_t)h.dz) | ((uintptr_t)h.x) | ((uintptr_t)h.dw)) & 15) != 0 || (((uintptr_t)h.db) & 3) != 0)
            return 0;
    }
    return 1;
}

// same problem table as svpc_gemm_group_wgrad_bf16_ws (svpc_wgrad_problem with bf16 dz / x, fp32 dw); db (optional, fp32 [n_out]): the bias
// gradient db += Σ_rows dz, taken from the dz tiles this launch holds in LDS anyway (one all-ones MFMA per fragment in the tile column 0).
// `workspace`: room for the k-part slabs of the balanced form (65,536 + 256 floats per part; without it every tile runs whole)
int svpc_gemm_group_wgrad_bf16_p8(const void* problems, int n, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (n == 0) return 0;
    SVPC_REQUIRE(svpc_gemm_group_wgrad_bf16_p8_ok(problems, n) == 1,
                 "gemm_group_wgrad_bf16_p8: 1..48 problems, widths >= 256 and multiples of 8, 16-byte aligned rows");
    const HostWgrad16ProblemP8* hp = reinterpret_cast<const HostWgrad16ProblemP8*>(problems);
    W8Args g{};
    g.n = n;
    static int split_env = -1, cus = -1;
    if (split_env < 0) { const char* e = getenv("SVPC_P8W_SPLIT"); split_env = e ? atoi(e) : 80; }   // target 64-deep k-tiles per part; 0 = never split
    if (cus < 0) {
        hipDeviceProp_t prop; int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int order[W8_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i) {                 // deepest reductions first: the short ones fill the tail of the launch
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && hp[order[j]].rows < hp[v].rows) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    int items = 0, slabs = 0, stiles = 0, seen = 0;
    for (int i = 0; i < n; ++i) {
        const HostWgrad16ProblemP8& h = hp[order[i]];
        W8Prob& q = g.p[i];
        q.dz = (const __bf16*)h.dz; q.x = (const __bf16*)h.x; q.dw = h.dw; q.db = h.db; q.n_out = h.n_out; q.n_in = h.n_in; q.rows = h.rows;
        q.ld_dz = h.ld_dz; q.ld_x = h.ld_x; q.ld_dw = h.ld_dw; q.tiles_n = ceil_div(h.n_in, 256);
        const int tp = ceil_div(h.n_out, 256) * q.tiles_n;
        const int units = ceil_div(h.rows, P8W_BK);
        int split = (split_env > 0 && workspace) ? units / split_env : 1;
        if (split > 8) split = 8;
        if (split < 2) split = 1;
        int whole = tp;
        if (split > 1) whole = seen >= cus ? 0 : (cus - seen < tp ? cus - seen : tp);
        if (split > 1 && (size_t)(slabs + (tp - whole) * split) * (65536 + 256) * sizeof(float) > workspace_bytes) { split = 1; whole = tp; }
        if (whole == tp) split = 1;
        q.whole = whole; q.split = split; q.slab0 = slabs; q.stile0 = stiles; q.tile0 = items;
        items += whole + (tp - whole) * split;
        if (split > 1) { slabs += (tp - whole) * split; stiles += tp - whole; }
        seen += tp;
    }
    g.total = items;
    g.nslabs = slabs;
    hipLaunchKernelGGL(gemm_group_wgrad16_p8_kernel, dim3(items), dim3(512), 0, stream, g, workspace);
    int rc = svpc_check_launch("gemm_group_wgrad_bf16_p8");
    if (rc || stiles == 0) return rc;
    hipLaunchKernelGGL(wgrad16_p8_fixup_kernel, dim3(stiles * 16), dim3(256), 0, stream, g, (const float*)workspace);
    return svpc_check_launch("gemm_group_wgrad_bf16_p8 fix-up");
}

}  // extern "C"
