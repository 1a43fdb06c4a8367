// MFMA attention core for sequences of ≤128 queries × ≤128 keys per (sequence, head), head dim 32 or 64 — the clip
// encoder (100×100×64), decoder self/cross attention and the step encoder of the hot path (reference semantics:
// src/rtransformer/model.py:194-219).  bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 softmax.
//
// One workgroup (4 waves) owns one (sequence, head).  Q (pre-scaled by 1/sqrt(dh)), K, V (and dO in backward) are read
// once from the packed fp32 projection output, rounded to bf16 and kept in LDS as [row][dh] images with (2·dh+16)-byte rows:
// the same image serves ds_read_b128 row fragments (conflict-free) and ds_read_b64_tr_b16 transposed fragments.
//
// Forward, wave w = query tile w (32 queries): Sᵀ = K·Qᵀ puts a query on a lane and its keys in registers, so the softmax
// row reduction is in-lane (+ one cross-half shuffle); the normalised (and dropped-out) P registers are directly the A operand
// of O = P·V ("accumulator as next operand", permuted-k form) with V fragments fetched by the transposed read in the same
// permuted key order.  Backward is one launch: pass 1 (wave = key tile) recomputes S, dP in the natural layout and
// accumulates dV = P̃ᵀ·dO, dK = dSᵀ·Q; pass 2 (wave = query tile) recomputes Sᵀ, dPᵀ and accumulates dQ = dS·K.
// No atomics, no HBM intermediates.  Bound: HBM (AI ≈ 25–50 FLOP/B): algorithmic bytes per (sequence, head) forward =
// (2·Lq + 2·Lk)·dh·4.
#include "common.h"
#include <stdlib.h>

#include "attn_common.h"

template <int DH> struct AImg { static constexpr int RS = DH * 2 + 16; };

// rows [0, len) of a (·, DH) matrix (fp32 or bf16 in HBM) → bf16 image (zero rows beyond len), optionally scaled.
// All of a thread's loads are issued before the first conversion/LDS store (the staging is latency-, not bandwidth-bound);
// ``stage_pair`` keeps two matrices in flight at once.
template <int DH, typename T, int AT, int NT = 256>
struct RowStage {
    static constexpr int EPU = sizeof(T) == 4 ? 4 : 8;          // elements per 16-byte staging unit
    static constexpr int UPR = DH / EPU, NIT = AT * UPR / NT;
    float4 raw[NIT];     // 16 bytes: 4 fp32 values or 8 bf16 values
    int tid;
    __device__ __forceinline__ RowStage(int t) : tid(t) {}
    // rows past the sequence are read from its last row and zeroed when the image is written (store): a per-lane predicate around the
    // load puts every load in an exec-masked region of its own, and the compiler then waits for each before it issues the next (seen in
    // the fp32 instances: 32 memory round trips in a row instead of one); a select right behind the load draws the wait up as well
    int len_;
    __device__ __forceinline__ void load(const T* __restrict__ src, int ld, int len) {
        len_ = len;
        if (len <= 0) {                                   // (wave-uniform)
#pragma unroll
            for (int it = 0; it < NIT; ++it) raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + NT * it, row = u / UPR, c = u - row * UPR;
            raw[it] = *reinterpret_cast<const float4*>(src + (size_t)min(row, len - 1) * ld + EPU * c);
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ img, float scale) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + NT * it, row = u / UPR, c = u - row * UPR;
            const bool ok = row < len_;
            const float4 rw = make_float4(ok ? raw[it].x : 0.f, ok ? raw[it].y : 0.f, ok ? raw[it].z : 0.f, ok ? raw[it].w : 0.f);
            if (sizeof(T) == 4) {
                bf16x4 b;
                b[0] = (__bf16)(rw.x * scale); b[1] = (__bf16)(rw.y * scale);
                b[2] = (__bf16)(rw.z * scale); b[3] = (__bf16)(rw.w * scale);
                *reinterpret_cast<bf16x4*>(img + row * AImg<DH>::RS + c * 8) = b;
            } else {
                union { float4 f; bf16x8 h; } cv;
                cv.f = rw;
                if (scale != 1.0f) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) cv.h[j] = (__bf16)((float)cv.h[j] * scale);
                }
                *reinterpret_cast<float4*>(img + row * AImg<DH>::RS + c * 16) = cv.f;
            }
        }
    }
};
template <int DH, typename T, int AT, int NT = 256>
__device__ __forceinline__ void stage_pair(int tid, char* imgA, const T* srcA, int ldA, int lenA, float scaleA,
                                           char* imgB, const T* srcB, int ldB, int lenB, float scaleB) {
    RowStage<DH, T, AT, NT> a(tid), b(tid);
    a.load(srcA, ldA, lenA);
    b.load(srcB, ldB, lenB);
    a.store(imgA, scaleA);
    b.store(imgB, scaleB);
}
template <int DH, typename T, int AT, int NT = 256>
__device__ __forceinline__ void stage_rows(int tid, char* __restrict__ img, const T* __restrict__ src, int ld, int len, float scale) {
    RowStage<DH, T, AT, NT> a(tid);
    a.load(src, ld, len);
    a.store(img, scale);
}
// one 32×32 accumulator tile → rows row0.. (valid below row_limit), columns col0..col0+31 of a row-major matrix.
// bf16: sub-dword stores are slow, so lane pairs swap one value and each lane stores two adjacent columns as one dword.
template <typename T>
__device__ __forceinline__ void store_tile(T* __restrict__ base, int ld, int row0, int row_limit, int col0, const floatx16& acc,
                                           float scale, int lane) {
    const int l31 = lane & 31;
    if (sizeof(T) == 4) {
        float* p = reinterpret_cast<float*>(base);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = row0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (r < row_limit) p[(size_t)r * ld + col0 + l31] = acc[e] * scale;
        }
    } else {
        __bf16* p = reinterpret_cast<__bf16*>(base);
        const bool odd = lane & 1;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float y0 = acc[e] * scale, y1 = acc[e + 1] * scale;
            const float py = __shfl_xor(odd ? y0 : y1, 1, 64);
            const int ee = e + (odd ? 1 : 0);
            const int r = row0 + (ee & 3) + 8 * (ee >> 2) + 4 * (lane >> 5);
            union { __bf16 h[2]; uint32_t u; } pk;
            pk.h[0] = (__bf16)(odd ? py : y0); pk.h[1] = (__bf16)(odd ? y1 : py);
            if (r < row_limit) *reinterpret_cast<uint32_t*>(p + (size_t)r * ld + col0 + (l31 & ~1)) = pk.u;
        }
    }
}
// bf16 tile accumulated TRANSPOSED (rows = 32 columns of the matrix in the accumulator's register pattern, lane & 31 = the matrix
// row): two v_permlane32_swap per pair of 4-column runs give a lane 8 consecutive columns → two 16-byte stores per tile and lane
// instead of eight shuffled dword stores.  `base` = row 0 of the matrix block, `col0` = first of the 32 columns.
__device__ __forceinline__ void store_tile_tr(__bf16* __restrict__ base, int ld, int row, int row_limit, int col0, const floatx16& acc,
                                              float scale, int lane) {
    const int lhi = lane >> 5;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        uint32_t y[4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int e = 4 * (k + g);
            union { __bf16 hh[2]; uint32_t u; } p0, p1;
            p0.hh[0] = (__bf16)(acc[e] * scale); p0.hh[1] = (__bf16)(acc[e + 1] * scale);
            p1.hh[0] = (__bf16)(acc[e + 2] * scale); p1.hh[1] = (__bf16)(acc[e + 3] * scale);
            y[2 * g] = p0.u; y[2 * g + 1] = p1.u;
        }
        auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
        if (row < row_limit) *reinterpret_cast<uint4*>(base + (size_t)row * ld + col0 + 8 * (k + lhi)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}
// MFMA operand fragment: 32 image rows starting at row0, 16 columns starting at 16·ds (lane = (row, half) → 8 columns)
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const char* __restrict__ img, int row0, int ds, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * AImg<DH>::RS + ds * 32 + (lane >> 5) * 16);
}
// transposed fragment for a product that sums over image ROWS r0..r0+15 in the accumulator-permuted order
// (element j of lane half h ↔ row r0 + 8(j>>2) + 4h + (j&3)); lane ↔ image column c0 + (lane & 31)
template <int RS>
__device__ __forceinline__ bf16x8 frag_tr_rs(const char* __restrict__ img, int r0, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g >> 1;
    const char* a = img + (r0 + 4 * h + q) * RS + (c0 + 16 * (g & 1) + 4 * p) * 2;
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 8 * RS));
    union { short s[8]; bf16x8 v; } u;
    u.s[0] = lo[0]; u.s[1] = lo[1]; u.s[2] = lo[2]; u.s[3] = lo[3];
    u.s[4] = hi[0]; u.s[5] = hi[1]; u.s[6] = hi[2]; u.s[7] = hi[3];
    return u.v;
}
template <int DH>
__device__ __forceinline__ bf16x8 frag_tr(const char* __restrict__ img, int r0, int c0, int lane) {
    return frag_tr_rs<AImg<DH>::RS>(img, r0, c0, lane);
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
    return r;
}

// group-wide sync: the 4 waves of a workgroup, or — PERWAVE — one wave with itself (LDS is in-order per wave; the fence keeps
// the compiler from moving the reads above the writes)
template <bool PERWAVE> __device__ __forceinline__ void group_sync() {
    if (PERWAVE) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

// PERWAVE (AT = 32 only): every wave owns a whole (sequence, head) — staging, softmax and both products — with its own LDS
// images and no workgroup barrier; 4 (sequence, head) pairs per workgroup.  For the 22-token decoder, ≤12-step and ≤3-slot
// sequences this puts 4× more problems in flight than one workgroup per pair (whose waves 1..3 idle after the staging).
template <int DH, typename T, int AT, bool PERWAVE>
__global__ __launch_bounds__(256, 2) void attn_mfma_fwd_kernel(MAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr int IB = AT * AImg<DH>::RS, NTL = AT / 32;
    constexpr int GROUP_BYTES = 3 * IB + AT * (int)sizeof(float), NT = PERWAVE ? 64 : 256;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31;
    const int wave = PERWAVE ? 0 : wv, tid = PERWAVE ? lane : (int)threadIdx.x;
    char* smem = smem_all + (PERWAVE ? wv * GROUP_BYTES : 0);
    char* Qs = smem; char* Ks = smem + IB; char* Vs = smem + 2 * IB;
    float* mterm = reinterpret_cast<float*>(smem + 3 * IB);
    const int sh = PERWAVE ? (int)blockIdx.x * 4 + wv : (int)blockIdx.x;
    if (PERWAVE && sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const DropCtx dctx(a.seed, a.site, a.p_drop);      // (the seed word is requested with the segment table, ahead of the row images: read where
                                                       // the draws are made it costs a memory round trip of its own in the middle of the kernel)
    {   // all three images in flight at once: one memory round trip for the prologue
        RowStage<DH, T, AT, NT> sk(tid), sv(tid), sq(tid);
        sk.load((const T*)a.K + (size_t)k_off * a.ldk + h * DH, a.ldk, k_len);
        sv.load((const T*)a.V + (size_t)k_off * a.ldv + h * DH, a.ldv, k_len);
        sq.load((const T*)a.Q + (size_t)q_off * a.ldq + h * DH, a.ldq, q_len);
        for (int j = tid; j < AT; j += NT)
            mterm[j] = j < k_len ? (1.0f - (a.key_mask ? a.key_mask[k_off + j] : 1.0f)) * -10000.0f : -INFINITY;
        sk.store(Ks, 1.0f); sv.store(Vs, 1.0f); sq.store(Qs, a.scale);
    }
    group_sync<PERWAVE>();
    const int q0 = 32 * wave;
    if (q0 >= q_len || wave >= NTL) return;
    const int nkt = (k_len + 31) >> 5;

    floatx16 st[NTL];
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[jt][e] = 0.f;
        if (jt < nkt) {
#pragma unroll
            for (int ds = 0; ds < DH / 16; ++ds)
                st[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(Ks, 32 * jt, ds, lane), frag_rows<DH>(Qs, q0, ds, lane),
                                                                 st[jt], 0, 0, 0);
        }
    }
    const int q = q0 + l31;                 // this lane's query
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {        // registers 4g..4g+3 ↔ four consecutive keys: their mask terms in one 16-byte LDS read
            const float4 m4 = *reinterpret_cast<const float4*>(mterm + 32 * jt + 8 * g + 4 * (lane >> 5));
            const float mt4[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int e = 4 * g; e < 4 * g + 4; ++e) {
                const int key = 32 * jt + acc_row(e, lane);
                float t = mt4[e & 3];
                if (a.causal) t = (key > q && key < k_len) ? -10000.0f : t;     // uniform branch: no per-element selects otherwise
                const float v = st[jt][e] + t;
                st[jt][e] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(st[jt][e], LOG2E, -mx * LOG2E)); st[jt][e] = p; sum += p; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lane < 32 && q < q_len && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + q] = mx + logf(sum);
    const uint32_t arow = dctx.row((u64)(s * a.H + h) * a.max_q + q);       // this lane's probability row
    floatx16 acc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        if (jt < nkt) {
            float pv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) pv[e] = st[jt][e] * inv;
            if (a.p_drop > 0.f) {
                dctx.mul16(arow, 32 * jt, lane, pv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8(&pv[8 * s2]);
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr<DH>(Vs, 32 * jt + 16 * s2, 32 * dt, lane), acc[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
        store_tile<T>((T*)a.O + (size_t)q_off * a.ldo + h * DH, a.ldo, q0, q_len, 32 * dt, acc[dt], 1.0f, lane);
}

// ---- forward for the bf16 activation streams with more than 32 rows per sequence (the clip encoder: 100 × 100 × 64).  Same
// arithmetic as attn_mfma_fwd_kernel, restructured for occupancy and wide memory operations — the kernel is latency-bound (one
// memory round trip, ≈50 KB per (sequence, head)), so what counts is how many pairs a CU has in flight and how few instructions
// each needs:
//   * Q never goes through LDS: a wave needs only its own 32 queries, and their MFMA operand fragments are exactly 16-byte row
//     pieces — loaded straight into registers (scaled there; 1/√64 is a power of two, so the bf16 values are the same).
//     LDS holds K and V only: 37 KB → FOUR workgroups per CU instead of two.
//   * K / V rows are staged with 16-byte loads and LDS stores (half the instructions of the 8-byte form).
//   * O is accumulated TRANSPOSED (Oᵀ = Vᵀ·P̃ᵀ: the V fragment in the A slot) so a lane owns one query row; two
//     v_permlane32_swap per pair of 4-column runs give it 8 consecutive head columns → four 16-byte stores per lane instead of
//     sixteen shuffled dword stores.
template <int DH>
__global__ __launch_bounds__(256, 4) void attn_stream_fwd_kernel(MAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int AT = 128, RS = AImg<DH>::RS, IB = AT * RS, NTL = AT / 32, UPR = DH / 8, NIT = AT * UPR / 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lhi = lane >> 5, tid = threadIdx.x;
    char* Ks = smem; char* Vs = smem + IB;
    float* mterm = reinterpret_cast<float*>(smem + 2 * IB);
    const int sh = blockIdx.x;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const DropCtx dctx(a.seed, a.site, a.p_drop);      // (the seed word is requested with the segment table, ahead of the row images: read where
                                                       // the draws are made it costs a memory round trip of its own in the middle of the kernel)
    const int q0 = 32 * wave;
    const __bf16* Kp = (const __bf16*)a.K + (size_t)k_off * a.ldk + h * DH;
    const __bf16* Vp = (const __bf16*)a.V + (size_t)k_off * a.ldv + h * DH;
    uint4 kraw[NIT], vraw[NIT];
    bf16x8 qf[DH / 16];
    {   // every global load of the prologue in flight at once
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it, row = u / UPR, c8 = u - row * UPR;
            kraw[it] = vraw[it] = make_uint4(0u, 0u, 0u, 0u);
            if (row < k_len) {
                kraw[it] = *reinterpret_cast<const uint4*>(Kp + (size_t)row * a.ldk + 8 * c8);
                vraw[it] = *reinterpret_cast<const uint4*>(Vp + (size_t)row * a.ldv + 8 * c8);
            }
        }
        const int qr = min(q0 + l31, q_len - 1);        // rows past the sequence repeat its last query (never stored)
        const __bf16* Qp = (const __bf16*)a.Q + (size_t)(q_off + max(qr, 0)) * a.ldq + h * DH + 8 * lhi;
#pragma unroll
        for (int ds = 0; ds < DH / 16; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(Qp + 16 * ds);
        for (int j = tid; j < AT; j += 256)
            mterm[j] = j < k_len ? (1.0f - (a.key_mask ? a.key_mask[k_off + j] : 1.0f)) * -10000.0f : -INFINITY;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it, row = u / UPR, c8 = u - row * UPR;
            *reinterpret_cast<uint4*>(Ks + row * RS + c8 * 16) = kraw[it];
            *reinterpret_cast<uint4*>(Vs + row * RS + c8 * 16) = vraw[it];
        }
#pragma unroll
        for (int ds = 0; ds < DH / 16; ++ds)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ds][j] = (__bf16)((float)qf[ds][j] * a.scale);
    }
    __syncthreads();
    if (q0 >= q_len) return;
    const int nkt = (k_len + 31) >> 5;

    floatx16 st[NTL];
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[jt][e] = 0.f;
        if (jt < nkt) {
#pragma unroll
            for (int ds = 0; ds < DH / 16; ++ds)
                st[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(Ks, 32 * jt, ds, lane), qf[ds], st[jt], 0, 0, 0);
        }
    }
    const int q = q0 + l31;                 // this lane's query
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {        // registers 4g..4g+3 ↔ four consecutive keys: their mask terms in one 16-byte LDS read
            const float4 m4 = *reinterpret_cast<const float4*>(mterm + 32 * jt + 8 * g + 4 * (lane >> 5));
            st[jt][4 * g] += m4.x; st[jt][4 * g + 1] += m4.y; st[jt][4 * g + 2] += m4.z; st[jt][4 * g + 3] += m4.w;
            mx = fmaxf(fmaxf(mx, st[jt][4 * g]), fmaxf(st[jt][4 * g + 1], fmaxf(st[jt][4 * g + 2], st[jt][4 * g + 3])));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(st[jt][e], LOG2E, -mx * LOG2E)); st[jt][e] = p; sum += p; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lane < 32 && q < q_len && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + q] = mx + logf(sum);
    const uint32_t arow = dctx.row((u64)(s * a.H + h) * a.max_q + q);       // this lane's probability row
    floatx16 acc[DH / 32];      // Oᵀ: rows = head columns 32·dt + acc_row(e), column = this lane's query
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        if (jt < nkt) {
            float pv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) pv[e] = st[jt][e] * inv;
            if (a.p_drop > 0.f) {
                dctx.mul16(arow, 32 * jt, lane, pv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = pack8(&pv[8 * s2]);
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<DH>(Vs, 32 * jt + 16 * s2, 32 * dt, lane), pb, acc[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one key tile at a time: the dropout hashes of later tiles must not be hoisted (registers)
    }
    __bf16* Op = (__bf16*)a.O + (size_t)(q_off + q) * a.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int k = 0; k < 4; k += 2) {            // runs k and k+1 of 4 columns → the 8-column chunks k (lanes 0-31) and k+1 (lanes 32-63)
            uint32_t y[4];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int e = 4 * (k + g);
                union { __bf16 hh[2]; uint32_t u; } p0, p1;
                p0.hh[0] = (__bf16)acc[dt][e]; p0.hh[1] = (__bf16)acc[dt][e + 1];
                p1.hh[0] = (__bf16)acc[dt][e + 2]; p1.hh[1] = (__bf16)acc[dt][e + 3];
                y[2 * g] = p0.u; y[2 * g + 1] = p1.u;
            }
            auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
            if (q < q_len) *reinterpret_cast<uint4*>(Op + 32 * dt + 8 * (k + lhi)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
}

template <int DH, typename T, int AT, bool PERWAVE>
__global__ __launch_bounds__(256, 2) void attn_mfma_bwd_kernel(MAttnArgs a) {   // 2 waves/SIMD: ≤ 256 registers, 2 workgroups per CU
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr bool TRS = sizeof(T) == 2;        // bf16 gradients: tiles accumulated transposed, a matrix row per lane (16-byte stores)
    constexpr int IB = AT * AImg<DH>::RS;
    // dSᵀ image for pass 2 ([key][query] bf16, rows padded by 16 B): written after pass 1 over the V and dO images, which pass 2
    // does not read (head dim 32 with 128-row images: those two are too small, the image gets LDS of its own)
    constexpr int TRS_ = AT * 2 + 16, T_BYTES = AT * TRS_, NQT = AT / 32;
    constexpr bool T_OVER = 2 * IB >= T_BYTES;
    constexpr int GROUP_BYTES = 4 * IB + 4 * AT * (int)sizeof(float) + (T_OVER ? 0 : T_BYTES), NT = PERWAVE ? 64 : 256;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31;
    const int wave = PERWAVE ? 0 : wv, tid = PERWAVE ? lane : (int)threadIdx.x;
    char* smem = smem_all + (PERWAVE ? wv * GROUP_BYTES : 0);
    char* Qs = smem; char* Ks = smem + IB; char* Vs = smem + 2 * IB; char* Ds = smem + 3 * IB;
    float* mterm = reinterpret_cast<float*>(smem + 4 * IB);
    float* lse = mterm + AT;
    float* delta = lse + AT;
    uint32_t* arow_tab = reinterpret_cast<uint32_t*>(delta + AT);          // dropout row hash of every query of the pair (attn_common.h)
    char* Ts = T_OVER ? Vs : reinterpret_cast<char*>(arow_tab + AT);
    const int sh = PERWAVE ? (int)blockIdx.x * 4 + wv : (int)blockIdx.x;
    if (PERWAVE && sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const DropCtx dctx(a.seed, a.site, a.p_drop);      // (the seed word is requested with the segment table, ahead of the row images: read where
                                                       // the draws are made it costs a memory round trip of its own in the middle of the kernel)
    // Every global load of the prologue (K, V, Q, dO images and the dO/O row segments of delta) is issued before the first
    // conversion or LDS store: one memory round trip instead of one per image / per delta row.
    {
        // (the two per-row scalars first, with clamped indices and no per-lane branch: written inside the store loop below they end up
        // behind the wait for the row images, each with a memory round trip of its own)
        float kmv = 1.0f, lsv = 0.f;
        if (tid < AT) {
            if (a.key_mask) kmv = a.key_mask[k_off + min(tid, max(k_len - 1, 0))];
            lsv = a.LSE[((size_t)s * a.H + h) * a.max_q + min(tid, max(q_len - 1, 0))];
        }
        RowStage<DH, T, AT, NT> sk(tid), sv(tid), sq(tid), sd(tid);
        sk.load((const T*)a.K + (size_t)k_off * a.ldk + h * DH, a.ldk, k_len);
        sv.load((const T*)a.V + (size_t)k_off * a.ldv + h * DH, a.ldv, k_len);
        sq.load((const T*)a.Q + (size_t)q_off * a.ldq + h * DH, a.ldq, q_len);
        sd.load((const T*)a.dO + (size_t)q_off * a.lddo + h * DH, a.lddo, q_len);
        // delta = rowsum(dO ⊙ O): two threads per row (NT = 2·AT), half a head row each
        const int r = tid >> 1, c0 = (tid & 1) * (DH / 2);
        constexpr int VW = sizeof(T) == 4 ? 4 : 8, NV = DH / 2 / VW;
        typedef float dvec __attribute__((ext_vector_type(4)));
        dvec x[NV], y[NV];
        if (r < q_len) {
            const T* po = (const T*)a.dO + (size_t)(q_off + r) * a.lddo + h * DH + c0;
            const T* oo = (const T*)a.O + (size_t)(q_off + r) * a.ldo + h * DH + c0;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                x[c] = reinterpret_cast<const dvec*>(po)[c];
                y[c] = reinterpret_cast<const dvec*>(oo)[c];
            }
        }
        static_assert(AT <= NT, "one thread per image row");
        if (tid < AT) {
            mterm[tid] = tid < k_len ? (1.0f - kmv) * -10000.0f : -INFINITY;
            lse[tid] = tid < q_len ? lsv : 0.f;
            arow_tab[tid] = dctx.row((u64)(s * a.H + h) * a.max_q + tid);
        }
        sk.store(Ks, 1.0f); sv.store(Vs, 1.0f); sq.store(Qs, a.scale); sd.store(Ds, 1.0f);
        float d = 0.f;
        if (r < q_len) {
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                if (sizeof(T) == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) d += x[c][j] * y[c][j];
                } else {
                    union { dvec f; bf16x8 h; } ux, uy;
                    ux.f = x[c]; uy.f = y[c];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += (float)ux.h[j] * (float)uy.h[j];
                }
            }
        }
        d += __shfl_xor(d, 1, 64);
        if ((tid & 1) == 0 && r < AT) delta[r] = d;
    }
    group_sync<PERWAVE>();
    const int nqt = (q_len + 31) >> 5, nkt = (k_len + 31) >> 5;

    bf16x8 keep[NQT][2];                          // this wave's key tile: dS against every query tile, as packed for the dK product
    // ---------------- pass 1: wave = key tile → dV, dK (natural layout: key on lane, queries in registers)
    if (wave < nkt) {
        const int k0 = 32 * wave, key = k0 + l31;
        const float mt = mterm[key];
        floatx16 dv[DH / 32], dk[DH / 32];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dv[dt][e] = 0.f; dk[dt][e] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            if (qt >= nqt) break;
            __builtin_amdgcn_sched_barrier(0);    // one query tile at a time (registers)
            floatx16 sc, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sc[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < DH / 16; ++ds) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(Qs, 32 * qt, ds, lane), frag_rows<DH>(Ks, k0, ds, lane), sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<DH>(Ds, 32 * qt, ds, lane), frag_rows<DH>(Vs, k0, ds, lane), dp, 0, 0, 0);
            }
            // keys ≥ k_len carry t = -inf (p = 0).  Queries: accumulator register e holds query 32·qt + 4·(lane>>5) + (e&3) + 8·(e>>2);
            // a group of four registers whose first query is already ≥ q_len is skipped outright (the last tile of a 100-row
            // sequence has 4 live rows: 4 of 16 registers) and contributes zeros to both products
            float pt[16], dsv[16];
            const int qb = 32 * qt + 4 * (lane >> 5);
            const int nvq = q_len - 32 * qt;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g < nvq) {
                    // the four queries of the group are consecutive: their log-sum-exp and delta terms in one 16-byte LDS read each
                    const float4 l4 = *reinterpret_cast<const float4*>(lse + qb + 8 * g), d4 = *reinterpret_cast<const float4*>(delta + qb + 8 * g);
                    const float lq4[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
                    float dm4[4] = {1.0f, 1.0f, 1.0f, 1.0f};       // registers 4g..4g+3 ↔ queries qb + 8g + 0..3 of this lane's key
                    if (a.p_drop > 0.f) {
                        const uint4 a4 = *reinterpret_cast<const uint4*>(arow_tab + qb + 8 * g);
                        const uint32_t ar4[4] = {a4.x, a4.y, a4.z, a4.w};
                        dctx.mul4(ar4, (uint32_t)key * SVPC_ATTN_PHI, dm4);
                    }
#pragma unroll
                    for (int e = 4 * g; e < 4 * g + 4; ++e) {
                        const int q = qb + (e & 3) + 8 * (e >> 2);
                        float t = mt;
                        if (a.causal) t = (key > q && key < k_len) ? -10000.0f : t;
                        const float p = __builtin_amdgcn_exp2f(fmaf(sc[e], LOG2E, (t - lq4[e & 3]) * LOG2E));
                        const float dm = dm4[e & 3];
                        pt[e] = p * dm;
                        dsv[e] = p * (dp[e] * dm - dq4[e & 3]);
                    }
                } else {
#pragma unroll
                    for (int e = 4 * g; e < 4 * g + 4; ++e) { pt[e] = 0.f; dsv[e] = 0.f; }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8(&pt[8 * s2]), da = pack8(&dsv[8 * s2]);
                keep[qt][s2] = da;
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt) {
                    if (TRS) {      // transposed tiles (head columns in registers, this lane's key = the matrix row): 16-byte row stores
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<DH>(Ds, 32 * qt + 16 * s2, 32 * dt, lane), pa, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<DH>(Qs, 32 * qt + 16 * s2, 32 * dt, lane), da, dk[dt], 0, 0, 0);
                    } else {
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr<DH>(Ds, 32 * qt + 16 * s2, 32 * dt, lane), dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_tr<DH>(Qs, 32 * qt + 16 * s2, 32 * dt, lane), dk[dt], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt) {
            if (TRS) {
                store_tile_tr((__bf16*)a.dV + (size_t)k_off * a.lddv + h * DH, a.lddv, key, k_len, 32 * dt, dv[dt], 1.0f, lane);
                store_tile_tr((__bf16*)a.dK + (size_t)k_off * a.lddk + h * DH, a.lddk, key, k_len, 32 * dt, dk[dt], 1.0f, lane);
            } else {
                store_tile<T>((T*)a.dV + (size_t)k_off * a.lddv + h * DH, a.lddv, k0, k_len, 32 * dt, dv[dt], 1.0f, lane);
                store_tile<T>((T*)a.dK + (size_t)k_off * a.lddk + h * DH, a.lddk, k0, k_len, 32 * dt, dk[dt], 1.0f, lane);   // Qs carries 1/sqrt(dh)
            }
        }
    }
    // ---------------- dSᵀ → LDS.  Every wave has finished reading the V / dO images (barrier); each key-tile wave then lays its rows
    // down — lane = key, a run of four consecutive queries per 8-byte store — over those images
    group_sync<PERWAVE>();
    if (wave < nkt) {
        const int krow = 32 * wave + l31;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            if (qt >= nqt) break;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                union { bf16x8 v; uint2 h[2]; } u;
                u.v = keep[qt][g >> 1];
                *reinterpret_cast<uint2*>(Ts + krow * TRS_ + (32 * qt + 8 * g + 4 * (lane >> 5)) * 2) = u.h[g & 1];
            }
        }
    }
    group_sync<PERWAVE>();
    // ---------------- pass 2: wave = query tile → dQᵀ = Kᵀ·dSᵀ (a query on a lane); dS comes back from LDS through the transposing
    // read in exactly the register order the key fragments use — no second evaluation of S, the softmax or the dropout hash
    if (wave < nqt) {
        const int q0 = 32 * wave, q = q0 + l31;
        floatx16 dq[DH / 32];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 da = frag_tr_rs<TRS_>(Ts, 32 * kt + 16 * s2, q0, lane);
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    dq[dt] = TRS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<DH>(Ks, 32 * kt + 16 * s2, 32 * dt, lane), da, dq[dt], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_tr<DH>(Ks, 32 * kt + 16 * s2, 32 * dt, lane), dq[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
            if (TRS) store_tile_tr((__bf16*)a.dQ + (size_t)q_off * a.lddq + h * DH, a.lddq, q, q_len, 32 * dt, dq[dt], a.scale, lane);
            else store_tile<T>((T*)a.dQ + (size_t)q_off * a.lddq + h * DH, a.lddq, q0, q_len, 32 * dt, dq[dt], a.scale, lane);
    }
}

// ---- bf16x3 forward for the split-stored activation streams (the ≤1e-4-parity throughput mode; layout: split16 in layernorm.hip,
// arithmetic: gemm_p8x3.hip).  Q, K, V arrive as hi + lo bf16 planes; both products are three-term split-bf16 sums
//     Sᵀ = K_hi·Q_loᵀ + K_lo·Q_hiᵀ + K_hi·Q_hiᵀ            P̃ = P̃_hi + P̃_lo (split in registers)
//     Oᵀ = V_hiᵀ·P̃_loᵀ + V_loᵀ·P̃_hiᵀ + V_hiᵀ·P̃_hiᵀ
// with fp32 accumulation, fp32 softmax and the same dropout draws as every other attention kernel of the library (the backward —
// attn_mfma_bwd_kernel on the hi planes — recomputes them).  Structure of attn_stream_fwd_kernel: Q fragments straight into
// registers, K / V planes staged with 16-byte units (four images: 74 KB → two workgroups per CU), O accumulated transposed and
// written as two planes of 16-byte pieces.  The 1/√dh scale is applied to the fp32 scores (exact for any dh).
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_stream_x3_fwd_kernel(X3AttnArgs xa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MAttnArgs& a = xa.m;
    constexpr int AT = 128, RS = AImg<DH>::RS, IB = AT * RS, NTL = AT / 32, UPR = DH / 8, NIT = AT * UPR / 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lhi = lane >> 5, tid = threadIdx.x;
    char* Kh = smem; char* Kl = smem + IB; char* Vh = smem + 2 * IB; char* Vl = smem + 3 * IB;
    float* mterm = reinterpret_cast<float*>(smem + 4 * IB);
    const int sh = blockIdx.x;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const DropCtx dctx(a.seed, a.site, a.p_drop);      // (the seed word is requested with the segment table, ahead of the row images: read where
                                                       // the draws are made it costs a memory round trip of its own in the middle of the kernel)
    const int q0 = 32 * wave;
    const __bf16* Kp = (const __bf16*)a.K + (size_t)k_off * a.ldk + h * DH;
    const __bf16* Vp = (const __bf16*)a.V + (size_t)k_off * a.ldv + h * DH;
    bf16x8 qh[DH / 16], ql[DH / 16];
    {   // every global load of the prologue in flight at once
        uint4 kh[NIT], kl[NIT], vh[NIT], vl[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it, row = u / UPR, c8 = u - row * UPR;
            kh[it] = kl[it] = vh[it] = vl[it] = make_uint4(0u, 0u, 0u, 0u);
            if (row < k_len) {
                const __bf16* kp = Kp + (size_t)row * a.ldk + 8 * c8;
                const __bf16* vp = Vp + (size_t)row * a.ldv + 8 * c8;
                kh[it] = *reinterpret_cast<const uint4*>(kp);
                kl[it] = *reinterpret_cast<const uint4*>(kp + xa.k_lo);
                vh[it] = *reinterpret_cast<const uint4*>(vp);
                vl[it] = *reinterpret_cast<const uint4*>(vp + xa.v_lo);
            }
        }
        const int qr = min(q0 + l31, q_len - 1);        // rows past the sequence repeat its last query (never stored)
        const __bf16* Qp = (const __bf16*)a.Q + (size_t)(q_off + max(qr, 0)) * a.ldq + h * DH + 8 * lhi;
#pragma unroll
        for (int ds = 0; ds < DH / 16; ++ds) {
            qh[ds] = *reinterpret_cast<const bf16x8*>(Qp + 16 * ds);
            ql[ds] = *reinterpret_cast<const bf16x8*>(Qp + xa.q_lo + 16 * ds);
        }
        for (int j = tid; j < AT; j += 256)
            mterm[j] = j < k_len ? (1.0f - (a.key_mask ? a.key_mask[k_off + j] : 1.0f)) * -10000.0f : -INFINITY;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it, row = u / UPR, c8 = u - row * UPR;
            *reinterpret_cast<uint4*>(Kh + row * RS + c8 * 16) = kh[it];
            *reinterpret_cast<uint4*>(Kl + row * RS + c8 * 16) = kl[it];
            *reinterpret_cast<uint4*>(Vh + row * RS + c8 * 16) = vh[it];
            *reinterpret_cast<uint4*>(Vl + row * RS + c8 * 16) = vl[it];
        }
    }
    __syncthreads();
    if (q0 >= q_len) return;
    const int nkt = (k_len + 31) >> 5;

    floatx16 st[NTL];
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[jt][e] = 0.f;
        if (jt < nkt) {
#pragma unroll
            for (int ds = 0; ds < DH / 16; ++ds) {
                const bf16x8 kfh = frag_rows<DH>(Kh, 32 * jt, ds, lane), kfl = frag_rows<DH>(Kl, 32 * jt, ds, lane);
                st[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql[ds], st[jt], 0, 0, 0);
                st[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[ds], st[jt], 0, 0, 0);
                st[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[ds], st[jt], 0, 0, 0);
            }
        }
    }
    const int q = q0 + l31;                 // this lane's query
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {        // registers 4g..4g+3 ↔ four consecutive keys: their mask terms in one 16-byte LDS read
            const float4 m4 = *reinterpret_cast<const float4*>(mterm + 32 * jt + 8 * g + 4 * (lane >> 5));
            st[jt][4 * g] = fmaf(st[jt][4 * g], a.scale, m4.x); st[jt][4 * g + 1] = fmaf(st[jt][4 * g + 1], a.scale, m4.y);
            st[jt][4 * g + 2] = fmaf(st[jt][4 * g + 2], a.scale, m4.z); st[jt][4 * g + 3] = fmaf(st[jt][4 * g + 3], a.scale, m4.w);
            mx = fmaxf(fmaxf(mx, st[jt][4 * g]), fmaxf(st[jt][4 * g + 1], fmaxf(st[jt][4 * g + 2], st[jt][4 * g + 3])));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(st[jt][e], LOG2E, -mx * LOG2E)); st[jt][e] = p; sum += p; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lane < 32 && q < q_len && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + q] = mx + logf(sum);
    const uint32_t arow = dctx.row((u64)(s * a.H + h) * a.max_q + q);       // this lane's probability row
    floatx16 acc[DH / 32];      // Oᵀ: rows = head columns 32·dt + acc_row(e), column = this lane's query
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        if (jt < nkt) {
            float pv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) pv[e] = st[jt][e] * inv;
            if (a.p_drop > 0.f) {
                dctx.mul16(arow, 32 * jt, lane, pv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 hh = (__bf16)pv[8 * s2 + j];
                    ph[j] = hh; pl[j] = (__bf16)(pv[8 * s2 + j] - (float)hh);
                }
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt) {
                    const bf16x8 vfh = frag_tr<DH>(Vh, 32 * jt + 16 * s2, 32 * dt, lane), vfl = frag_tr<DH>(Vl, 32 * jt + 16 * s2, 32 * dt, lane);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, acc[dt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one key tile at a time: the dropout hashes of later tiles must not be hoisted (registers)
    }
    __bf16* Op = (__bf16*)a.O + (size_t)(q_off + q) * a.ldo + h * DH;
#pragma unroll
    for (int plane = 0; plane < 2; ++plane)
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int k = 0; k < 4; k += 2) {            // runs k and k+1 of 4 columns → the 8-column chunks k (lanes 0-31) and k+1 (lanes 32-63)
                uint32_t y[4];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int e = 4 * (k + g);
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = acc[dt][e + j];
                        v4[j] = plane ? v - (float)(__bf16)v : v;
                    }
                    union { __bf16 hh[2]; uint32_t u; } p0, p1;
                    p0.hh[0] = (__bf16)v4[0]; p0.hh[1] = (__bf16)v4[1];
                    p1.hh[0] = (__bf16)v4[2]; p1.hh[1] = (__bf16)v4[3];
                    y[2 * g] = p0.u; y[2 * g + 1] = p1.u;
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
                if (q < q_len)
                    *reinterpret_cast<uint4*>(Op + (plane ? xa.o_lo : 0) + 32 * dt + 8 * (k + lhi)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
}

// ---- bf16x3 forward for SHORT split-stored sequences (≤ 32 queries × ≤ 32 keys per (sequence, head): the decoder's causal 22-token
// self-attention and its 2-3-slot cross-attention, model.py:620-663): one WAVE per (sequence, head), four pairs per workgroup, no
// workgroup barrier — the PERWAVE form of attn_mfma_fwd_kernel with the three-term products of attn_stream_x3_fwd_kernel.  K / V
// planes in wave-private LDS images (4 × 32 rows), Q fragments straight into registers; causal ∧ key-pad mask as one −10000 term.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_small_x3_fwd_kernel(X3AttnArgs xa) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const MAttnArgs& a = xa.m;
    constexpr int AT = 32, RS = AImg<DH>::RS, IB = AT * RS, UPR = DH / 8, NIT = AT * UPR / 64;
    constexpr int WAVE_BYTES = 4 * IB + AT * (int)sizeof(float);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lhi = lane >> 5;
    char* smem = smem_all + wv * WAVE_BYTES;
    char* Kh = smem; char* Kl = smem + IB; char* Vh = smem + 2 * IB; char* Vl = smem + 3 * IB;
    float* mterm = reinterpret_cast<float*>(smem + 4 * IB);
    const int sh = (int)blockIdx.x * 4 + wv;
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const DropCtx dctx(a.seed, a.site, a.p_drop);      // (the seed word is requested with the segment table, ahead of the row images: read where
                                                       // the draws are made it costs a memory round trip of its own in the middle of the kernel)
    const __bf16* Kp = (const __bf16*)a.K + (size_t)k_off * a.ldk + h * DH;
    const __bf16* Vp = (const __bf16*)a.V + (size_t)k_off * a.ldv + h * DH;
    bf16x8 qh[DH / 16], ql[DH / 16];
    {
        uint4 kh[NIT], kl[NIT], vh[NIT], vl[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = lane + 64 * it, row = u / UPR, c8 = u - row * UPR;
            kh[it] = kl[it] = vh[it] = vl[it] = make_uint4(0u, 0u, 0u, 0u);
            if (row < k_len) {
                const __bf16* kp = Kp + (size_t)row * a.ldk + 8 * c8;
                const __bf16* vp = Vp + (size_t)row * a.ldv + 8 * c8;
                kh[it] = *reinterpret_cast<const uint4*>(kp);
                kl[it] = *reinterpret_cast<const uint4*>(kp + xa.k_lo);
                vh[it] = *reinterpret_cast<const uint4*>(vp);
                vl[it] = *reinterpret_cast<const uint4*>(vp + xa.v_lo);
            }
        }
        const int qr = max(min(l31, q_len - 1), 0);       // rows past the sequence repeat its last query (never stored)
        const __bf16* Qp = (const __bf16*)a.Q + (size_t)(q_off + qr) * a.ldq + h * DH + 8 * lhi;
#pragma unroll
        for (int ds = 0; ds < DH / 16; ++ds) {
            qh[ds] = *reinterpret_cast<const bf16x8*>(Qp + 16 * ds);
            ql[ds] = *reinterpret_cast<const bf16x8*>(Qp + xa.q_lo + 16 * ds);
        }
        if (lane < AT) mterm[lane] = lane < k_len ? (1.0f - (a.key_mask ? a.key_mask[k_off + lane] : 1.0f)) * -10000.0f : -INFINITY;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = lane + 64 * it, row = u / UPR, c8 = u - row * UPR;
            *reinterpret_cast<uint4*>(Kh + row * RS + c8 * 16) = kh[it];
            *reinterpret_cast<uint4*>(Kl + row * RS + c8 * 16) = kl[it];
            *reinterpret_cast<uint4*>(Vh + row * RS + c8 * 16) = vh[it];
            *reinterpret_cast<uint4*>(Vl + row * RS + c8 * 16) = vl[it];
        }
    }
    group_sync<true>();
    floatx16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ds = 0; ds < DH / 16; ++ds) {
        const bf16x8 kfh = frag_rows<DH>(Kh, 0, ds, lane), kfl = frag_rows<DH>(Kl, 0, ds, lane);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql[ds], st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[ds], st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[ds], st, 0, 0, 0);
    }
    const int q = l31;                 // this lane's query
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 m4 = *reinterpret_cast<const float4*>(mterm + 8 * g + 4 * lhi);
        const float mt4[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int e = 4 * g; e < 4 * g + 4; ++e) {
            const int key = acc_row(e, lane);
            float t = mt4[e & 3];
            if (a.causal) t = (key > q && key < k_len) ? -10000.0f : t;
            const float v = fmaf(st[e], a.scale, t);
            st[e] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float p = __builtin_amdgcn_exp2f(fmaf(st[e], LOG2E, -mx * LOG2E)); st[e] = p; sum += p; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lane < 32 && q < q_len && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + q] = mx + logf(sum);
    const uint32_t arow = dctx.row((u64)(s * a.H + h) * a.max_q + q);       // this lane's probability row
    float pv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) pv[e] = st[e] * inv;
    if (a.p_drop > 0.f) {
        dctx.mul16(arow, 0, lane, pv);
    }
    floatx16 acc[DH / 32];      // Oᵀ: rows = head columns 32·dt + acc_row(e), column = this lane's query
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hh = (__bf16)pv[8 * s2 + j];
            ph[j] = hh; pl[j] = (__bf16)(pv[8 * s2 + j] - (float)hh);
        }
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt) {
            const bf16x8 vfh = frag_tr<DH>(Vh, 16 * s2, 32 * dt, lane), vfl = frag_tr<DH>(Vl, 16 * s2, 32 * dt, lane);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, acc[dt], 0, 0, 0);
        }
    }
    __bf16* Op = (__bf16*)a.O + (size_t)(q_off + q) * a.ldo + h * DH;
#pragma unroll
    for (int plane = 0; plane < 2; ++plane)
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                uint32_t y[4];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int e = 4 * (k + g);
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = acc[dt][e + j];
                        v4[j] = plane ? v - (float)(__bf16)v : v;
                    }
                    union { __bf16 hh[2]; uint32_t u; } p0, p1;
                    p0.hh[0] = (__bf16)v4[0]; p0.hh[1] = (__bf16)v4[1];
                    p1.hh[0] = (__bf16)v4[2]; p1.hh[1] = (__bf16)v4[3];
                    y[2 * g] = p0.u; y[2 * g + 1] = p1.u;
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(y[0], y[2], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(y[1], y[3], false, false);
                if (q < q_len)
                    *reinterpret_cast<uint4*>(Op + (plane ? xa.o_lo : 0) + 32 * dt + 8 * (k + lhi)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
}

// attention_pipe.hip: the persistent, LDS-DMA-pipelined forward for ≤ 104-row sequences of the bf16 / split streams
bool attn_pipe_supported(const X3AttnArgs& xa, int dh, bool x3);
int attn_pipe_fwd_launch(const X3AttnArgs& xa, bool x3, hipStream_t stream);
bool attn_pipe_bwd_supported(const MAttnArgs& a, int dh);
int attn_pipe_bwd_launch(const MAttnArgs& a, hipStream_t stream);

static int mattn_set_lds(const void* fn, size_t bytes) { return svpc_raise_lds_once(fn, "attn_mfma"); }   // once per kernel symbol, process-wide table (api.cpp)
static bool mattn_ok(int dh, int max_q, int max_k, int ldq, int ldk, int ldv, const void* Q, const void* K, const void* V, int dt) {
    const int al = 16, lm = dt ? 8 : 4;   // 16-byte staging units: 8 bf16 or 4 fp32 elements
    return (dh == 64 || dh == 32) && max_q <= AT_MAX && max_k <= AT_MAX && ldq % lm == 0 && ldk % lm == 0 && ldv % lm == 0 &&
           (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & (al - 1)) == 0;
}

template <typename T, int DH, int AT>
static int mattn_fwd_go(const MAttnArgs& a, int n_pairs, hipStream_t stream) {
    constexpr bool PW = (AT == 32);
    const size_t lds = (3 * (size_t)AT * AImg<DH>::RS + AT * sizeof(float)) * (PW ? 4 : 1);
    int rc = mattn_set_lds((const void*)attn_mfma_fwd_kernel<DH, T, AT, PW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<DH, T, AT, PW>), dim3(PW ? ceil_div(n_pairs, 4) : n_pairs), dim3(256), lds, stream, a);
    return svpc_check_launch("attn_mfma_fwd");
}
template <typename T, int DH, int AT>
static int mattn_bwd_go(const MAttnArgs& a, int n_pairs, hipStream_t stream) {
    constexpr bool PW = (AT == 32);
    constexpr size_t t_bytes = (size_t)AT * (AT * 2 + 16);         // dSᵀ image: lies over the V / dO images when they are large enough
    const size_t lds = (4 * (size_t)AT * AImg<DH>::RS + 4 * AT * sizeof(float) + (2 * (size_t)AT * AImg<DH>::RS >= t_bytes ? 0 : t_bytes)) *
                       (PW ? 4 : 1);
    int rc = mattn_set_lds((const void*)attn_mfma_bwd_kernel<DH, T, AT, PW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_mfma_bwd_kernel<DH, T, AT, PW>), dim3(PW ? ceil_div(n_pairs, 4) : n_pairs), dim3(256), lds, stream, a);
    return svpc_check_launch("attn_mfma_bwd");
}
// image height: 32 rows when every sequence has ≤ 32 queries and keys (decoder, step encoder, memory slots), else 128
template <int DH>
static int mattn_stream_fwd_go(const MAttnArgs& a, int n_pairs, hipStream_t stream) {
    const size_t lds = 2 * (size_t)128 * AImg<DH>::RS + 128 * sizeof(float);
    hipLaunchKernelGGL((attn_stream_fwd_kernel<DH>), dim3(n_pairs), dim3(256), lds, stream, a);
    return svpc_check_launch("attn_stream_fwd");
}
static bool mattn_stream_ok(const MAttnArgs& a) {       // 16-byte row pieces everywhere
    static int on = -1;
    if (on < 0) { const char* e = getenv("SVPC_ATTN_STREAM"); on = e ? atoi(e) : 1; }
    return on && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 &&
           ((((uintptr_t)a.Q) | ((uintptr_t)a.K) | ((uintptr_t)a.V) | ((uintptr_t)a.O)) & 15) == 0;
}
template <typename T>
static int mattn_fwd_launch(const MAttnArgs& a, int dh, int n_blocks, hipStream_t stream) {
    const bool small = a.max_q <= 32 && a.max_k <= 32;
    if (sizeof(T) == 2 && !small && !a.causal && mattn_stream_ok(a)) {    // (the causal decoder sequences take the 32-row form)
        X3AttnArgs xa{};
        xa.m = a;
        if (attn_pipe_supported(xa, dh, false)) return attn_pipe_fwd_launch(xa, false, stream);
        return dh == 64 ? mattn_stream_fwd_go<64>(a, n_blocks, stream) : mattn_stream_fwd_go<32>(a, n_blocks, stream);
    }
    if (dh == 64) return small ? mattn_fwd_go<T, 64, 32>(a, n_blocks, stream) : mattn_fwd_go<T, 64, 128>(a, n_blocks, stream);
    return small ? mattn_fwd_go<T, 32, 32>(a, n_blocks, stream) : mattn_fwd_go<T, 32, 128>(a, n_blocks, stream);
}
template <typename T>
static int mattn_bwd_launch(const MAttnArgs& a, int dh, int n_blocks, hipStream_t stream) {
    const bool small = a.max_q <= 32 && a.max_k <= 32;
    if (sizeof(T) == 2 && attn_pipe_bwd_supported(a, dh)) return attn_pipe_bwd_launch(a, stream);
    if (dh == 64) return small ? mattn_bwd_go<T, 64, 32>(a, n_blocks, stream) : mattn_bwd_go<T, 64, 128>(a, n_blocks, stream);
    return small ? mattn_bwd_go<T, 32, 32>(a, n_blocks, stream) : mattn_bwd_go<T, 32, 128>(a, n_blocks, stream);
}

extern "C" {

// 1 if the MFMA path supports this problem (head dim 32/64, ≤128 rows per sequence, rows in 16-byte units of the element type dt:
// 0 = fp32 → leading dimensions % 4, 1 = bf16 → % 8)
int svpc_attn_mfma_supported(int dh, int max_q, int max_k, int ldq, int ldk, int ldv, int dt) {
    return mattn_ok(dh, max_q, max_k, ldq, ldk, ldv, nullptr, nullptr, nullptr, dt) ? 1 : 0;
}

// dt: element type of Q, K, V, O (0 = fp32, 1 = bf16); LSE and the softmax are fp32
int svpc_attn_mfma_fwd_t(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int dt, float* LSE,
                         const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                         float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(mattn_ok(dh, max_q, max_k, ldq, ldk, ldv, Q, K, V, dt), "attn_mfma: unsupported shape/alignment");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "attn_mfma: dropout needs a seed pointer");
    SVPC_REQUIRE(dt == 0 || (ldo % 2 == 0 && ((((uintptr_t)O) & 3) == 0)), "attn_mfma: bf16 output needs 4-byte aligned rows");
    MAttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.seq = seq; a.n_seq = n_seq;
    a.H = H; a.max_q = max_q; a.max_k = max_k; a.key_mask = key_mask; a.causal = causal; a.scale = scale; a.p_drop = p_drop;
    a.site = site; a.seed = seed;
    return dt ? mattn_fwd_launch<__bf16>(a, dh, n_seq * H, stream) : mattn_fwd_launch<float>(a, dh, n_seq * H, stream);
}
int svpc_attn_mfma_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                       const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                       float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    return svpc_attn_mfma_fwd_t(Q, ldq, K, ldk, V, ldv, O, ldo, 0, LSE, seq, n_seq, H, dh, max_q, max_k, key_mask, causal, scale, p_drop,
                                site, seed, stream);
}

// bf16x3 forward over split-stored Q / K / V / O (hi planes at the given pointers with the rows' leading dimensions, lo planes
// q_lo / k_lo / v_lo / o_lo elements behind them); non-causal, ≤128 rows per sequence, head dim 32 / 64.  LSE as the other kernels.
// reference: BertSelfAttention core, src/rtransformer/model.py:194-219 (clip encoder).
int svpc_attn_x3_fwd(const void* Q, int ldq, int q_lo, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, void* O, int ldo,
                     int o_lo, float* LSE, const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                     float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream);
int svpc_attn_stream_x3_fwd(const void* Q, int ldq, int q_lo, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, void* O,
                            int ldo, int o_lo, float* LSE, const int* seq, int n_seq, int H, int dh, int max_q, int max_k,
                            const float* key_mask, float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    return svpc_attn_x3_fwd(Q, ldq, q_lo, K, ldk, k_lo, V, ldv, v_lo, O, ldo, o_lo, LSE, seq, n_seq, H, dh, max_q, max_k, key_mask, 0, scale,
                            p_drop, site, seed, stream);
}
// general entry: sequences of ≤ 32 queries and keys take the one-wave-per-pair kernel (causal allowed: the decoder's self-attention),
// longer ones the stream kernel (non-causal only)
int svpc_attn_x3_fwd(const void* Q, int ldq, int q_lo, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, void* O, int ldo,
                     int o_lo, float* LSE, const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                     float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    const bool small = max_q <= 32 && max_k <= 32;
    SVPC_REQUIRE(small || !causal, "attn_x3: causal masks only on sequences of <= 32 rows");
    SVPC_REQUIRE(mattn_ok(dh, max_q, max_k, ldq, ldk, ldv, Q, K, V, 1) && ldo % 8 == 0 && ((((uintptr_t)O)) & 15) == 0 &&
                     q_lo % 8 == 0 && k_lo % 8 == 0 && v_lo % 8 == 0 && o_lo % 8 == 0,
                 "attn_stream_x3: unsupported shape/alignment (16-byte row pieces in every plane)");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "attn_stream_x3: dropout needs a seed pointer");
    X3AttnArgs xa{};
    MAttnArgs& a = xa.m;
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.seq = seq; a.n_seq = n_seq;
    a.H = H; a.max_q = max_q; a.max_k = max_k; a.key_mask = key_mask; a.causal = causal; a.scale = scale; a.p_drop = p_drop;
    a.site = site; a.seed = seed;
    xa.q_lo = q_lo; xa.k_lo = k_lo; xa.v_lo = v_lo; xa.o_lo = o_lo;
    if (small) {
        const int n_pairs = n_seq * H;
        if (dh == 64) {
            const size_t l = 4 * (4 * (size_t)32 * AImg<64>::RS + 32 * sizeof(float));
            int rc = mattn_set_lds((const void*)attn_small_x3_fwd_kernel<64>, l);
            if (rc) return rc;
            hipLaunchKernelGGL((attn_small_x3_fwd_kernel<64>), dim3(ceil_div(n_pairs, 4)), dim3(256), l, stream, xa);
        } else {
            const size_t l = 4 * (4 * (size_t)32 * AImg<32>::RS + 32 * sizeof(float));
            hipLaunchKernelGGL((attn_small_x3_fwd_kernel<32>), dim3(ceil_div(n_pairs, 4)), dim3(256), l, stream, xa);
        }
        return svpc_check_launch("attn_small_x3_fwd");
    }
    if (attn_pipe_supported(xa, dh, true)) return attn_pipe_fwd_launch(xa, true, stream);
    const size_t lds = 4 * (size_t)128 * AImg<64>::RS + 128 * sizeof(float);
    if (dh == 64) {
        int rc = mattn_set_lds((const void*)attn_stream_x3_fwd_kernel<64>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((attn_stream_x3_fwd_kernel<64>), dim3(n_seq * H), dim3(256), lds, stream, xa);
    } else {
        const size_t lds32 = 4 * (size_t)128 * AImg<32>::RS + 128 * sizeof(float);
        hipLaunchKernelGGL((attn_stream_x3_fwd_kernel<32>), dim3(n_seq * H), dim3(256), lds32, stream, xa);
    }
    return svpc_check_launch("attn_stream_x3_fwd");
}

int svpc_attn_mfma_bwd_t(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo, int dt,
                         const float* LSE, const void* dO, int lddo, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv,
                         const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                         float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(mattn_ok(dh, max_q, max_k, ldq, ldk, ldv, Q, K, V, dt) && lddo % 4 == 0 && ((((uintptr_t)dO) & (dt ? 7 : 15)) == 0),
                 "attn_mfma: unsupported shape/alignment");
    SVPC_REQUIRE(ldo % (dt ? 8 : 4) == 0 && lddo % (dt ? 8 : 4) == 0 && (((uintptr_t)O | (uintptr_t)dO) & 15) == 0,
                 "attn_mfma: O and dO rows must be 16-byte aligned");
    SVPC_REQUIRE(dt == 0 || (lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0 && (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0),
                 "attn_mfma: bf16 gradients need 16-byte aligned rows");
    MAttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = const_cast<void*>(O); a.ldo = ldo;
    a.LSE = const_cast<float*>(LSE); a.seq = seq; a.n_seq = n_seq; a.H = H; a.max_q = max_q; a.max_k = max_k; a.key_mask = key_mask;
    a.causal = causal; a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
    return dt ? mattn_bwd_launch<__bf16>(a, dh, n_seq * H, stream) : mattn_bwd_launch<float>(a, dh, n_seq * H, stream);
}
int svpc_attn_mfma_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                       const float* LSE, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                       const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                       float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    return svpc_attn_mfma_bwd_t(Q, ldq, K, ldk, V, ldv, O, ldo, 0, LSE, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, seq, n_seq, H, dh, max_q,
                                max_k, key_mask, causal, scale, p_drop, site, seed, stream);
}

}  // extern "C"
