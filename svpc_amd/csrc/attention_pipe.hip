// Persistent, software-pipelined attention forward for the clip encoder's bf16 / split (bf16x3) activation streams: sequences of
// ≤ 104 queries × ≤ 104 keys per (sequence, head), head dim 64, non-causal (reference semantics: src/rtransformer/model.py:194-219;
// the arithmetic — three-term split-bf16 products, fp32 softmax, the library's dropout draws, LSE — is that of
// attn_stream_fwd_kernel / attn_stream_x3_fwd_kernel in attention_mfma.hip, which remain the forms for longer sequences).
//
// Why a second structure.  The one-workgroup-per-(sequence, head) kernels live load → barrier → compute → store with nothing in
// flight while they compute, and the co-resident workgroups of a CU run in lockstep: HBM idles while the chip computes and the
// matrix pipes idle while it loads (0.45 of the HBM roofline in round 3).  Here a workgroup is PERSISTENT and walks the pairs
// blockIdx.x, blockIdx.x + gridDim.x, …:
//   * wave 7 is a LOADER: it brings the K / V planes of the pair TWO ahead of the one being computed into a three-stage LDS ring by
//     LDS-DMA (global_load_lds_dwordx4: no staging registers), so a CU always has one to two pairs (53–106 KB in bf16x3) in flight;
//     its only waits are counted (`vmcnt` = the instructions of the newest pair), the compute waves never wait for memory it moves;
//   * waves 0–6 COMPUTE, two per SIMD: wave w = the 16 queries 16·w … of the current pair on v_mfma_f32_16x16x32_bf16 (7 × 16 = 112 ≥
//     104 rows: 89 % of the issued tiles are live, 78 % with four 32-query waves): Sᵀ = K·Qᵀ (a query on a lane column, its keys
//     over the four 16-lane groups and the accumulator registers: softmax = in-lane work + two cross-group shuffles), the P̃
//     registers are directly the B operand of Oᵀ = Vᵀ·P̃ᵀ (key order of a k-step permuted the same way in the V fragments, which come
//     through ds_read_b64_tr_b16).  Q fragments come straight from global memory into registers one pair ahead; O leaves by 16-byte
//     stores (v_permlane16_swap pairs the 4-column runs of two accumulator tiles) that nobody waits for;
//   * ONE workgroup barrier per pair (ring hand-over both ways: "pair c has landed" / "the stage of pair c − 1 is free").
// LDS images are [104 rows][128 B] with no padding (an LDS-DMA instruction writes 1 KiB = 8 rows lane-linearly); bank conflicts of
// the ds_read_b128 row fragments AND the ds_read_b64_tr_b16 transposed fragments are removed by one XOR swizzle of the 16-byte
// chunk index, chunk' = chunk ^ f(row), f = (row bit 1) << 2 | (row bit 2 ^ row bit 3) << 1 | (row bit 2), applied on the DMA's per-lane
// SOURCE address and on every read address (never on the LDS destination) — checked exhaustively by tools/lds_bank_check.py.  Rows
// past a sequence's end are filled from its last row (finite values; their scores carry −inf, their probabilities are exactly 0).
// Bound: HBM — algorithmic bytes per pair (2·Lq + 2·Lk)·64·e, e = 2 (bf16) or 4 (split).
#include "attn_common.h"
#include <stdlib.h>

typedef const void __attribute__((address_space(1))) * pp_gptr;
typedef void __attribute__((address_space(3))) * pp_lptr;
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int PP_ROWS = 104;                  // image rows: 13 DMA blocks of 8
constexpr int PP_RB = 128;                    // bytes per image row (64 bf16 head columns)
constexpr int PP_IMG = PP_ROWS * PP_RB;       // 13,312 B per plane image
constexpr int PP_NBLK = PP_ROWS / 8;
constexpr int PP_NSTAGE = 3;
constexpr int PP_NT = 7;                      // 16-row tiles per image (the last one has 8 rows)
constexpr int PP_MT = 128 * (int)sizeof(float);   // mask terms of one stage (keys 0 … 127)

__device__ __forceinline__ int pp_swz(int r) { return (((r >> 1) & 1) << 2) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 2) & 1); }

typedef short4v __attribute__((address_space(3))) * pp_tr_ptr;
__device__ __forceinline__ bf16x8 pp_join(short4v lo, short4v hi) {        // two 8-byte halves, no element shuffles
    union { short4v h[2]; bf16x8 v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}
typedef float pp_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 pp_h2 __attribute__((ext_vector_type(2)));
// two floats → one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pp_cvt2(float lo, float hi) {
    const pp_f2 v = {lo, hi};
    union { pp_h2 h; uint32_t u; } pk;
    pk.h = __builtin_convertvector(v, pp_h2);
    return pk.u;
}
__device__ __forceinline__ bf16x8 pp_frag(const uint32_t (&w)[4]) {
    union { uint32_t u[4]; bf16x8 v; } x;
    x.u[0] = w[0]; x.u[1] = w[1]; x.u[2] = w[2]; x.u[3] = w[3];
    return x.v;
}
__device__ __forceinline__ uint32_t pp_pack2(float lo, float hi) {
    union { __bf16 h[2]; uint32_t u; } pk;
    pk.h[0] = (__bf16)lo; pk.h[1] = (__bf16)hi;
    return pk.u;
}

template <bool X3>
__global__ __launch_bounds__(512) void attn_pipe_fwd_kernel(X3AttnArgs xa, int n_pairs) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const MAttnArgs& a = xa.m;
    constexpr int NPL = X3 ? 4 : 2, STAGE = NPL * PP_IMG, VPL = X3 ? 2 : 1;       // planes per stage; index of the first V plane
    constexpr int NI = NPL * PP_NBLK + 2;                                         // DMA instructions per pair (+ 2: the key mask)
    static_assert(NI <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x, G = gridDim.x;
    const int n_my = bid < n_pairs ? (n_pairs - bid + G - 1) / G : 0;
    float* const mterm_all = reinterpret_cast<float*>(smem + PP_NSTAGE * STAGE);
    // timing experiments only (dbg & 16): s_memtime stamps kept in LDS (an LDS write does not touch the loader's vmcnt bookkeeping) and
    // dumped by wave 0 at the end: slot [iteration][8] — 0-2 loader (after barrier, after issue, after the landing wait), 3-7 compute wave 0
    // (at barrier, after barrier, end of compute, after the Q wait, after the stores)
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(smem + PP_NSTAGE * (STAGE + PP_MT));
    const bool do_stamp = (xa.dbg & 16) && bid < 16;
#define PP_STAMP(it, slot) do { if (do_stamp && lane == 0 && (it) < 16) stamps[(it) * 8 + (slot)] = clock64(); } while (0)
    // Per-pair metadata: lane j keeps the segment-table entries of this workgroup's j-th pair (host check: ≤ 64 pairs per workgroup)
    // and every iteration broadcasts its own with v_readlane.  Loaded inside the loops they are VECTOR loads (the table is not provably
    // invariant), and the wait for one of them is a vmcnt(0) that drains the prefetches in flight.
    int m_qoff = 0, m_qlen = 0, m_koff = 0, m_klen = 0;
    if (lane < n_my) {
        const int s = (bid + lane * G) / a.H;
        m_qoff = a.seq[s]; m_qlen = a.seq[a.n_seq + s]; m_koff = a.seq[2 * a.n_seq + s]; m_klen = a.seq[3 * a.n_seq + s];
    }
    asm volatile("" : "+v"(m_qoff), "+v"(m_qlen), "+v"(m_koff), "+v"(m_klen));      // landed before either loop starts

    if (wave == 7) {
        // ------------------------------------------------------------------------------------------------ loader
        __builtin_amdgcn_s_setprio(2);
        const int lrow = lane >> 3, ch0 = (lane & 7) ^ pp_swz(lrow);      // chunk this lane sources for even 8-row blocks (odd: ^ 2)
        auto issue = [&](int j) {
            const int pair = bid + j * G, s = pair / a.H, h = pair - s * a.H;
            const int k_off = __builtin_amdgcn_readlane(m_koff, j), k_len = __builtin_amdgcn_readlane(m_klen, j);
            char* const st = smem + (j % PP_NSTAGE) * STAGE;
            const char* const Kp = reinterpret_cast<const char*>((const __bf16*)a.K + (size_t)k_off * a.ldk + h * 64);
            const char* const Vp = reinterpret_cast<const char*>((const __bf16*)a.V + (size_t)k_off * a.ldv + h * 64);
#pragma unroll 1
            for (int b = 0; b < PP_NBLK; ++b) {
                const int re = max(min(8 * b + lrow, k_len - 1), 0), ch = ch0 ^ ((b & 1) << 1);
                const unsigned ko = ((unsigned)re * (unsigned)a.ldk + 8u * ch) * 2u, vo = ((unsigned)re * (unsigned)a.ldv + 8u * ch) * 2u;
                char* const d = st + b * 1024;
                __builtin_amdgcn_global_load_lds((pp_gptr)(Kp + ko), (pp_lptr)d, 16, 0, 0);
                if (X3) __builtin_amdgcn_global_load_lds((pp_gptr)(Kp + ko + 2 * (size_t)xa.k_lo), (pp_lptr)(d + PP_IMG), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((pp_gptr)(Vp + vo), (pp_lptr)(d + VPL * PP_IMG), 16, 0, 0);
                if (X3) __builtin_amdgcn_global_load_lds((pp_gptr)(Vp + vo + 2 * (size_t)xa.v_lo), (pp_lptr)(d + 3 * PP_IMG), 16, 0, 0);
            }
            // the raw key-pad mask of the pair's keys (converted to additive terms once it has landed); without a mask any readable word
            char* const mt = reinterpret_cast<char*>(mterm_all) + (j % PP_NSTAGE) * PP_MT;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = a.key_mask ? a.key_mask + k_off + max(min(lane + 64 * i, k_len - 1), 0) : reinterpret_cast<const float*>(a.seq);
                __builtin_amdgcn_global_load_lds((pp_gptr)src, (pp_lptr)(mt + 256 * i), 4, 0, 0);
            }
        };
        const bool no_dma = xa.dbg & 1;
        if (n_my > 0 && !no_dma) issue(0);
        if (n_my > 1 && !no_dma) issue(1);
        for (int c = 0; c < n_my; ++c) {
            // everything but the newest pair's instructions has landed → pair c is in LDS
            if (c + 1 < n_my && !no_dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {
                const int k_len = __builtin_amdgcn_readlane(m_klen, c);
                const unsigned mt = (unsigned)(size_t)((pp_lptr)(reinterpret_cast<char*>(mterm_all) + (c % PP_NSTAGE) * PP_MT)) + 4u * lane;
                // (inline asm: a compiler-visible LDS read here would be preceded by vmcnt(0) — the DMA writes LDS — and drain the ring)
                float v0, v1;
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(mt) : "memory");
                if (!a.key_mask) { v0 = 1.0f; v1 = 1.0f; }
                v0 = lane < k_len ? (1.0f - v0) * (-10000.0f * LOG2E) : -INFINITY;        // (exp2 domain: the compute waves' scores carry log2 e)
                v1 = lane + 64 < k_len ? (1.0f - v1) * (-10000.0f * LOG2E) : -INFINITY;
                asm volatile("ds_write_b32 %2, %0\n\tds_write_b32 %2, %1 offset:256\n\ts_waitcnt lgkmcnt(0)" ::"v"(v0), "v"(v1), "v"(mt) : "memory");
            }
            PP_STAMP(c, 2);
            __builtin_amdgcn_s_barrier();                    // A_c: pair c landed; the stage of pair c − 1 is free
            PP_STAMP(c, 0);
            if (c + 2 < n_my && !no_dma) issue(c + 2);
            PP_STAMP(c, 1);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute waves
    const int g = lane >> 4, l15 = lane & 15, q0 = 16 * wave;
    const DropCtx dctx(a.seed, a.site, a.p_drop);
    // Lane parts of the fragment addresses (image-relative bytes).  K row fragment (key tile t, k-step ks = head columns 32·ks …):
    // row 16·t + l15, chunk 4·ks + g → (kb ^ ks << 6) + 2048·t; the last tile's rows stop at the image's last row (kb6).  V transposed
    // fragment (k-step u = keys 32·u …, half ab, column tile dt): row 32·u + 16·ab + 4·g + q, chunk 2·dt + (p >> 1) → (vb ^ dt << 5) +
    // 128·(32·u + 16·ab): a row's swizzle depends on its bits 1-3 only; keys 96-111 stop at the image's last row (vb3).
    const int kb = l15 * PP_RB + ((g ^ pp_swz(l15)) << 4);
    const int kr6 = min(96 + l15, PP_ROWS - 1), kb6 = kr6 * PP_RB + ((g ^ pp_swz(kr6)) << 4);
    const int vq = l15 >> 2, vp = l15 & 3, vr = 4 * g + vq;
    const int vb = vr * PP_RB + (((vp >> 1) ^ pp_swz(vr)) << 4) + 8 * (vp & 1);
    const int vr3 = min(96 + vr, PP_ROWS - 1), vb3 = vr3 * PP_RB + (((vp >> 1) ^ pp_swz(vr3)) << 4) + 8 * (vp & 1);
    // Q fragments: the next pair's are requested into a second register set before the barrier and land under this pair's arithmetic
    // (requested into the live set right after the score products, the compiler waits for them at once to copy them: measured)
    bf16x8 nqh[2], nql[2];
    auto load_q = [&](int j) {
        const int pair = bid + j * G, s = pair / a.H, h = pair - s * a.H;
        const int q_off = __builtin_amdgcn_readlane(m_qoff, j), q_len = __builtin_amdgcn_readlane(m_qlen, j);
        const int qr = max(min(q0 + l15, q_len - 1), 0);          // rows past the sequence repeat its last query (never stored)
        const __bf16* Qp = (const __bf16*)a.Q + (size_t)(q_off + qr) * a.ldq + h * 64 + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            nqh[ks] = *reinterpret_cast<const bf16x8*>(Qp + 32 * ks);
            if (X3) nql[ks] = *reinterpret_cast<const bf16x8*>(Qp + xa.q_lo + 32 * ks);
        }
    };
    auto land_q = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            asm volatile("" : "+v"(nqh[ks]));
            if (X3) asm volatile("" : "+v"(nql[ks]));
        }
    };
    if (n_my > 0) {
        load_q(0);
        land_q();                                 // the loop is entered with no load pending on either edge
    }
    for (int c = 0; c < n_my; ++c) {
        const int pair = bid + c * G, s = pair / a.H, h = pair - s * a.H;
        const int q_off = __builtin_amdgcn_readlane(m_qoff, c), q_len = __builtin_amdgcn_readlane(m_qlen, c), k_len = __builtin_amdgcn_readlane(m_klen, c);
        bf16x8 qh[2], ql[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qh[ks] = nqh[ks];
            if (X3) ql[ks] = nql[ks];
            else {                                // the bf16 form carries 1/sqrt(dh) on Q (exact: a power of two)
#pragma unroll
                for (int j = 0; j < 8; ++j) qh[ks][j] = (__bf16)((float)qh[ks][j] * a.scale);
            }
        }
        load_q(min(c + 1, n_my - 1));             // (unconditional — the last iteration re-requests its own: a conditional request makes the
                                                  // compiler merge register sets behind it, i.e. wait for the loads on the spot)
        if (wave == 0) PP_STAMP(c, 3);
        __builtin_amdgcn_s_barrier();                        // A_c
        if (wave == 0) PP_STAMP(c, 4);
        // LDS byte addresses of this iteration's fragments: stage base + lane part, everything else an instruction immediate
        // This iteration's fragment addresses = LDS base + (stage offset + lane part) + an instruction immediate.  The integer parts are
        // pinned in registers (empty asm): the compiler otherwise re-derives stage base + lane part in front of every single read.
        const int soff = (c % PP_NSTAGE) * STAGE;
        int ki0 = soff + kb, ki1 = soff + (kb ^ 64), ki60 = soff + kb6, ki61 = soff + (kb6 ^ 64);      // K row fragments, k-step 0 / 1
        int mi = PP_NSTAGE * STAGE + (c % PP_NSTAGE) * PP_MT + 16 * g;
        asm volatile("" : "+v"(ki0), "+v"(ki1), "+v"(ki60), "+v"(ki61), "+v"(mi));
        const char* const ka0 = smem + ki0; const char* const ka1 = smem + ki1;
        const char* const ka60 = smem + ki60; const char* const ka61 = smem + ki61;
        const float* const mta = reinterpret_cast<const float*>(smem + mi);
        if (q0 < q_len && !(xa.dbg & 2)) {
            // ---- Sᵀ = K·Qᵀ: tile t = keys 16·t …; register r of lane (g, l15) ↔ key 16·t + 4·g + r, query q0 + l15.  Every tile of the
            // image is computed: keys past the sequence read finite rows and carry −inf mask terms.
            floatx4 sc[PP_NT];
            const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // the K fragments of all seven tiles first, then the products TERM-major: consecutive MFMAs write different accumulators
                bf16x8 kfh[PP_NT], kfl[PP_NT];
#pragma unroll
                for (int t = 0; t < PP_NT; ++t) {
                    const char* const o = t < 6 ? (ks ? ka1 : ka0) + 2048 * t : (ks ? ka61 : ka60);
                    kfh[t] = *reinterpret_cast<const bf16x8*>(o);
                    if (X3) kfl[t] = *reinterpret_cast<const bf16x8*>(o + PP_IMG);
                }
                // (the very first product of an accumulator takes a literal zero: no register initialisation)
                if (X3) {
#pragma unroll
                    for (int t = 0; t < PP_NT; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh[t], ql[ks], ks ? sc[t] : zero4, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < PP_NT; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfl[t], qh[ks], sc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < PP_NT; ++t) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh[t], qh[ks], (X3 || ks) ? sc[t] : zero4, 0, 0, 0);
            }
            const int q = q0 + l15;                 // this lane's query
            // softmax in the exp2 domain: t = s·(scale·log2 e) + m·log2 e (the loader stores the mask terms already multiplied)
            const float cs = X3 ? a.scale * LOG2E : LOG2E;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < PP_NT; ++t) {       // registers 0..3 ↔ four consecutive keys: their mask terms in one 16-byte LDS read
                const float4 m4 = *reinterpret_cast<const float4*>(mta + 16 * t);
                sc[t][0] = fmaf(sc[t][0], cs, m4.x); sc[t][1] = fmaf(sc[t][1], cs, m4.y);
                sc[t][2] = fmaf(sc[t][2], cs, m4.z); sc[t][3] = fmaf(sc[t][3], cs, m4.w);
                mx = fmaxf(fmaxf(mx, sc[t][0]), fmaxf(sc[t][1], fmaxf(sc[t][2], sc[t][3])));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < PP_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float p = __builtin_amdgcn_exp2f(sc[t][r] - mx); sc[t][r] = p; sum += p; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            // (v_log_f32 / v_rcp_f32: 1 ulp, where logf and the IEEE division cost a dozen instructions each)
            if (g == 0 && q < q_len && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + q] = (mx + __builtin_amdgcn_logf(sum)) * (1.0f / LOG2E);
            // P stays unnormalised (and un-rescaled by the dropout's 1/(1-p)): O is multiplied once at the end
            const float oscale = dctx.ik * __builtin_amdgcn_rcpf(sum);
            // dropout draws of this lane's query row (common.h: svpc_attn_draw16): A + (4·g)·φ once, a literal added per element
            const uint32_t arow = dctx.row((u64)(s * a.H + h) * a.max_q + q) + (uint32_t)(4 * g) * SVPC_ATTN_PHI;
            // ---- Oᵀ = Vᵀ·P̃ᵀ: k-step u = keys 32·u …, element j of a lane's B fragment ↔ key 32·u + 16·(j >> 2) + 4·g + (j & 3)
            // = registers of the score tiles 2·u (j < 4) and 2·u + 1; the V fragments read the same keys in the same order
            floatx4 acc[4];      // tile dt: register r ↔ head column 16·dt + 4·g + r of this lane's query
            // V fragments: lane part ^ (dt << 5), one base per column tile; keys 96-111 from the row-clamped lane part
            const char* va[4]; const char* va3[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                int i0 = soff + VPL * PP_IMG + (vb ^ (dt << 5)), i3 = soff + VPL * PP_IMG + (vb3 ^ (dt << 5));
                asm volatile("" : "+v"(i0), "+v"(i3));
                va[dt] = smem + i0; va3[dt] = smem + i3;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 2 * u + (j >> 2);
                    pv[j] = t < PP_NT ? sc[t < PP_NT ? t : 0][j & 3] : 0.f;
                }
                if (a.p_drop > 0.f) {
#pragma unroll
                    for (int j = 0; j < (u < 3 ? 8 : 4); ++j)
                        pv[j] = svpc_attn_draw16(arow + (uint32_t)(32 * u + 16 * (j >> 2) + (j & 3)) * SVPC_ATTN_PHI) >= dctx.thr ? pv[j] : 0.0f;
                }
                // hi = bf16(p) and lo = bf16(p − hi), two values per v_cvt_pk_bf16_f32
                uint32_t ph[4], pl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ph[j] = pp_cvt2(pv[2 * j], pv[2 * j + 1]);
                    if (X3) pl[j] = pp_cvt2(pv[2 * j] - __uint_as_float(ph[j] << 16), pv[2 * j + 1] - __uint_as_float(ph[j] & 0xffff0000u));
                }
                const bf16x8 phv = pp_frag(ph), plv = X3 ? pp_frag(pl) : phv;
                // keys 32·u … +15 and … +16 … +31 (the second half of the last k-step, keys 112-127, is never live: zeros); all eight
                // fragments first, then the products term-major over the four column tiles
                bf16x8 vfh[4], vfl[4];
                const short4v z4 = {0, 0, 0, 0};
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* const oa = u < 3 ? va[dt] + 4096 * u : va3[dt]; const char* const ob = va[dt] + 4096 * u + 2048;
                    vfh[dt] = pp_join(__builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)oa),
                                      u < 3 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)ob) : z4);
                    if (X3) vfl[dt] = pp_join(__builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(oa + PP_IMG)),
                                              u < 3 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(ob + PP_IMG)) : z4);
                }
                if (X3) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh[dt], plv, u ? acc[dt] : zero4, 0, 0, 0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl[dt], phv, acc[dt], 0, 0, 0);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh[dt], phv, (X3 || u) ? acc[dt] : zero4, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // one k-step at a time: the dropout draws of later ones must not be hoisted (registers)
            }
            // the next pair's Q fragments are waited for HERE, before this pair's stores enter the memory queue: they were requested
            // a whole compute phase ago, whereas a wait at their first use (next iteration) would also drain the stores just issued
            if (wave == 0) PP_STAMP(c, 5);
            land_q();
            if (wave == 0) PP_STAMP(c, 6);
            // tiles 2e and 2e+1 of a plane: v_permlane16_swap hands the odd lane groups' tile-2e runs to the even groups and the even
            // groups' tile-(2e+1) runs to the odd ones → every lane holds 8 consecutive head columns: 16·(2e + (g & 1)) + 8·(g >> 1)
            __bf16* Op = (__bf16*)a.O + (size_t)(q_off + q) * a.ldo + h * 64 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                uint32_t xh[2], yh[2], xl[2], yl[2];
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const float v0 = acc[2 * e + d][2 * r2] * oscale, v1 = acc[2 * e + d][2 * r2 + 1] * oscale;
                        const uint32_t hh = pp_cvt2(v0, v1);
                        (d ? yh : xh)[r2] = hh;
                        if (X3) (d ? yl : xl)[r2] = pp_cvt2(v0 - __uint_as_float(hh << 16), v1 - __uint_as_float(hh & 0xffff0000u));
                    }
#pragma unroll
                for (int plane = 0; plane < (X3 ? 2 : 1); ++plane) {
                    auto s0 = __builtin_amdgcn_permlane16_swap(plane ? xl[0] : xh[0], plane ? yl[0] : yh[0], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(plane ? xl[1] : xh[1], plane ? yl[1] : yh[1], false, false);
                    if (q < q_len && !(xa.dbg & 4))
                        *reinterpret_cast<uint4*>(Op + (plane ? xa.o_lo : 0) + 32 * e) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
            if (wave == 0) PP_STAMP(c, 7);
        }
    }
    if (do_stamp && wave == 0) {           // (the loader's last stamps precede its last barrier arrival or follow it by a few instructions)
        __builtin_amdgcn_s_sleep(100);
        for (int i = lane; i < 128; i += 64) xa.dbg_buf[bid * 128 + i] = stamps[i];
    }
#undef PP_STAMP
}

// ---------------------------------------------------------------------------------------------------------------- backward
// dQ, dK, dV of the same attention core for the bf16 streams (in bf16x3 mode: the hi planes), one launch, no atomics, no HBM
// intermediates — the arithmetic of attn_mfma_bwd_kernel (P recomputed from the forward's LSE and the same dropout draws) on the
// pipelined skeleton: persistent workgroups, a LOADER wave bringing the Q, K, V, dO images of the next pair into a two-stage LDS ring by
// LDS-DMA (plus the raw key mask and LSE rows, which it turns into the per-pair tables: mask terms and LSE in the exp2 domain, the
// dropout row hashes), seven COMPUTE waves on 16-row tiles:
//   pass 1, wave w = KEY tile w (its K and V fragments stay in registers for the pair): per query tile Sᵀ-layout products
//     S[q][k] = Q·Kᵀ, dP̃[q][k] = dO·Vᵀ (a key on a lane column, queries over lane groups and registers), P̃ = P ⊙ M and
//     dS = P ⊙ (dP̃ ⊙ M − δ) in registers; two query tiles at a time are the B operands of dVᵀ += dOᵀ·P̃ and dKᵀ += Qᵀ·dS (A operands:
//     transposed reads of the dO / Q images); dS also goes to an LDS image [key][query];
//   pass 2, wave w = QUERY tile w: dQᵀ += Kᵀ·dSᵀ with dSᵀ read back through ds_read_b64_tr_b16 (224-byte rows: conflict-free).
// δ = rowsum(dO ⊙ O) is computed by the compute waves one pair ahead from rows they request themselves (16 rows per wave).
// Two workgroup barriers per pair (ring hand-over; dS image complete).  Bound: HBM — (3·Lq + 2·Lk read + Lq + 2·Lk written)·64·2 B per pair.
constexpr int PB_NSTAGE = 2;
constexpr int PB_TAB = 4 * 512;                   // per stage: mask terms (keys), LSE·log2e, δ, dropout row hashes (queries): 128 words each
constexpr int PB_STAGE = 4 * PP_IMG + PB_TAB;     // Q, K, V, dO images + tables
// column offset (bytes, multiple of 8, < 224) of a dS-image row, rotated by 16 bytes for rows whose index has bit 3 set
__device__ __forceinline__ int pb_ds_rot(int off, int row) {
    const int o = off + ((row & 8) ? 16 : 0);
    return o >= 224 ? o - 224 : o;
}
constexpr int PB_DS_RS = 224;                     // dS image row: 112 queries bf16; 224 = 7·32 → the eight rows of a transposed read
constexpr int PB_DS = 112 * PB_DS_RS;             // fall on eight different 32-byte bank groups

__global__ __launch_bounds__(512) void attn_pipe_bwd_kernel(MAttnArgs a, int n_pairs) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NI = 4 * PP_NBLK + 4;           // DMA instructions per pair: four images, key mask, LSE
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x, G = gridDim.x;
    const int n_my = bid < n_pairs ? (n_pairs - bid + G - 1) / G : 0;
    char* const ds_img = smem + PB_NSTAGE * PB_STAGE;
    int m_qoff = 0, m_qlen = 0, m_koff = 0, m_klen = 0;      // (see the forward: segment-table entries by v_readlane)
    if (lane < n_my) {
        const int s = (bid + lane * G) / a.H;
        m_qoff = a.seq[s]; m_qlen = a.seq[a.n_seq + s]; m_koff = a.seq[2 * a.n_seq + s]; m_klen = a.seq[3 * a.n_seq + s];
    }
    const DropCtx dctx(a.seed, a.site, a.p_drop);
    asm volatile("" : "+v"(m_qoff), "+v"(m_qlen), "+v"(m_koff), "+v"(m_klen));
    (void)NI;

    if (wave == 7) {
        // ------------------------------------------------------------------------------------------------ loader
        __builtin_amdgcn_s_setprio(2);
        const int lrow = lane >> 3, ch0 = (lane & 7) ^ pp_swz(lrow);
        auto issue = [&](int j) {
            const int pair = bid + j * G, s = pair / a.H, h = pair - s * a.H;
            const int q_off = __builtin_amdgcn_readlane(m_qoff, j), q_len = __builtin_amdgcn_readlane(m_qlen, j);
            const int k_off = __builtin_amdgcn_readlane(m_koff, j), k_len = __builtin_amdgcn_readlane(m_klen, j);
            char* const st = smem + (j % PB_NSTAGE) * PB_STAGE;
            const char* const Qp = reinterpret_cast<const char*>((const __bf16*)a.Q + (size_t)q_off * a.ldq + h * 64);
            const char* const Kp = reinterpret_cast<const char*>((const __bf16*)a.K + (size_t)k_off * a.ldk + h * 64);
            const char* const Vp = reinterpret_cast<const char*>((const __bf16*)a.V + (size_t)k_off * a.ldv + h * 64);
            const char* const Dp = reinterpret_cast<const char*>((const __bf16*)a.dO + (size_t)q_off * a.lddo + h * 64);
#pragma unroll 1
            for (int b = 0; b < PP_NBLK; ++b) {
                const int rq = max(min(8 * b + lrow, q_len - 1), 0), rk = max(min(8 * b + lrow, k_len - 1), 0), ch = ch0 ^ ((b & 1) << 1);
                char* const d = st + b * 1024;
                __builtin_amdgcn_global_load_lds((pp_gptr)(Qp + ((unsigned)rq * (unsigned)a.ldq + 8u * ch) * 2u), (pp_lptr)d, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((pp_gptr)(Kp + ((unsigned)rk * (unsigned)a.ldk + 8u * ch) * 2u), (pp_lptr)(d + PP_IMG), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((pp_gptr)(Vp + ((unsigned)rk * (unsigned)a.ldv + 8u * ch) * 2u), (pp_lptr)(d + 2 * PP_IMG), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((pp_gptr)(Dp + ((unsigned)rq * (unsigned)a.lddo + 8u * ch) * 2u), (pp_lptr)(d + 3 * PP_IMG), 16, 0, 0);
            }
            char* const tab = st + 4 * PP_IMG;       // raw key mask → [0, 512), raw LSE → [512, 1024)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* km = a.key_mask ? a.key_mask + k_off + max(min(lane + 64 * i, k_len - 1), 0) : reinterpret_cast<const float*>(a.seq);
                __builtin_amdgcn_global_load_lds((pp_gptr)km, (pp_lptr)(tab + 256 * i), 4, 0, 0);
                const float* ls = a.LSE + ((size_t)s * a.H + h) * a.max_q + max(min(lane + 64 * i, q_len - 1), 0);
                __builtin_amdgcn_global_load_lds((pp_gptr)ls, (pp_lptr)(tab + 512 + 256 * i), 4, 0, 0);
            }
        };
        if (n_my > 0) issue(0);
        for (int c = 0; c < n_my; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // pair c is in LDS (two stages: nothing younger is in flight)
            {
                const int pair = bid + c * G, s = pair / a.H, h = pair - s * a.H;
                const int q_len = __builtin_amdgcn_readlane(m_qlen, c), k_len = __builtin_amdgcn_readlane(m_klen, c);
                float* const tab = reinterpret_cast<float*>(smem + (c % PB_NSTAGE) * PB_STAGE + 4 * PP_IMG);
                const u64 row0 = (u64)(s * a.H + h) * a.max_q;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int x = lane + 64 * i;
                    const float km = a.key_mask ? tab[x] : 1.0f, ls = tab[128 + x];
                    tab[x] = x < k_len ? (1.0f - km) * (-10000.0f * LOG2E) : -INFINITY;     // mask terms, exp2 domain
                    tab[128 + x] = x < q_len ? ls * LOG2E : INFINITY;                        // queries past the sequence: P = 0
                    reinterpret_cast<uint32_t*>(tab)[384 + x] = dctx.row(row0 + x);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (a raw s_barrier does not wait for this wave's LDS stores)
            __builtin_amdgcn_s_barrier();                    // A_c: pair c and its tables are in LDS; the other stage is free
            if (c + 1 < n_my) issue(c + 1);
            __builtin_amdgcn_s_barrier();                    // M_c (the compute waves' mid-pair barrier)
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute waves
    const int g = lane >> 4, l15 = lane & 15, t0 = 16 * wave;          // this wave's tile: keys t0 … in pass 1, queries t0 … in pass 2
    // lane parts of the fragment addresses (see the forward): row fragments of a 16-row tile t (rows 16·t + l15, chunk 4·ks + g) and
    // transposed fragments (rows 32·u + 16·ab + 4·g + q, chunk 2·dt + (p >> 1)); `…6` / `…3`: the last tile's rows clamped to the image
    const int kb = l15 * PP_RB + ((g ^ pp_swz(l15)) << 4);
    const int kr6 = min(96 + l15, PP_ROWS - 1), kb6 = kr6 * PP_RB + ((g ^ pp_swz(kr6)) << 4);
    const int krw = min(t0 + l15, PP_ROWS - 1), kbw = krw * PP_RB + ((g ^ pp_swz(krw)) << 4);      // this wave's own tile
    const int vq = l15 >> 2, vp = l15 & 3, vr = 4 * g + vq;
    const int vb = vr * PP_RB + (((vp >> 1) ^ pp_swz(vr)) << 4) + 8 * (vp & 1);
    const int vr3 = min(96 + vr, PP_ROWS - 1), vb3 = vr3 * PP_RB + (((vp >> 1) ^ pp_swz(vr3)) << 4) + 8 * (vp & 1);
    // dSᵀ fragment of pass 2: rows 4·g + q, columns t0 + 4·p.  Rows with bit 3 set keep their columns rotated by 16 bytes (pb_ds_rot):
    // with 224-byte rows the keys l and l + 8 of a pass-1 `ds_write_b64` met in the same banks (8·56 dwords = 7·64: SQ_LDS_BANK_CONFLICT 15 %
    // of the LDS cycles in round 4); rotated, the 16 rows × 2 lane groups of a half-wave tile the 64 banks exactly, and the transposed
    // reads (8 rows × 4 chunks per half-wave) stay conflict-free — rows 8..15 just sit 4 banks further on.
    const int dsb = vr * PB_DS_RS + pb_ds_rot(8 * vp + 2 * t0, vr);
    const float cs = a.scale * LOG2E;
    const uint32_t kphi = (uint32_t)(t0 + l15) * SVPC_ATTN_PHI;
    // δ one pair ahead: this wave's 16 query rows, 16 head columns per lane group
    bf16x8 no_[2], nd_[2];
    auto load_od = [&](int j) {
        const int pair = bid + j * G, s = pair / a.H, h = pair - s * a.H;
        const int q_off = __builtin_amdgcn_readlane(m_qoff, j), q_len = __builtin_amdgcn_readlane(m_qlen, j);
        const int r = max(min(t0 + l15, q_len - 1), 0);
        const __bf16* op = (const __bf16*)a.O + (size_t)(q_off + r) * a.ldo + h * 64 + 16 * g;
        const __bf16* dp = (const __bf16*)a.dO + (size_t)(q_off + r) * a.lddo + h * 64 + 16 * g;
#pragma unroll
        for (int i = 0; i < 2; ++i) { no_[i] = *reinterpret_cast<const bf16x8*>(op + 8 * i); nd_[i] = *reinterpret_cast<const bf16x8*>(dp + 8 * i); }
    };
    auto land_od = [&]() {                         // the rows are waited for BEFORE a pair's stores enter the memory queue (see the forward)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(no_[i]), "+v"(nd_[i]));
    };
    auto put_delta = [&](int j) {                  // δ of pair j into its stage's table
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf((float)no_[i][e], (float)nd_[i][e], d);
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        if (g == 0) reinterpret_cast<float*>(smem + (j % PB_NSTAGE) * PB_STAGE + 4 * PP_IMG)[256 + t0 + l15] = d;
    };
    if (n_my > 0) { load_od(0); land_od(); put_delta(0); }
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const short4v z4 = {0, 0, 0, 0};
    for (int c = 0; c < n_my; ++c) {
        const int pair = bid + c * G, s = pair / a.H, h = pair - s * a.H;
        const int q_off = __builtin_amdgcn_readlane(m_qoff, c), q_len = __builtin_amdgcn_readlane(m_qlen, c);
        const int k_off = __builtin_amdgcn_readlane(m_koff, c), k_len = __builtin_amdgcn_readlane(m_klen, c);
        load_od(min(c + 1, n_my - 1));              // (unconditional: see the forward's Q request)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's δ entries are in LDS
        __builtin_amdgcn_s_barrier();               // A_c
        const int soff = (c % PB_NSTAGE) * PB_STAGE;
        const int nkt = (k_len + 15) >> 4, nqt = (q_len + 15) >> 4;      // live 16-row tiles (≤ 7)
        int qi0 = soff + kb, qi1 = soff + (kb ^ 64), qi60 = soff + kb6, qi61 = soff + (kb6 ^ 64), wi0 = soff + kbw, ti = soff + 4 * PP_IMG + 16 * g;
        asm volatile("" : "+v"(qi0), "+v"(qi1), "+v"(qi60), "+v"(qi61), "+v"(wi0), "+v"(ti));
        const char* const tabq = smem + ti;          // + 512: LSE', + 1024: δ, + 1536: row hashes — of queries 4·g …
        // ---------------------------------------------------------------- pass 1: key tile `wave`
        if (wave < nkt) {
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[ks] = *reinterpret_cast<const bf16x8*>(smem + (ks ? wi0 ^ 64 : wi0) + PP_IMG);
                vf[ks] = *reinterpret_cast<const bf16x8*>(smem + (ks ? wi0 ^ 64 : wi0) + 2 * PP_IMG);
            }
            const float mt = reinterpret_cast<const float*>(smem + soff + 4 * PP_IMG)[t0 + l15];
            const int key = t0 + l15;
            floatx4 dv[4], dk[4];
            const char* ta[4]; const char* ta3[4];          // transposed fragments of the Q image (dO: + 3·PP_IMG)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                int i0 = soff + (vb ^ (dt << 5)), i3 = soff + (vb3 ^ (dt << 5));
                asm volatile("" : "+v"(i0), "+v"(i3));
                ta[dt] = smem + i0; ta3[dt] = smem + i3;
            }
            char* const dsw = ds_img + key * PB_DS_RS;                  // dS image: row = key, four consecutive queries per 8-byte store
            const int dsw_b3 = key & 8;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t pp_[4], dd_[4];                   // P̃ and dS of query tiles 2u, 2u+1 as one k-step's B fragment
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int qt = 2 * u + hf;
                    if (qt >= PP_NT) { pp_[2 * hf] = pp_[2 * hf + 1] = dd_[2 * hf] = dd_[2 * hf + 1] = 0u; continue; }
                    floatx4 sc, dp;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const char* const o = smem + (qt < 6 ? (ks ? qi1 : qi0) + 2048 * qt : (ks ? qi61 : qi60));
                        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(o), df = *reinterpret_cast<const bf16x8*>(o + 3 * PP_IMG);
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], ks ? sc : zero4, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vf[ks], ks ? dp : zero4, 0, 0, 0);
                    }
                    // register r ↔ query 16·qt + 4·g + r of this lane's key
                    const float4 l4 = *reinterpret_cast<const float4*>(tabq + 512 + 64 * qt), d4 = *reinterpret_cast<const float4*>(tabq + 1024 + 64 * qt);
                    const uint4 a4 = *reinterpret_cast<const uint4*>(tabq + 1536 + 64 * qt);
                    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
                    const uint32_t ar[4] = {a4.x, a4.y, a4.z, a4.w};
                    float pt[4], dsv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sc[r], cs, mt) - lq[r]);
                        float m = 1.0f;
                        if (a.p_drop > 0.f) m = svpc_attn_draw16(ar[r] + kphi) >= dctx.thr ? dctx.ik : 0.0f;
                        pt[r] = p * m;
                        dsv[r] = p * fmaf(dp[r], m, -dq[r]);
                    }
                    pp_[2 * hf] = pp_cvt2(pt[0], pt[1]); pp_[2 * hf + 1] = pp_cvt2(pt[2], pt[3]);
                    dd_[2 * hf] = pp_cvt2(dsv[0], dsv[1]); dd_[2 * hf + 1] = pp_cvt2(dsv[2], dsv[3]);
                    *reinterpret_cast<uint2*>(dsw + pb_ds_rot(8 * g + 32 * qt, dsw_b3)) = make_uint2(dd_[2 * hf], dd_[2 * hf + 1]);
                }
                const bf16x8 pf = pp_frag(pp_), df8 = pp_frag(dd_);
                bf16x8 qtr[4], dtr[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* const oa = u < 3 ? ta[dt] + 4096 * u : ta3[dt]; const char* const ob = ta[dt] + 4096 * u + 2048;
                    qtr[dt] = pp_join(__builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)oa), u < 3 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)ob) : z4);
                    dtr[dt] = pp_join(__builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(oa + 3 * PP_IMG)),
                                      u < 3 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(ob + 3 * PP_IMG)) : z4);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtr[dt], pf, u ? dv[dt] : zero4, 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], df8, u ? dk[dt] : zero4, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            land_od();
            // dV, dK rows of this lane's key: tiles 2e, 2e+1 paired by v_permlane16_swap → 8 consecutive head columns per lane
            __bf16* const vp_ = (__bf16*)a.dV + (size_t)(k_off + key) * a.lddv + h * 64 + 16 * (g & 1) + 8 * (g >> 1);
            __bf16* const kp_ = (__bf16*)a.dK + (size_t)(k_off + key) * a.lddk + h * 64 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t vx0 = pp_cvt2(dv[2 * e][0], dv[2 * e][1]), vx1 = pp_cvt2(dv[2 * e][2], dv[2 * e][3]);
                const uint32_t vy0 = pp_cvt2(dv[2 * e + 1][0], dv[2 * e + 1][1]), vy1 = pp_cvt2(dv[2 * e + 1][2], dv[2 * e + 1][3]);
                const uint32_t kx0 = pp_cvt2(dk[2 * e][0] * a.scale, dk[2 * e][1] * a.scale), kx1 = pp_cvt2(dk[2 * e][2] * a.scale, dk[2 * e][3] * a.scale);
                const uint32_t ky0 = pp_cvt2(dk[2 * e + 1][0] * a.scale, dk[2 * e + 1][1] * a.scale), ky1 = pp_cvt2(dk[2 * e + 1][2] * a.scale, dk[2 * e + 1][3] * a.scale);
                auto v0 = __builtin_amdgcn_permlane16_swap(vx0, vy0, false, false);
                auto v1 = __builtin_amdgcn_permlane16_swap(vx1, vy1, false, false);
                auto k0 = __builtin_amdgcn_permlane16_swap(kx0, ky0, false, false);
                auto k1 = __builtin_amdgcn_permlane16_swap(kx1, ky1, false, false);
                if (key < k_len) {
                    *reinterpret_cast<uint4*>(vp_ + 32 * e) = make_uint4(v0[0], v1[0], v0[1], v1[1]);
                    *reinterpret_cast<uint4*>(kp_ + 32 * e) = make_uint4(k0[0], k1[0], k0[1], k1[1]);
                }
            }
        }
        else land_od();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // M_c: the dS image is complete
        // ---------------------------------------------------------------- pass 2: query tile `wave`
        if (wave < nqt) {
            floatx4 dqa[4];
            const char* ka[4]; const char* ka3[4];          // transposed fragments of the K image
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                int i0 = soff + PP_IMG + (vb ^ (dt << 5)), i3 = soff + PP_IMG + (vb3 ^ (dt << 5));
                asm volatile("" : "+v"(i0), "+v"(i3));
                ka[dt] = smem + i0; ka3[dt] = smem + i3;
            }
            const char* const dsr = ds_img + dsb;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // keys 32·u … +15 (tile 2u) and +16 … +31 (tile 2u + 1); tiles no wave wrote this pair (past the sequence's keys) read as zeros
                const bool la = 2 * u < nkt, lb = 2 * u + 1 < nkt && u < 3;
                const bf16x8 sf = pp_join(la ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(dsr + 32 * u * PB_DS_RS)) : z4,
                                          lb ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)(dsr + (32 * u + 16) * PB_DS_RS)) : z4);
                bf16x8 ktr[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* const oa = u < 3 ? ka[dt] + 4096 * u : ka3[dt]; const char* const ob = ka[dt] + 4096 * u + 2048;
                    ktr[dt] = pp_join(__builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)oa), u < 3 ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_tr_ptr)ob) : z4);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr[dt], sf, u ? dqa[dt] : zero4, 0, 0, 0);
            }
            const int q = t0 + l15;
            __bf16* const qp_ = (__bf16*)a.dQ + (size_t)(q_off + q) * a.lddq + h * 64 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t x0 = pp_cvt2(dqa[2 * e][0] * a.scale, dqa[2 * e][1] * a.scale), x1 = pp_cvt2(dqa[2 * e][2] * a.scale, dqa[2 * e][3] * a.scale);
                const uint32_t y0 = pp_cvt2(dqa[2 * e + 1][0] * a.scale, dqa[2 * e + 1][1] * a.scale), y1 = pp_cvt2(dqa[2 * e + 1][2] * a.scale, dqa[2 * e + 1][3] * a.scale);
                auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                if (q < q_len) *reinterpret_cast<uint4*>(qp_ + 32 * e) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
        // δ of the next pair (its rows were requested before this pair's barrier), into the stage this pair's predecessor has left
        if (c + 1 < n_my) put_delta(c + 1);
    }
}

static int pp_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// 1 if the pipelined forward takes this problem (the callers fall back to the one-workgroup-per-pair kernels otherwise)
static int pp_dbg = 0;
static unsigned long long* pp_dbg_buf = nullptr;
// timing experiments (tools/dbg/pipe_*.py; never set by the product): bits 1 no LDS-DMA, 2 no arithmetic, 4 no O stores,
// 16 cycle stamps of workgroups 0-15 into `buf` (16 × 128 u64)
extern "C" int svpc_attn_pipe_debug(int bits, void* buf) { pp_dbg = bits; pp_dbg_buf = (unsigned long long*)buf; return 0; }
static int pp_on = -1;
// tuning / A-B switch: 1 = sequences of ≤ 104 rows take the pipelined forward (default; SVPC_ATTN_PIPE=0 in the environment turns it
// off), 0 = always the one-workgroup-per-pair kernels.  Returns the previous setting.
extern "C" int svpc_attn_pipe_enable(int on) {
    if (pp_on < 0) { const char* e = getenv("SVPC_ATTN_PIPE"); pp_on = e ? atoi(e) : 1; }
    const int was = pp_on;
    if (on >= 0) pp_on = on ? 1 : 0;
    return was;
}
bool attn_pipe_supported(const X3AttnArgs& xa, int dh, bool x3) {
    const int on = (svpc_attn_pipe_enable(-1), pp_on);
    const MAttnArgs& a = xa.m;
    const bool al = a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 &&
                    ((((uintptr_t)a.Q) | ((uintptr_t)a.K) | ((uintptr_t)a.V) | ((uintptr_t)a.O)) & 15) == 0 &&
                    (!x3 || (xa.q_lo % 8 == 0 && xa.k_lo % 8 == 0 && xa.v_lo % 8 == 0 && xa.o_lo % 8 == 0));
    return on && dh == 64 && !a.causal && a.max_q <= PP_ROWS && a.max_k <= PP_ROWS && a.max_q > 32 && al && a.n_seq * a.H <= 64 * pp_cus();
}
int attn_pipe_fwd_launch(const X3AttnArgs& xa_in, bool x3, hipStream_t stream) {
    X3AttnArgs xa = xa_in;
    xa.dbg = pp_dbg; xa.dbg_buf = pp_dbg_buf;
    if (!xa.dbg_buf) xa.dbg &= ~16;
    const int n_pairs = xa.m.n_seq * xa.m.H;
    if (x3) {
        const size_t lds = (size_t)PP_NSTAGE * (4 * PP_IMG + PP_MT) + 1024;
        int rc = svpc_raise_lds_once((const void*)attn_pipe_fwd_kernel<true>, "attn_pipe");
        if (rc) return rc;
        hipLaunchKernelGGL((attn_pipe_fwd_kernel<true>), dim3(min(n_pairs, pp_cus())), dim3(512), lds, stream, xa, n_pairs);
    } else {
        const size_t lds = (size_t)PP_NSTAGE * (2 * PP_IMG + PP_MT) + ((xa.dbg & 16) ? 1024 : 0);
        int rc = svpc_raise_lds_once((const void*)attn_pipe_fwd_kernel<false>, "attn_pipe");
        if (rc) return rc;
        hipLaunchKernelGGL((attn_pipe_fwd_kernel<false>), dim3(min(n_pairs, 2 * pp_cus())), dim3(512), lds, stream, xa, n_pairs);
    }
    return svpc_check_launch("attn_pipe_fwd");
}

// ---- backward
bool attn_pipe_bwd_supported(const MAttnArgs& a, int dh) {
    const int on = (svpc_attn_pipe_enable(-1), pp_on);
    const bool al = a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddo % 8 == 0 && a.lddq % 8 == 0 &&
                    a.lddk % 8 == 0 && a.lddv % 8 == 0 &&
                    ((((uintptr_t)a.Q) | ((uintptr_t)a.K) | ((uintptr_t)a.V) | ((uintptr_t)a.O) | ((uintptr_t)a.dO) | ((uintptr_t)a.dQ) |
                      ((uintptr_t)a.dK) | ((uintptr_t)a.dV)) & 15) == 0;
    return on && dh == 64 && !a.causal && a.max_q <= PP_ROWS && a.max_k <= PP_ROWS && a.max_q > 32 && al && a.LSE != nullptr &&
           a.n_seq * a.H <= 64 * pp_cus();
}
int attn_pipe_bwd_launch(const MAttnArgs& a, hipStream_t stream) {
    const int n_pairs = a.n_seq * a.H;
    const size_t lds = (size_t)PB_NSTAGE * PB_STAGE + PB_DS;
    int rc = svpc_raise_lds_once((const void*)attn_pipe_bwd_kernel, "attn_pipe_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_pipe_bwd_kernel, dim3(min(n_pairs, pp_cus())), dim3(512), lds, stream, a, n_pairs);
    return svpc_check_launch("attn_pipe_bwd");
}
