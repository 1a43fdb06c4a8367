// Pointer-generator, caption loss and Gumbel re-sampling kernels (reference: src/rtransformer/model.py:896-923
// pointer_generator_network, :37-55 LabelSmoothingLoss, :1018 gumbel_softmax(hard) @ word embeddings).
//
//  ptr_attn      per step j: score[t,e] = <Wing(B_j)[e], dec[j,t]>, pi = softmax_e, att[t] = Σ_e pi[t,e]·B_j[e]
//                (one workgroup per step: proj/bank rows come from L2, the Lt×E score matrix lives in LDS;
//                 backward needs no atomics because every reduction is local to the step)
//  ptr_mix_loss  per text row: P[:V] = g·softmax(logits); P[V:C] = 0; P[id] += (1-g)·pi[e]/|I_e| for the video's
//                (ingredient → word-id) CSR; label-smoothed KL with log(P+1e-12), last class gets no smoothing mass
//                (one workgroup per row, the row lives in LDS; duplicate ids resolved with LDS float adds)
//  gumbel        per text row: straight-through one-hot of argmax softmax((log(P+1e-12)+G)/tau) → embedding row gather
// All are latency/HBM-bound (rows ≤ 4,224 × ≤ ~1,000 columns).
#include "common.h"

constexpr int PTR_EMAX = 32;

// ------------------------------------------------------------------------------------------------ ptr_attn
// grid: T steps, 256 threads.  The Lt×D decoder rows (resp. their attended-vector gradients) and the entity rows, 16 at a time,
// are staged in LDS with all of a thread's loads in flight, so the Lt·E dot products read LDS (≈100-cycle latency) instead of
// paying one memory round trip each; the column-wise products (att, ddec, dproj, dbank) read every global row once, with the E
// (or Lt) values of a column held in registers.  No atomics: every reduction is local to the step.
constexpr int PTR_EC = 16;      // entity rows staged per pass
constexpr int PTR_LTMAX = 32;   // sentence length bound of the register-resident columns

__device__ __forceinline__ void ptr_stage(float* __restrict__ dst, const float* __restrict__ src, int n_floats) {
    // n_floats % 4 == 0, both 16-byte aligned; four float4 per thread in flight
    const int n4 = n_floats >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    const int NT = blockDim.x;
    int i = threadIdx.x;
    for (; i + 3 * NT < n4; i += 4 * NT) {
        const float4 a = s4[i], b = s4[i + NT], c = s4[i + 2 * NT], d = s4[i + 3 * NT];
        d4[i] = a; d4[i + NT] = b; d4[i + 2 * NT] = c; d4[i + 3 * NT] = d;
    }
    for (; i < n4; i += NT) d4[i] = s4[i];
}
// sc[t*32 + e0+e] = <rows_t[t], ent[e]> for t < lt, e < ec, both operands in LDS (one wave per dot, lanes over D)
__device__ __forceinline__ void ptr_dots(const float* __restrict__ rows_t, const float* __restrict__ ent, float* __restrict__ sc, int lt,
                                         int ec, int e0, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    for (int pe = wave; pe < lt * ec; pe += NW) {
        const int t = pe / ec, e = pe - t * ec;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += rows_t[(size_t)t * D + d] * ent[(size_t)e * D + d];
        dot = wave_sum(dot);
        if (lane == 0) sc[t * PTR_EMAX + e0 + e] = dot;
    }
}

// LDS: lt·D (decoder rows) + PTR_EC·D (entity chunk) + lt·32 (scores/pi) floats
// (NT = blockDim.x threads: 768 at D = 768 — one column per thread in the column phases, 12 waves on the row·entity dot products;
// one workgroup per step and one round of workgroups, so the time of a launch is the latency of ONE workgroup)
__global__ __launch_bounds__(768) void ptr_attn_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ proj,
                                                           const float* __restrict__ bank, const int* __restrict__ step_ne,
                                                           float* __restrict__ pi, float* __restrict__ att, int lt, int em, int D,
                                                           const float* __restrict__ pgen_w, const float* __restrict__ pgen_b,
                                                           float* __restrict__ pgen) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* rows = psm;                         // lt × D
    float* ent = rows + (size_t)lt * D;        // PTR_EC × D
    float* sc = ent + (size_t)PTR_EC * D;      // lt × PTR_EMAX
    const int j = blockIdx.x, E = step_ne[j];
    const float* pj = proj + (size_t)j * em * D;
    const float* bj = bank + (size_t)j * em * D;
    ptr_stage(rows, dec + (size_t)j * lt * D, lt * D);
    for (int e0 = 0; e0 < E; e0 += PTR_EC) {
        const int ec = min(PTR_EC, E - e0);
        __syncthreads();
        ptr_stage(ent, pj + (size_t)e0 * D, ec * D);
        __syncthreads();
        ptr_dots(rows, ent, sc, lt, ec, e0, D);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < lt; t += blockDim.x) {
        float m = -INFINITY;
        for (int e = 0; e < E; ++e) m = fmaxf(m, sc[t * PTR_EMAX + e]);
        float s = 0.f;
        for (int e = 0; e < E; ++e) { const float v = expf(sc[t * PTR_EMAX + e] - m); sc[t * PTR_EMAX + e] = v; s += v; }
        const float inv = 1.0f / s;
        for (int e = 0; e < em; ++e) {
            const float v = e < E ? sc[t * PTR_EMAX + e] * inv : 0.f;
            if (e < E) sc[t * PTR_EMAX + e] = v;
            pi[((size_t)j * lt + t) * em + e] = v;
        }
        for (int e = E; e < ((E + 3) & ~3); ++e) sc[t * PTR_EMAX + e] = 0.f;      // the 16-byte reads below cover whole quads
    }
    __syncthreads();
    // att[t][d] = Σ_e pi[t][e]·bank[e][d]: a thread owns column d, the E bank values of the column sit in registers
    float gp = 0.f;                            // (the gate is taken for lt = 1 only: host check)
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float bv[PTR_EMAX];
#pragma unroll
        for (int e = 0; e < PTR_EMAX; ++e) bv[e] = e < E ? bj[(size_t)e * D + d] : 0.f;
        for (int t = 0; t < lt; ++t) {       // (the pi row comes in 16-byte LDS reads: every thread reads the same E values)
            float acc = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < PTR_EMAX / 4; ++q4) {
                if (4 * q4 < E) {
                    const float4 p4 = *reinterpret_cast<const float4*>(sc + t * PTR_EMAX + 4 * q4);     // entries ≥ E meet bv = 0
                    acc += p4.x * bv[4 * q4] + p4.y * bv[4 * q4 + 1] + p4.z * bv[4 * q4 + 2] + p4.w * bv[4 * q4 + 3];
                }
            }
            if (att) att[((size_t)j * lt + t) * D + d] = acc;
            // the generation gate of the row, p_gen = sigmoid([dec ; att]·w + b) (model.py:905-908), while its operands are here: the
            // decoding iterations take it from this launch instead of a concatenation and a 1,536-deep one-column projection
            if (pgen_w) gp += rows[(size_t)t * D + d] * pgen_w[d] + acc * pgen_w[D + d];
        }
    }
    if (pgen_w) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
        const float ws = wave_sum(gp);
        __syncthreads();                       // every thread is done with the pi rows in `sc`: its first words take the wave partials
        float* gred = sc;
        if (lane == 0) gred[wave] = ws;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = gred[0];
            for (int w = 1; w < NW; ++w) tot += gred[w];
            pgen[j] = 1.0f / (1.0f + expf(-(tot + pgen_b[0])));
        }
    }
}

// dproj/dbank are (T, em, D) and fully written (zeros for e ≥ E); ddec (T*lt, D) fully written.
// LDS: lt·D (datt rows) + PTR_EC·D (entity chunk) + 2·lt·32 floats
__global__ __launch_bounds__(768) void ptr_attn_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ proj,
                                                           const float* __restrict__ bank, const int* __restrict__ step_ne,
                                                           const float* __restrict__ pi, const float* __restrict__ dpi,
                                                           const float* __restrict__ datt, float* __restrict__ ddec,
                                                           float* __restrict__ dproj, float* __restrict__ dbank, int lt, int em, int D) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* rows = psm;                         // lt × D : datt rows of this step
    float* ent = rows + (size_t)lt * D;        // PTR_EC × D
    float* dsc = ent + (size_t)PTR_EC * D;     // lt × PTR_EMAX
    float* pis = dsc + lt * PTR_EMAX;          // lt × PTR_EMAX
    float* dscT = pis + lt * PTR_EMAX;         // PTR_EMAX × PTR_LTMAX: the same two, entity-major (the column phase walks t in 16-byte reads)
    float* pisT = dscT + PTR_EMAX * PTR_LTMAX;
    const int j = blockIdx.x, E = step_ne[j];
    const float* pj = proj + (size_t)j * em * D;
    const float* bj = bank + (size_t)j * em * D;
    const float* dj = dec + (size_t)j * lt * D;
    ptr_stage(rows, datt + (size_t)j * lt * D, lt * D);
    for (int e0 = 0; e0 < E; e0 += PTR_EC) {
        const int ec = min(PTR_EC, E - e0);
        __syncthreads();
        ptr_stage(ent, bj + (size_t)e0 * D, ec * D);
        __syncthreads();
        ptr_dots(rows, ent, dsc, lt, ec, e0, D);      // datt[t]·bank[e]
    }
    __syncthreads();
    for (int i = threadIdx.x; i < lt * E; i += blockDim.x) {
        const int t = i / E, e = i - t * E;
        const size_t o = ((size_t)j * lt + t) * em + e;
        dsc[t * PTR_EMAX + e] += dpi ? dpi[o] : 0.f;
        pis[t * PTR_EMAX + e] = pi[o];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < lt; t += blockDim.x) {
        float mix = 0.f;
        for (int e = 0; e < E; ++e) mix += pis[t * PTR_EMAX + e] * dsc[t * PTR_EMAX + e];
        for (int e = 0; e < E; ++e) {
            const float v = pis[t * PTR_EMAX + e] * (dsc[t * PTR_EMAX + e] - mix);
            dsc[t * PTR_EMAX + e] = v;
            dscT[e * PTR_LTMAX + t] = v;
            pisT[e * PTR_LTMAX + t] = pis[t * PTR_EMAX + e];
        }
        for (int e = E; e < ((E + 3) & ~3); ++e) dsc[t * PTR_EMAX + e] = 0.f;      // the 16-byte reads below cover whole quads
    }
    for (int i = threadIdx.x; i < E * PTR_LTMAX; i += blockDim.x) {                  // … and whole quads of t in the transposed copies
        const int e = i / PTR_LTMAX, t = i - e * PTR_LTMAX;
        if (t >= lt) { dscT[e * PTR_LTMAX + t] = 0.f; pisT[e * PTR_LTMAX + t] = 0.f; }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        // ddec[t][d] = Σ_e dsc[t][e]·proj[e][d]
        float pv[PTR_EMAX];
#pragma unroll
        for (int e = 0; e < PTR_EMAX; ++e) pv[e] = e < E ? pj[(size_t)e * D + d] : 0.f;
        // dproj[e][d] = Σ_t dsc[t][e]·dec[t][d],  dbank[e][d] = Σ_t pi[t][e]·datt[t][d]: the Lt values of the column in registers
        float dv[PTR_LTMAX], av[PTR_LTMAX];
#pragma unroll
        for (int t = 0; t < PTR_LTMAX; ++t) {
            dv[t] = t < lt ? dj[(size_t)t * D + d] : 0.f;
            av[t] = t < lt ? rows[(size_t)t * D + d] : 0.f;
        }
        for (int t = 0; t < lt; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < PTR_EMAX / 4; ++q4) {
                if (4 * q4 < E) {
                    const float4 p4 = *reinterpret_cast<const float4*>(dsc + t * PTR_EMAX + 4 * q4);    // entries ≥ E meet pv = 0
                    acc += p4.x * pv[4 * q4] + p4.y * pv[4 * q4 + 1] + p4.z * pv[4 * q4 + 2] + p4.w * pv[4 * q4 + 3];
                }
            }
            ddec[((size_t)j * lt + t) * D + d] = acc;
        }
        for (int e = 0; e < em; ++e) {
            float ap = 0.f, ab = 0.f;
            if (e < E) {
#pragma unroll
                for (int q4 = 0; q4 < PTR_LTMAX / 4; ++q4) {
                    if (4 * q4 < lt) {          // (dv / av are zero for t ≥ lt, the copies are zero there too)
                        const float4 d4 = *reinterpret_cast<const float4*>(dscT + e * PTR_LTMAX + 4 * q4);
                        const float4 p4 = *reinterpret_cast<const float4*>(pisT + e * PTR_LTMAX + 4 * q4);
                        ap += d4.x * dv[4 * q4] + d4.y * dv[4 * q4 + 1] + d4.z * dv[4 * q4 + 2] + d4.w * dv[4 * q4 + 3];
                        ab += p4.x * av[4 * q4] + p4.y * av[4 * q4 + 1] + p4.z * av[4 * q4 + 2] + p4.w * av[4 * q4 + 3];
                    }
                }
            }
            dproj[((size_t)j * em + e) * D + d] = ap;
            dbank[((size_t)j * em + e) * D + d] = ab;
        }
    }
}


// ------------------------------------------------------------------------------------------------ ptr_attn with the generation gate (training)
// The pointer attention of a step together with the generation gate of its lt rows, p_gen[t] = sigmoid([dec_t ; att_t]·w + b)
// (reference: src/rtransformer/model.py:899-908).  In the training forward the attended vector att feeds NOTHING but that gate, so it never
// goes to HBM: forward writes pi and p_gen; backward takes d pi and d p_gen and uses
//     datt_t = dz_t·w2 (dz_t = dg_t·g_t(1-g_t), w2 = w[D:])   ⇒   datt_t·bank_e = dz_t·<w2, bank_e>: E dot products instead of lt·E,
//     dbank_e = (Σ_t pi[t,e]·dz_t)·w2 + [softmax path through proj: none — proj has its own gradient],
//     dw = Σ_t dz_t·[dec_t ; att_t] with Σ_t dz_t·att_t = Σ_e c_e·bank_e, c_e = Σ_t dz_t·pi[t,e];  db = Σ_t dz_t
// (per-step partial sums of dw | db go to `wpart` (T × (2D+1)), added up by the table-driven finalizer).  Replaces, per step of the
// model, the concatenation [dec ; att] (16 MB), a 1,536-deep one-column projection and its dgrad / wgrad / bias-sum launches.
__global__ __launch_bounds__(768) void ptr_attn_gate_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ proj,
                                                                const float* __restrict__ bank, const int* __restrict__ step_ne,
                                                                float* __restrict__ pi, int lt, int em, int D,
                                                                const float* __restrict__ pgen_w, const float* __restrict__ pgen_b,
                                                                float* __restrict__ pgen, const int* __restrict__ row_off,
                                                                const int* __restrict__ row_len) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* rows = psm;                         // lt × D
    float* ent = rows + (size_t)lt * D;        // PTR_EC × D
    float* sc = ent + (size_t)PTR_EC * D;      // lt × PTR_EMAX
    float* gred = sc + lt * PTR_EMAX;          // (waves) × PTR_LTMAX partial gate sums
    const int j = blockIdx.x, E = step_ne[j];
    // ragged sentences (the valid tokens only): sentence j owns rows [row_off[j], row_off[j] + row_len[j]); `lt` stays the padded length
    // (the LDS carve above); null tables: rows [j·lt, (j+1)·lt)
    const int roff = row_off ? row_off[j] : j * lt;
    const int ltj = row_len ? row_len[j] : lt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const float* pj = proj + (size_t)j * em * D;
    const float* bj = bank + (size_t)j * em * D;
    ptr_stage(rows, dec + (size_t)roff * D, ltj * D);
    for (int i = threadIdx.x; i < NW * PTR_LTMAX; i += blockDim.x) gred[i] = 0.f;
    for (int e0 = 0; e0 < E; e0 += PTR_EC) {
        const int ec = min(PTR_EC, E - e0);
        __syncthreads();
        ptr_stage(ent, pj + (size_t)e0 * D, ec * D);
        __syncthreads();
        ptr_dots(rows, ent, sc, ltj, ec, e0, D);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ltj; t += blockDim.x) {
        float m = -INFINITY;
        for (int e = 0; e < E; ++e) m = fmaxf(m, sc[t * PTR_EMAX + e]);
        float s = 0.f;
        for (int e = 0; e < E; ++e) { const float v = expf(sc[t * PTR_EMAX + e] - m); sc[t * PTR_EMAX + e] = v; s += v; }
        const float inv = 1.0f / s;
        for (int e = 0; e < em; ++e) {
            const float v = e < E ? sc[t * PTR_EMAX + e] * inv : 0.f;
            if (e < E) sc[t * PTR_EMAX + e] = v;
            pi[((size_t)roff + t) * em + e] = v;
        }
        for (int e = E; e < ((E + 3) & ~3); ++e) sc[t * PTR_EMAX + e] = 0.f;
    }
    __syncthreads();
    // column phase (wave-uniform trip count: the per-row gate sums are wave reductions): a thread owns column d
    for (int d0 = 0; d0 < D; d0 += blockDim.x) {
        const int d = d0 + threadIdx.x;
        const bool on = d < D;
        const int dc = on ? d : D - 1;
        float bv[PTR_EMAX];
#pragma unroll
        for (int e = 0; e < PTR_EMAX; ++e) bv[e] = (e < E && on) ? bj[(size_t)e * D + dc] : 0.f;
        const float w1 = on ? pgen_w[dc] : 0.f, w2 = on ? pgen_w[D + dc] : 0.f;
        for (int t = 0; t < ltj; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < PTR_EMAX / 4; ++q4) {
                if (4 * q4 < E) {
                    const float4 p4 = *reinterpret_cast<const float4*>(sc + t * PTR_EMAX + 4 * q4);
                    acc += p4.x * bv[4 * q4] + p4.y * bv[4 * q4 + 1] + p4.z * bv[4 * q4 + 2] + p4.w * bv[4 * q4 + 3];
                }
            }
            const float gs = wave_sum(rows[(size_t)t * D + dc] * w1 + acc * w2);
            if (lane == 0) gred[wave * PTR_LTMAX + t] += gs;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ltj; t += blockDim.x) {
        float tot = pgen_b[0];
        for (int w = 0; w < NW; ++w) tot += gred[w * PTR_LTMAX + t];
        pgen[(size_t)roff + t] = 1.0f / (1.0f + expf(-tot));
    }
}

// ---- the same forward, rows in registers (round 5).  The kernel above stages the sentence's rows and an entity chunk in LDS, evaluates the
// lt·E scores one WAVE per dot product (two LDS reads per multiply: 1.35 MB of LDS traffic per sentence) and then walks the rows a second
// time per column for the gate.  The gate's argument needs no attended vector at all:
//     [dec_t ; att_t]·w = dec_t·w1 + Σ_e pi[t,e]·<bank_e, w2>      (w = [w1 | w2])
// — E dot products u_e = <bank_e, w2> per sentence and one more dot product per row.  So: the projected entity rows in LDS once, u_e by
// one wave per entity, and every wave takes rows t = wave, wave + NW, …: the row straight from HBM into registers (KPL values per lane),
// its E + 1 dot products (proj_e from LDS, w1 from registers) reduced by ONE reduce-scatter butterfly; a thread per row then does the
// softmax over the E scores and the gate.  D = 64·KPL ≤ 768, E ≤ 31 (value slot 31 carries dec_t·w1); other shapes: the kernel above.
template <int KPL>
__global__ __launch_bounds__(64 * KPL) void ptr_attn_gate_fwd_rows_kernel(const float* __restrict__ dec, const float* __restrict__ proj,
                                                                          const float* __restrict__ bank, const int* __restrict__ step_ne,
                                                                          float* __restrict__ pi, int lt, int em, const float* __restrict__ pgen_w,
                                                                          const float* __restrict__ pgen_b, float* __restrict__ pgen,
                                                                          const int* __restrict__ row_off, const int* __restrict__ row_len) {
    constexpr int D = 64 * KPL, NW = KPL;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* ent = psm;                          // E × D projected entity rows
    float* sc = ent + (size_t)em * D;          // PTR_LTMAX × 32: scores (slot 31: dec_t·w1)
    float* ue = sc + PTR_LTMAX * 32;           // 32: <bank_e, w2>
    const int j = blockIdx.x, E = step_ne[j];
    const int roff = row_off ? row_off[j] : j * lt;
    const int ltj = row_len ? row_len[j] : lt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* pj = proj + (size_t)j * em * D;
    const float* bj = bank + (size_t)j * em * D;
    // this wave's first row and the gate's two weight rows are requested first; the entity rows go to LDS
    float w1[KPL], w2[KPL], row[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) { w1[k] = pgen_w[lane + 64 * k]; w2[k] = pgen_w[D + lane + 64 * k]; }
    {
        const float* r0 = dec + (size_t)(roff + min(wave, ltj - 1)) * D;
#pragma unroll
        for (int k = 0; k < KPL; ++k) row[k] = r0[lane + 64 * k];
    }
    ptr_stage(ent, pj, E * D);
    for (int e = wave; e < E; e += NW) {       // u_e = <bank_e, w2>
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KPL; ++k) dot = fmaf(bj[(size_t)e * D + lane + 64 * k], w2[k], dot);
        dot = wave_sum(dot);
        if (lane == 0) ue[e] = dot;
    }
    __syncthreads();
    for (int t = wave; t < ltj; t += NW) {
        float part[32];
#pragma unroll
        for (int e = 0; e < 31; ++e) {
            float d_ = 0.f;
            if (e < E) {
#pragma unroll
                for (int k = 0; k < KPL; ++k) d_ = fmaf(row[k], ent[(size_t)e * D + lane + 64 * k], d_);
            }
            part[e] = d_;
        }
        float dw = 0.f;
#pragma unroll
        for (int k = 0; k < KPL; ++k) dw = fmaf(row[k], w1[k], dw);
        part[31] = dw;
        if (t + NW < ltj) {                    // the wave's next row lands under the reduction
            const float* rn = dec + (size_t)(roff + t + NW) * D;
#pragma unroll
            for (int k = 0; k < KPL; ++k) row[k] = rn[lane + 64 * k];
        }
        const float v = wave_reduce_scatter32(part, lane);
        if ((lane & 1) == 0) sc[t * 32 + (lane >> 1)] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ltj; t += blockDim.x) {
        float m = -INFINITY;
        for (int e = 0; e < E; ++e) m = fmaxf(m, sc[t * 32 + e]);
        float s_ = 0.f;
        for (int e = 0; e < E; ++e) { const float v = expf(sc[t * 32 + e] - m); sc[t * 32 + e] = v; s_ += v; }
        const float inv = 1.0f / s_;
        float tot = pgen_b[0] + sc[t * 32 + 31];
        for (int e = 0; e < em; ++e) {
            const float v = e < E ? sc[t * 32 + e] * inv : 0.f;
            pi[((size_t)roff + t) * em + e] = v;
            if (e < E) tot = fmaf(v, ue[e], tot);
        }
        pgen[(size_t)roff + t] = 1.0f / (1.0f + expf(-tot));
    }
}

// LDS: (1 + PTR_EC)·D floats (w2 row, entity chunk) + 2·lt·32 + 2·32·32 + 3·32 floats
__global__ __launch_bounds__(768) void ptr_attn_gate_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ proj,
                                                                const float* __restrict__ bank, const int* __restrict__ step_ne,
                                                                const float* __restrict__ pi, const float* __restrict__ dpi,
                                                                const float* __restrict__ pgen, const float* __restrict__ dpgen,
                                                                const float* __restrict__ pgen_w, float* __restrict__ ddec,
                                                                float* __restrict__ dproj, float* __restrict__ dbank,
                                                                float* __restrict__ wpart, int lt, int em, int D,
                                                                const int* __restrict__ row_off, const int* __restrict__ row_len) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float* w2row = psm;                        // D: w[D:]
    float* ent = w2row + D;                    // PTR_EC × D
    float* dsc = ent + (size_t)PTR_EC * D;     // lt × PTR_EMAX
    float* pis = dsc + lt * PTR_EMAX;          // lt × PTR_EMAX
    float* dscT = pis + lt * PTR_EMAX;         // PTR_EMAX × PTR_LTMAX
    float* pisT = dscT + PTR_EMAX * PTR_LTMAX; // PTR_EMAX × PTR_LTMAX (kept for the layout's sake: c_e is taken from it)
    float* ue = pisT + PTR_EMAX * PTR_LTMAX;   // PTR_EMAX: <w2, bank_e>   (ptr_dots writes row 0 of a lt×32 image)
    float* dzs = ue + PTR_EMAX;                // PTR_LTMAX: dz_t
    float* ces = dzs + PTR_LTMAX;              // PTR_EMAX: c_e
    const int j = blockIdx.x, E = step_ne[j];
    const int roff = row_off ? row_off[j] : j * lt;          // (ragged sentences: see the forward; `lt` stays the padded length of the carve)
    const int ltj = row_len ? row_len[j] : lt;
    const float* pj = proj + (size_t)j * em * D;
    const float* bj = bank + (size_t)j * em * D;
    const float* dj = dec + (size_t)roff * D;
    ptr_stage(w2row, pgen_w + D, D);
    for (int t = threadIdx.x; t < PTR_LTMAX; t += blockDim.x) {
        float dz = 0.f;
        if (t < ltj) { const float g = pgen[(size_t)roff + t]; dz = (dpgen ? dpgen[(size_t)roff + t] : 0.f) * g * (1.0f - g); }
        dzs[t] = dz;
    }
    for (int e0 = 0; e0 < E; e0 += PTR_EC) {
        const int ec = min(PTR_EC, E - e0);
        __syncthreads();
        ptr_stage(ent, bj + (size_t)e0 * D, ec * D);
        __syncthreads();
        ptr_dots(w2row, ent, ue, 1, ec, e0, D);          // u_e = <w2, bank_e>
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ltj * E; i += blockDim.x) {
        const int t = i / E, e = i - t * E;
        const size_t o = ((size_t)roff + t) * em + e;
        dsc[t * PTR_EMAX + e] = dzs[t] * ue[e] + (dpi ? dpi[o] : 0.f);
        pis[t * PTR_EMAX + e] = pi[o];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ltj; t += blockDim.x) {
        float mix = 0.f;
        for (int e = 0; e < E; ++e) mix += pis[t * PTR_EMAX + e] * dsc[t * PTR_EMAX + e];
        for (int e = 0; e < E; ++e) {
            const float v = pis[t * PTR_EMAX + e] * (dsc[t * PTR_EMAX + e] - mix);
            dsc[t * PTR_EMAX + e] = v;
            dscT[e * PTR_LTMAX + t] = v;
        }
        for (int e = E; e < ((E + 3) & ~3); ++e) dsc[t * PTR_EMAX + e] = 0.f;
    }
    for (int i = threadIdx.x; i < E * PTR_LTMAX; i += blockDim.x) {
        const int e = i / PTR_LTMAX, t = i - e * PTR_LTMAX;
        if (t >= ltj) dscT[e * PTR_LTMAX + t] = 0.f;
    }
    for (int e = threadIdx.x; e < PTR_EMAX; e += blockDim.x) {        // c_e = Σ_t dz_t·pi[t,e]
        float c = 0.f;
        if (e < E) for (int t = 0; t < ltj; ++t) c += dzs[t] * pis[t * PTR_EMAX + e];
        ces[e] = c;
    }
    __syncthreads();
    float* wp = wpart + (size_t)j * (2 * D + 1);
    if (threadIdx.x == 0) {
        float sb = 0.f;
        for (int t = 0; t < ltj; ++t) sb += dzs[t];
        wp[2 * D] = sb;
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float pv[PTR_EMAX];
#pragma unroll
        for (int e = 0; e < PTR_EMAX; ++e) pv[e] = e < E ? pj[(size_t)e * D + d] : 0.f;
        float dv[PTR_LTMAX];
#pragma unroll
        for (int t = 0; t < PTR_LTMAX; ++t) dv[t] = t < ltj ? dj[(size_t)t * D + d] : 0.f;
        const float w1 = pgen_w[d], w2 = w2row[d];
        // ddec[t][d] = Σ_e dsc[t][e]·proj[e][d] + dz_t·w1[d]
        for (int t = 0; t < ltj; ++t) {
            float acc = dzs[t] * w1;
#pragma unroll
            for (int q4 = 0; q4 < PTR_EMAX / 4; ++q4) {
                if (4 * q4 < E) {
                    const float4 p4 = *reinterpret_cast<const float4*>(dsc + t * PTR_EMAX + 4 * q4);
                    acc += p4.x * pv[4 * q4] + p4.y * pv[4 * q4 + 1] + p4.z * pv[4 * q4 + 2] + p4.w * pv[4 * q4 + 3];
                }
            }
            ddec[((size_t)roff + t) * D + d] = acc;
        }
        // dw1[d] partial = Σ_t dz_t·dec[t][d]
        float s1 = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < PTR_LTMAX / 4; ++q4) {
            if (4 * q4 < ltj) {
                const float4 z4 = *reinterpret_cast<const float4*>(dzs + 4 * q4);
                s1 += z4.x * dv[4 * q4] + z4.y * dv[4 * q4 + 1] + z4.z * dv[4 * q4 + 2] + z4.w * dv[4 * q4 + 3];
            }
        }
        wp[d] = s1;
        // dproj[e][d] = Σ_t dsc[t][e]·dec[t][d];  dbank[e][d] = c_e·w2[d];  dw2[d] partial = Σ_e c_e·bank[e][d]
        float s2 = 0.f;
        for (int e = 0; e < em; ++e) {
            float ap = 0.f, ab = 0.f;
            if (e < E) {
#pragma unroll
                for (int q4 = 0; q4 < PTR_LTMAX / 4; ++q4) {
                    if (4 * q4 < ltj) {
                        const float4 d4 = *reinterpret_cast<const float4*>(dscT + e * PTR_LTMAX + 4 * q4);
                        ap += d4.x * dv[4 * q4] + d4.y * dv[4 * q4 + 1] + d4.z * dv[4 * q4 + 2] + d4.w * dv[4 * q4 + 3];
                    }
                }
                const float c = ces[e];
                ab = c * w2;
                s2 += c * bj[(size_t)e * D + d];
            }
            dproj[((size_t)j * em + e) * D + d] = ap;
            dbank[((size_t)j * em + e) * D + d] = ab;
        }
        wp[D + d] = s2;
    }
}

// ------------------------------------------------------------------------------------------------ ptr_mix_loss
struct MixArgs {
    const float* logits; const float* g; const float* pi; const int* labels; const int* row_c; const int* row_vid;
    const int* csr_off; const int* csr_ent; const int* csr_id; const float* csr_w;
    float* P; float* loss_rows; int V; int c_max; int em; float smoothing;
    // backward
    const float* dP_ext; const float* dloss; float* dlogits; float* dg; float* dpi;
    // label_smoothing == 0 (reference model.py:869-870: nn.CrossEntropyLoss(ignore_index=-1) applied to the PROBABILITIES): row weight
    // 1 / (valid rows of the row's video) — the criterion is a mean per video; non-null selects that branch
    const float* row_w;
};

// row_w[r] = 1 / #{r' of r's video with label != -1}: one workgroup, counts in LDS (videos in chunks of 1024).  A video without a valid
// row gets 1/0 = inf on rows that are all ignored (weight never used); torch's mean over an empty selection is NaN there as well.
__global__ __launch_bounds__(1024) void ce_row_weights_kernel(const int* __restrict__ labels, const int* __restrict__ row_vid, int R,
                                                               int n_vid, float* __restrict__ row_w) {
    __shared__ int cnt[1024];
    for (int v0 = 0; v0 < n_vid; v0 += 1024) {
        cnt[threadIdx.x] = 0;
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += 1024) {
            const int b = row_vid[r] - v0;
            if (b >= 0 && b < 1024 && labels[r] >= 0) atomicAdd(&cnt[b], 1);
        }
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += 1024) {
            const int b = row_vid[r] - v0;
            if (b >= 0 && b < 1024) row_w[r] = 1.0f / (float)cnt[b];
        }
        __syncthreads();
    }
}

// grid: R rows. LDS: c_max floats.
__global__ __launch_bounds__(256) void ptr_mix_loss_fwd_kernel(MixArgs a) {
    extern __shared__ float row[];
    __shared__ float red[4];
    const int r = blockIdx.x, V = a.V, C = a.row_c[r];
    const float* lg = a.logits + (size_t)r * V;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, lg[v]);
    m = block_max_256(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) { const float e = expf(lg[v] - m); row[v] = e; s += e; }
    s = block_sum_256(s, red);
    const float gate = a.g ? a.g[r] : 1.0f;
    const float sc = gate / s;
    for (int v = threadIdx.x; v < a.c_max; v += 256) row[v] = v < V ? row[v] * sc : 0.f;
    __syncthreads();
    if (a.g) {
        const int b = a.row_vid[r];
        for (int n = a.csr_off[b] + threadIdx.x; n < a.csr_off[b + 1]; n += 256)
            atomicAdd(&row[a.csr_id[n]], (1.0f - gate) * a.pi[(size_t)r * a.em + a.csr_ent[n]] * a.csr_w[n]);
        __syncthreads();
    }
    const int y = a.labels[r];
    float loss = 0.f;
    if (a.row_w) {
        // cross-entropy OF THE PROBABILITIES (the reference's label_smoothing == 0 branch): w_r · (logsumexp_{v<C} P_v − P_y)
        float pm = -INFINITY;
        for (int v = threadIdx.x; v < a.c_max; v += 256) {
            const float p = row[v];
            a.P[(size_t)r * a.c_max + v] = p;
            if (v < C) pm = fmaxf(pm, p);
        }
        pm = block_max_256(pm, red);
        float se = 0.f;
        for (int v = threadIdx.x; v < C; v += 256) se += expf(row[v] - pm);
        se = block_sum_256(se, red);
        if (threadIdx.x == 0) a.loss_rows[r] = y >= 0 ? a.row_w[r] * (pm + logf(se) - row[y]) : 0.f;
        return;
    }
    const float qs = a.smoothing / (float)(C - 1), conf = 1.0f - a.smoothing;
    for (int v = threadIdx.x; v < a.c_max; v += 256) {
        const float p = row[v];
        a.P[(size_t)r * a.c_max + v] = p;
        if (y >= 0 && v < C) {
            const float qv = v == y ? conf : (v == C - 1 ? 0.f : qs);
            if (qv > 0.f) loss += qv * (logf(qv) - logf(p + 1e-12f));
        }
    }
    loss = block_sum_256(loss, red);
    if (threadIdx.x == 0) a.loss_rows[r] = loss;
}

// LDS: 2*c_max floats (dP row, softmax row)
__global__ __launch_bounds__(256) void ptr_mix_loss_bwd_kernel(MixArgs a) {
    extern __shared__ float sm[];
    __shared__ float red[4];
    __shared__ float dpis[PTR_EMAX];
    float* dP = sm;
    float* smx = sm + a.c_max;
    const int r = blockIdx.x, V = a.V, C = a.row_c[r];
    const int y = a.labels[r];
    const float go = a.dloss ? a.dloss[r] : 1.0f;
    const float qs = a.smoothing / (float)(C - 1), conf = 1.0f - a.smoothing;
    const float* lg = a.logits + (size_t)r * V;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, lg[v]);
    m = block_max_256(m, red);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) { const float e = expf(lg[v] - m); smx[v] = e; s += e; }
    s = block_sum_256(s, red);
    const float inv = 1.0f / s;
    float ce_m = 0.f, ce_inv = 0.f;
    if (a.row_w && y >= 0) {          // softmax of the PROBABILITIES over the row's C classes (see the forward)
        float pm = -INFINITY;
        for (int v = threadIdx.x; v < C; v += 256) pm = fmaxf(pm, a.P[(size_t)r * a.c_max + v]);
        pm = block_max_256(pm, red);
        float se = 0.f;
        for (int v = threadIdx.x; v < C; v += 256) se += expf(a.P[(size_t)r * a.c_max + v] - pm);
        se = block_sum_256(se, red);
        ce_m = pm; ce_inv = go * a.row_w[r] / se;
    }
    for (int v = threadIdx.x; v < a.c_max; v += 256) {
        float d = a.dP_ext ? a.dP_ext[(size_t)r * a.c_max + v] : 0.f;
        if (y >= 0 && v < C) {
            if (a.row_w) {
                d += ce_inv * expf(a.P[(size_t)r * a.c_max + v] - ce_m) - (v == y ? go * a.row_w[r] : 0.f);
            } else {
                const float qv = v == y ? conf : (v == C - 1 ? 0.f : qs);
                if (qv > 0.f) d -= go * qv / (a.P[(size_t)r * a.c_max + v] + 1e-12f);
            }
        }
        dP[v] = d;
        if (v < V) smx[v] *= inv;
    }
    __syncthreads();
    const float gate = a.g ? a.g[r] : 1.0f;
    float dot = 0.f;                      // Σ_v dP_v · sm_v
    for (int v = threadIdx.x; v < V; v += 256) dot += dP[v] * smx[v];
    dot = block_sum_256(dot, red);
    // d logits: sm_v · g · (dP_v - dot)
    for (int v = threadIdx.x; v < V; v += 256) a.dlogits[(size_t)r * V + v] = smx[v] * gate * (dP[v] - dot);
    if (a.g) {
        const int b = a.row_vid[r];
        if (threadIdx.x < PTR_EMAX) dpis[threadIdx.x] = 0.f;
        __syncthreads();
        float cp = 0.f;                   // Σ_n dP[id]·pi[e]·w
        for (int n = a.csr_off[b] + threadIdx.x; n < a.csr_off[b + 1]; n += 256) {
            const float t = dP[a.csr_id[n]] * a.csr_w[n];
            cp += t * a.pi[(size_t)r * a.em + a.csr_ent[n]];
            atomicAdd(&dpis[a.csr_ent[n]], (1.0f - gate) * t);
        }
        cp = block_sum_256(cp, red);
        if (threadIdx.x == 0) a.dg[r] = dot - cp;
        if (threadIdx.x < a.em) a.dpi[(size_t)r * a.em + threadIdx.x] = dpis[threadIdx.x];
    }
}

// ------------------------------------------------------------------------------------------------ gumbel
// fwd: idx[r] = argmax_j<C (log(P+eps)+G)/tau ; stats[r] = (max, sumexp) ; bow[r] = coef·emb[idx] (0 if idx ≥ V)
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const float* __restrict__ P, const float* __restrict__ noise,
                                                         const int* __restrict__ row_c, const float* __restrict__ emb,
                                                         float* __restrict__ bow, int* __restrict__ idx_out,
                                                         float* __restrict__ stats, int c_max, int V, int W, float inv_tau) {
    __shared__ float red[4];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int r = blockIdx.x, C = row_c[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < C; v += 256) {
        const float l = (logf(P[(size_t)r * c_max + v] + 1e-12f) + noise[(size_t)r * c_max + v]) * inv_tau;
        if (l > best) { best = l; bi = v; }
    }
    // wave argmax (first index on ties), then across the 4 waves
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    best = sval[0]; bi = sidx[0];
    for (int w = 1; w < 4; ++w)
        if (sval[w] > best || (sval[w] == best && sidx[w] < bi)) { best = sval[w]; bi = sidx[w]; }
    float s = 0.f;
    for (int v = threadIdx.x; v < C; v += 256)
        s += expf((logf(P[(size_t)r * c_max + v] + 1e-12f) + noise[(size_t)r * c_max + v]) * inv_tau - best);
    s = block_sum_256(s, red);
    const float ymax = 1.0f / s;                       // soft probability of the arg-max class
    const float coef = (1.0f - ymax) + ymax;           // hard - y.detach() + y, evaluated like the reference
    if (threadIdx.x == 0) { idx_out[r] = bi; stats[2 * r] = best; stats[2 * r + 1] = s; }
    for (int c = threadIdx.x; c < W; c += 256) bow[(size_t)r * W + c] = bi < V ? coef * emb[(size_t)bi * W + c] : 0.f;
}

// bwd: dy (R, V) = dbow @ embᵀ comes from the GEMM; dP_j = y_j (dy_j[j<V] - Σ_k<V y_k dy_k) / tau / (P_j + eps)
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const float* __restrict__ P, const float* __restrict__ noise,
                                                         const int* __restrict__ row_c, const float* __restrict__ stats,
                                                         const float* __restrict__ dy, float* __restrict__ dP, int c_max, int V,
                                                         float inv_tau) {
    __shared__ float red[4];
    const int r = blockIdx.x, C = row_c[r];
    const float mx = stats[2 * r], inv = 1.0f / stats[2 * r + 1];
    float dot = 0.f;
    for (int v = threadIdx.x; v < C && v < V; v += 256) {
        const float yv = expf((logf(P[(size_t)r * c_max + v] + 1e-12f) + noise[(size_t)r * c_max + v]) * inv_tau - mx) * inv;
        dot += yv * dy[(size_t)r * V + v];
    }
    dot = block_sum_256(dot, red);
    for (int v = threadIdx.x; v < c_max; v += 256) {
        float out = 0.f;
        if (v < C) {
            const float p = P[(size_t)r * c_max + v];
            const float yv = expf((logf(p + 1e-12f) + noise[(size_t)r * c_max + v]) * inv_tau - mx) * inv;
            const float d = v < V ? dy[(size_t)r * V + v] : 0.f;
            out = yv * (d - dot) * inv_tau / (p + 1e-12f);
        }
        dP[(size_t)r * c_max + v] = out;
    }
}
// The same two with ONE WAVE per row (c_max ≤ 64·NPL): the row's logits stay in registers — no second log / exp pass, no workgroup
// barrier, four rows per workgroup.  Same arithmetic per element; the row sums add in lane order instead of thread order.
template <int NPL>
__global__ __launch_bounds__(256) void gumbel_fwd_row_kernel(const float* __restrict__ P, const float* __restrict__ noise,
                                                             const int* __restrict__ row_c, const float* __restrict__ emb,
                                                             float* __restrict__ bow, int* __restrict__ idx_out,
                                                             float* __restrict__ stats, int R, int c_max, int V, int W, float inv_tau) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const int C = row_c[r];
    const float* Pr = P + (size_t)r * c_max;
    const float* Nr = noise + (size_t)r * c_max;
    float l[NPL];
    float best = -INFINITY; int bi = 0x7fffffff;
    // (all loads of the row first, with clamped indices: inside `v < C ? … : …` every load sat in a predicated region of its own with a
    // drain behind it — 2·NPL memory round trips per row: tools/isa_audit.py)
    float pr[NPL], nr[NPL];
    const int cl = max(C - 1, 0);
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const int vc = min(lane + 64 * i, cl); pr[i] = Pr[vc]; nr[i] = Nr[vc]; }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int v = lane + 64 * i;
        l[i] = v < C ? (logf(pr[i] + 1e-12f) + nr[i]) * inv_tau : -INFINITY;
        if (l[i] > best) { best = l[i]; bi = v; }           // ascending v: the first index wins a tie
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i)
        if (lane + 64 * i < C) s += expf(l[i] - best);
    s = wave_sum(s);
    const float ymax = 1.0f / s;                       // soft probability of the arg-max class
    const float coef = (1.0f - ymax) + ymax;           // hard - y.detach() + y, evaluated like the reference
    if (lane == 0) { idx_out[r] = bi; stats[2 * r] = best; stats[2 * r + 1] = s; }
    for (int c = lane; c < W; c += 64) bow[(size_t)r * W + c] = bi < V ? coef * emb[(size_t)bi * W + c] : 0.f;
}
template <int NPL>
__global__ __launch_bounds__(256) void gumbel_bwd_row_kernel(const float* __restrict__ P, const float* __restrict__ noise,
                                                             const int* __restrict__ row_c, const float* __restrict__ stats,
                                                             const float* __restrict__ dy, float* __restrict__ dP, int R, int c_max, int V,
                                                             float inv_tau) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const int C = row_c[r];
    const float mx = stats[2 * r], inv = 1.0f / stats[2 * r + 1];
    const float* Pr = P + (size_t)r * c_max;
    const float* Nr = noise + (size_t)r * c_max;
    float y[NPL], pe[NPL], d[NPL];
    float dot = 0.f;
    float pr[NPL], nr[NPL], dr[NPL];              // all loads of the row first, clamped indices (see gumbel_fwd_row_kernel)
    const int cl = max(C - 1, 0);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int vc = min(lane + 64 * i, cl);
        pr[i] = Pr[vc]; nr[i] = Nr[vc]; dr[i] = dy[(size_t)r * V + min(vc, V - 1)];
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int v = lane + 64 * i;
        y[i] = 0.f; pe[i] = 1.f; d[i] = 0.f;
        if (v < C) {
            const float p = pr[i];
            pe[i] = p + 1e-12f;
            y[i] = expf((logf(pe[i]) + nr[i]) * inv_tau - mx) * inv;
            d[i] = v < V ? dr[i] : 0.f;
            if (v < V) dot += y[i] * d[i];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int v = lane + 64 * i;
        if (v < c_max) dP[(size_t)r * c_max + v] = v < C ? y[i] * (d[i] - dot) * inv_tau / pe[i] : 0.f;
    }
}
// demb[idx[r]] += coef·dbow[r]
__global__ __launch_bounds__(256) void gumbel_emb_grad_kernel(const float* __restrict__ dbow, const int* __restrict__ idx,
                                                              const float* __restrict__ stats, float* __restrict__ demb, int V, int W) {
    const int r = blockIdx.x, t = idx[r];
    if (t >= V) return;
    const float ymax = 1.0f / stats[2 * r + 1];
    const float coef = (1.0f - ymax) + ymax;
    for (int c = threadIdx.x; c < W; c += 256) atomicAdd(&demb[(size_t)t * W + c], coef * dbow[(size_t)r * W + c]);
}

static int ptr_set_lds(const void* fn) { return svpc_raise_lds_once(fn, "ptr_attn"); }   // once per kernel symbol, process-wide table (api.cpp)

extern "C" {

int svpc_ptr_attn_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, float* att, int T,
                      int lt, int e_max, int D, hipStream_t s) {
    if (T == 0) return 0;
    SVPC_REQUIRE(e_max <= PTR_EMAX, "ptr_attn: at most 32 entities");
    SVPC_REQUIRE(D % 4 == 0 && ((((uintptr_t)dec) | ((uintptr_t)proj) | ((uintptr_t)bank)) & 15) == 0, "ptr_attn: rows must be 16-byte aligned");
    const size_t lds = ((size_t)(lt + PTR_EC) * D + (size_t)lt * PTR_EMAX) * sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "ptr_attn: sentence rows do not fit LDS");
    int rc = ptr_set_lds((const void*)ptr_attn_fwd_kernel);
    if (rc) return rc;
    const int nt = D >= 768 ? 768 : (D >= 512 ? 512 : 256);
    hipLaunchKernelGGL(ptr_attn_fwd_kernel, dim3(T), dim3(nt), lds, s, dec, proj, bank, step_ne, pi, att, lt, e_max, D, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr);
    return svpc_check_launch("ptr_attn_fwd");
}
// the same with the generation gate p_gen = sigmoid([dec ; att]·w + b) per row (w: 2D floats, b: 1) computed in the launch; att may be
// null (forward only: the decoding iterations need pi and p_gen); lt = 1 (one new position per sentence), D ≤ 768
int svpc_ptr_attn_pgen_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, float* att,
                           const float* pgen_w, const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, hipStream_t s) {
    if (T == 0) return 0;
    SVPC_REQUIRE(e_max <= PTR_EMAX, "ptr_attn: at most 32 entities");
    SVPC_REQUIRE(D % 4 == 0 && ((((uintptr_t)dec) | ((uintptr_t)proj) | ((uintptr_t)bank)) & 15) == 0, "ptr_attn: rows must be 16-byte aligned");
    SVPC_REQUIRE(pgen_w && pgen_b && pgen && lt == 1 && D <= 768, "ptr_attn_pgen: one position per step row, D <= 768");
    const size_t lds = ((size_t)(lt + PTR_EC) * D + (size_t)lt * PTR_EMAX) * sizeof(float);
    int rc = ptr_set_lds((const void*)ptr_attn_fwd_kernel);
    if (rc) return rc;
    const int nt = D >= 768 ? 768 : (D >= 512 ? 512 : 256);
    hipLaunchKernelGGL(ptr_attn_fwd_kernel, dim3(T), dim3(nt), lds, s, dec, proj, bank, step_ne, pi, att, lt, e_max, D, pgen_w, pgen_b, pgen);
    return svpc_check_launch("ptr_attn_pgen_fwd");
}
int svpc_ptr_attn_bwd(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                      const float* datt, float* ddec, float* dproj, float* dbank, int T, int lt, int e_max, int D, hipStream_t s) {
    if (T == 0) return 0;
    SVPC_REQUIRE(e_max <= PTR_EMAX, "ptr_attn: at most 32 entities");
    SVPC_REQUIRE(lt <= PTR_LTMAX, "ptr_attn: at most 32 tokens per sentence");
    SVPC_REQUIRE(D % 4 == 0 && ((((uintptr_t)datt) | ((uintptr_t)proj) | ((uintptr_t)bank)) & 15) == 0, "ptr_attn: rows must be 16-byte aligned");
    const size_t lds = ((size_t)(lt + PTR_EC) * D + (size_t)2 * lt * PTR_EMAX + (size_t)2 * PTR_EMAX * PTR_LTMAX) * sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "ptr_attn: sentence rows do not fit LDS");
    int rc = ptr_set_lds((const void*)ptr_attn_bwd_kernel);
    if (rc) return rc;
    const int nt = D >= 768 ? 768 : (D >= 512 ? 512 : 256);
    hipLaunchKernelGGL(ptr_attn_bwd_kernel, dim3(T), dim3(nt), lds, s, dec, proj, bank, step_ne, pi, dpi, datt, ddec, dproj, dbank, lt,
                       e_max, D);
    return svpc_check_launch("ptr_attn_bwd");
}

int svpc_ptr_attn_gate_fwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, const float* pgen_w,
                             const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, const int* row_off, const int* row_len,
                             hipStream_t s);
int svpc_ptr_attn_gate_bwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                             const float* pgen, const float* dpgen, const float* pgen_w, float* ddec, float* dproj, float* dbank,
                             float* wpart, int T, int lt, int e_max, int D, const int* row_off, const int* row_len, hipStream_t s);
// pointer attention + generation gate of every row, training form (see ptr_attn_gate_*_kernel): forward → pi (T·lt, e_max), pgen (T·lt);
// backward ← dpi, dpgen → ddec, dproj, dbank and the per-step partial sums wpart (T, 2D+1) of [d pgen_w | d pgen_b]
int svpc_ptr_attn_gate_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, const float* pgen_w,
                           const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, hipStream_t s) {
    return svpc_ptr_attn_gate_fwd_r(dec, proj, bank, step_ne, pi, pgen_w, pgen_b, pgen, T, lt, e_max, D, nullptr, nullptr, s);
}
// the same over RAGGED sentences (valid tokens only): sentence j owns the rows [row_off[j], row_off[j] + row_len[j]) of dec, pi, pgen
// (row_len[j] <= lt, the padded length)
int svpc_ptr_attn_gate_fwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, const float* pgen_w,
                             const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, const int* row_off, const int* row_len,
                             hipStream_t s) {
    if (T == 0) return 0;
    SVPC_REQUIRE((row_off == nullptr) == (row_len == nullptr), "ptr_attn_gate: row_off and row_len come together");
    SVPC_REQUIRE(e_max <= PTR_EMAX, "ptr_attn: at most 32 entities");
    SVPC_REQUIRE(lt <= PTR_LTMAX, "ptr_attn: at most 32 tokens per sentence");
    SVPC_REQUIRE(pgen_w && pgen_b && pgen, "ptr_attn_gate: gate weights missing");
    SVPC_REQUIRE(D % 4 == 0 && ((((uintptr_t)dec) | ((uintptr_t)proj) | ((uintptr_t)bank)) & 15) == 0, "ptr_attn: rows must be 16-byte aligned");
    static int rows_form = -1;
    if (rows_form < 0) { const char* e = getenv("SVPC_PTR_GATE_ROWS"); rows_form = e ? atoi(e) : 1; }
    const size_t lds_r = ((size_t)e_max * D + PTR_LTMAX * 32 + 32) * sizeof(float);
    if (rows_form && (D == 768 || D == 512 || D == 256) && e_max <= 31 && lds_r <= 150 * 1024 && ((((uintptr_t)pgen_w)) & 3) == 0) {
#define PTR_ROWS_GO(KPLV)                                                                                                          \
        do {                                                                                                                       \
            int rc_ = ptr_set_lds((const void*)ptr_attn_gate_fwd_rows_kernel<KPLV>); if (rc_) return rc_;                           \
            hipLaunchKernelGGL((ptr_attn_gate_fwd_rows_kernel<KPLV>), dim3(T), dim3(64 * KPLV), lds_r, s, dec, proj, bank, step_ne, pi, lt, e_max, \
                               pgen_w, pgen_b, pgen, row_off, row_len);                                                             \
        } while (0)
        if (D == 768) PTR_ROWS_GO(12); else if (D == 512) PTR_ROWS_GO(8); else PTR_ROWS_GO(4);
#undef PTR_ROWS_GO
        return svpc_check_launch("ptr_attn_gate_fwd");
    }
    const int nt = D >= 768 ? 768 : (D >= 512 ? 512 : 256);
    const size_t lds = ((size_t)(lt + PTR_EC) * D + (size_t)lt * PTR_EMAX + (size_t)(nt / 64) * PTR_LTMAX) * sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "ptr_attn: sentence rows do not fit LDS");
    int rc = ptr_set_lds((const void*)ptr_attn_gate_fwd_kernel);
    if (rc) return rc;
    hipLaunchKernelGGL(ptr_attn_gate_fwd_kernel, dim3(T), dim3(nt), lds, s, dec, proj, bank, step_ne, pi, lt, e_max, D, pgen_w, pgen_b, pgen,
                       row_off, row_len);
    return svpc_check_launch("ptr_attn_gate_fwd");
}
int svpc_ptr_attn_gate_bwd(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                           const float* pgen, const float* dpgen, const float* pgen_w, float* ddec, float* dproj, float* dbank,
                           float* wpart, int T, int lt, int e_max, int D, hipStream_t s) {
    return svpc_ptr_attn_gate_bwd_r(dec, proj, bank, step_ne, pi, dpi, pgen, dpgen, pgen_w, ddec, dproj, dbank, wpart, T, lt, e_max, D, nullptr,
                                    nullptr, s);
}
int svpc_ptr_attn_gate_bwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                             const float* pgen, const float* dpgen, const float* pgen_w, float* ddec, float* dproj, float* dbank,
                             float* wpart, int T, int lt, int e_max, int D, const int* row_off, const int* row_len, hipStream_t s) {
    if (T == 0) return 0;
    SVPC_REQUIRE((row_off == nullptr) == (row_len == nullptr), "ptr_attn_gate: row_off and row_len come together");
    SVPC_REQUIRE(e_max <= PTR_EMAX, "ptr_attn: at most 32 entities");
    SVPC_REQUIRE(lt <= PTR_LTMAX, "ptr_attn: at most 32 tokens per sentence");
    SVPC_REQUIRE(D % 4 == 0 && ((((uintptr_t)pgen_w) | ((uintptr_t)proj) | ((uintptr_t)bank)) & 15) == 0, "ptr_attn: rows must be 16-byte aligned");
    const size_t lds = ((size_t)(1 + PTR_EC) * D + (size_t)2 * lt * PTR_EMAX + (size_t)2 * PTR_EMAX * PTR_LTMAX + 2 * PTR_EMAX + PTR_LTMAX) *
                       sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "ptr_attn: entity rows do not fit LDS");
    int rc = ptr_set_lds((const void*)ptr_attn_gate_bwd_kernel);
    if (rc) return rc;
    const int nt = D >= 768 ? 768 : (D >= 512 ? 512 : 256);
    hipLaunchKernelGGL(ptr_attn_gate_bwd_kernel, dim3(T), dim3(nt), lds, s, dec, proj, bank, step_ne, pi, dpi, pgen, dpgen, pgen_w, ddec, dproj,
                       dbank, wpart, lt, e_max, D, row_off, row_len);
    return svpc_check_launch("ptr_attn_gate_bwd");
}
int svpc_ptr_mix_loss_fwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                          const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w, float* P,
                          float* loss_rows, int R, int V, int c_max, int e_max, float smoothing, hipStream_t s) {
    if (R == 0) return 0;
    MixArgs a{};
    a.logits = logits; a.g = g; a.pi = pi; a.labels = labels; a.row_c = row_c; a.row_vid = row_vid; a.csr_off = csr_off;
    a.csr_ent = csr_ent; a.csr_id = csr_id; a.csr_w = csr_w; a.P = P; a.loss_rows = loss_rows; a.V = V; a.c_max = c_max;
    a.em = e_max; a.smoothing = smoothing;
    SVPC_REQUIRE(smoothing > 0.f, "ptr_mix_loss: label_smoothing == 0 is the cross-entropy branch (svpc_ptr_mix_ce_fwd)");
    hipLaunchKernelGGL(ptr_mix_loss_fwd_kernel, dim3(R), dim3(256), (size_t)c_max * sizeof(float), s, a);
    return svpc_check_launch("ptr_mix_loss_fwd");
}
int svpc_ptr_mix_ce_bwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                        const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w,
                        const float* P, const float* dP_ext, const float* dloss, float* dlogits, float* dg, float* dpi, int R, int V,
                        int c_max, int e_max, float smoothing, const float* row_w, hipStream_t s);
int svpc_ptr_mix_loss_bwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                          const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w,
                          const float* P, const float* dP_ext, const float* dloss, float* dlogits, float* dg, float* dpi, int R, int V,
                          int c_max, int e_max, float smoothing, hipStream_t s) {
    return svpc_ptr_mix_ce_bwd(logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, P, dP_ext, dloss, dlogits, dg, dpi, R,
                               V, c_max, e_max, smoothing, nullptr, s);
}
int svpc_ptr_mix_ce_fwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                        const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w, float* P,
                        float* loss_rows, int R, int V, int c_max, int e_max, float smoothing, const float* row_w, hipStream_t s) {
    if (R == 0) return 0;
    SVPC_REQUIRE(row_w != nullptr || smoothing > 0.f, "ptr_mix_loss: label_smoothing == 0 needs the per-row weights (svpc_ce_row_weights)");
    MixArgs a{};
    a.logits = logits; a.g = g; a.pi = pi; a.labels = labels; a.row_c = row_c; a.row_vid = row_vid; a.csr_off = csr_off;
    a.csr_ent = csr_ent; a.csr_id = csr_id; a.csr_w = csr_w; a.P = P; a.loss_rows = loss_rows; a.V = V; a.c_max = c_max;
    a.em = e_max; a.smoothing = smoothing; a.row_w = row_w;
    hipLaunchKernelGGL(ptr_mix_loss_fwd_kernel, dim3(R), dim3(256), (size_t)c_max * sizeof(float), s, a);
    return svpc_check_launch("ptr_mix_loss_fwd");
}
int svpc_ptr_mix_ce_bwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                        const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w,
                        const float* P, const float* dP_ext, const float* dloss, float* dlogits, float* dg, float* dpi, int R, int V,
                        int c_max, int e_max, float smoothing, const float* row_w, hipStream_t s) {
    if (R == 0) return 0;
    SVPC_REQUIRE(row_w != nullptr || smoothing > 0.f, "ptr_mix_loss: label_smoothing == 0 needs the per-row weights (svpc_ce_row_weights)");
    MixArgs a{};
    a.logits = logits; a.g = g; a.pi = pi; a.labels = labels; a.row_c = row_c; a.row_vid = row_vid; a.csr_off = csr_off;
    a.csr_ent = csr_ent; a.csr_id = csr_id; a.csr_w = csr_w; a.P = const_cast<float*>(P); a.V = V; a.c_max = c_max; a.em = e_max;
    a.smoothing = smoothing; a.dP_ext = dP_ext; a.dloss = dloss; a.dlogits = dlogits; a.dg = dg; a.dpi = dpi; a.row_w = row_w;
    hipLaunchKernelGGL(ptr_mix_loss_bwd_kernel, dim3(R), dim3(256), (size_t)2 * c_max * sizeof(float), s, a);
    return svpc_check_launch("ptr_mix_loss_bwd");
}
int svpc_ce_row_weights(const int* labels, const int* row_vid, int R, int n_vid, float* row_w, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(ce_row_weights_kernel, dim3(1), dim3(1024), 0, s, labels, row_vid, R, n_vid, row_w);
    return svpc_check_launch("ce_row_weights");
}
int svpc_gumbel_fwd(const float* P, const float* noise, const int* row_c, const float* emb, float* bow, int* idx, float* stats, int R,
                    int c_max, int V, int W, float tau, hipStream_t s) {
    if (R == 0) return 0;
    if (c_max <= 1024)
        hipLaunchKernelGGL(gumbel_fwd_row_kernel<16>, dim3(ceil_div(R, 4)), dim3(256), 0, s, P, noise, row_c, emb, bow, idx, stats, R, c_max, V,
                           W, 1.0f / tau);
    else
        hipLaunchKernelGGL(gumbel_fwd_kernel, dim3(R), dim3(256), 0, s, P, noise, row_c, emb, bow, idx, stats, c_max, V, W, 1.0f / tau);
    return svpc_check_launch("gumbel_fwd");
}
int svpc_gumbel_bwd(const float* P, const float* noise, const int* row_c, const float* stats, const float* dy, float* dP, int R,
                    int c_max, int V, float tau, hipStream_t s) {
    if (R == 0) return 0;
    if (c_max <= 1024)
        hipLaunchKernelGGL(gumbel_bwd_row_kernel<16>, dim3(ceil_div(R, 4)), dim3(256), 0, s, P, noise, row_c, stats, dy, dP, R, c_max, V,
                           1.0f / tau);
    else
        hipLaunchKernelGGL(gumbel_bwd_kernel, dim3(R), dim3(256), 0, s, P, noise, row_c, stats, dy, dP, c_max, V, 1.0f / tau);
    return svpc_check_launch("gumbel_bwd");
}
int svpc_gumbel_emb_grad(const float* dbow, const int* idx, const float* stats, float* demb, int R, int V, int W, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(gumbel_emb_grad_kernel, dim3(R), dim3(256), 0, s, dbow, idx, stats, demb, V, W);
    return svpc_check_launch("gumbel_emb_grad");
}

}  // extern "C"
