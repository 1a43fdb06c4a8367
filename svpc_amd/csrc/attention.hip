// Segmented multi-head attention core, forward and backward, fp32 (reference: src/rtransformer/model.py:194-219 —
// scores/sqrt(dh) + (1-mask)·(-10000), softmax over keys, dropout on the probabilities, P·V, heads merged).
// One kernel serves every use on the hot path through a segmentation table (q_off,q_len,k_off,k_len per sequence):
//   clip encoder  100×100 keys (key-pad mask)            step-wise encoder  S_b×S_b (ragged)
//   decoder self  22×22 (causal ∧ pad)                   decoder→memory     22×M, M ≤ 3
// Q/K/V are read in place from the packed projection output (row stride ld*, column offset per head), the context is
// written head-merged.  K and V of one (sequence, head) live in LDS with a +1 padded row (conflict-free column reads);
// one 64-lane wave owns a query row: lanes sweep keys for QKᵀ, shuffle-reduce the softmax, lanes sweep head dims for PV.
// HBM-bound: algorithmic bytes per (sequence, head) = (Lq + 2·Lk + Lq)·dh·4.
// Backward is two deterministic passes without atomics: rows → dQ (+δ = dO·O), keys → dK, dV (scores recomputed from LSE).
#include "common.h"

struct AttnArgs {
    const float* Q; int ldq; const float* K; int ldk; const float* V; int ldv;
    float* O; int ldo; float* LSE;            // LSE: (n_seq, H, max_q)
    const int* seq;                           // (4, n_seq): q_off, q_len, k_off, k_len
    int n_seq, H, dh, max_q, max_k;
    const float* key_mask;                    // per key row (flat), 1 = attend; may be null
    int causal; float scale; float p_drop; uint32_t site; const u64* seed;
    // backward only
    const float* dO; int lddo; float* dQ; int lddq; float* dK; int lddk; float* dV; int lddv; float* delta;
};

constexpr int QROWS = 32;   // query rows per workgroup (4 waves × 8 rows)
constexpr int MAXKPL = 4;   // keys per lane → max_k ≤ 256

__device__ __forceinline__ float mask_term(const AttnArgs& a, int k_off, int i, int j) {
    float m = a.key_mask ? a.key_mask[k_off + j] : 1.0f;
    if (a.causal && j > i) m = 0.f;
    return (1.0f - m) * -10000.0f;
}

// grid: (n_seq*H, ceil(max_q/QROWS)); dynamic LDS: 2*Lk*(dh+1) + 4*dh + 4*max_k floats
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ float smem[];
    const int sh = blockIdx.x, s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const int dh = a.dh, ldr = dh + 1;
    float* Ks = smem;
    float* Vs = Ks + a.max_k * ldr;
    float* qb = Vs + a.max_k * ldr;            // 4 × dh
    float* pb = qb + 4 * dh;                   // 4 × max_k
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = blockIdx.y * QROWS;
    if (i0 >= q_len) return;
    // this wave's query rows, requested with the K / V images (read inside the loop, every group of four rows began with a memory round
    // trip of its own); the loop ends with the sequence (a 12-row sequence ran 8 rounds of three barriers, 5 of them over nothing)
    float qreg[QROWS / 4];
#pragma unroll
    for (int it = 0; it < QROWS / 4; ++it)
        qreg[it] = a.Q[(size_t)(q_off + min(i0 + it * 4 + wave, q_len - 1)) * a.ldq + h * dh + min(lane, dh - 1)];
    float mterm[MAXKPL];              // the key-pad term of this lane's keys: once per workgroup, not one conditional load per query row
#pragma unroll
    for (int u = 0; u < MAXKPL; ++u) mterm[u] = 0.f;
    if (a.key_mask) {
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) mterm[u] = (1.0f - a.key_mask[k_off + min(lane + 64 * u, k_len - 1)]) * -10000.0f;
    }
    for (int e0 = 0; e0 < k_len * dh; e0 += 4 * 256) {          // four pieces of each image in flight per thread (clamped, unconditional)
        float kq[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * 256 + (int)threadIdx.x, k_len * dh - 1), j = e / dh, d = e - j * dh;
            kq[u] = a.K[(size_t)(k_off + j) * a.ldk + h * dh + d];
            vq[u] = a.V[(size_t)(k_off + j) * a.ldv + h * dh + d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + (int)threadIdx.x, j = e / dh, d = e - j * dh;
            if (e < k_len * dh) { Ks[j * ldr + d] = kq[u]; Vs[j * ldr + d] = vq[u]; }
        }
    }
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float ik = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
#pragma unroll
    for (int it = 0; it < QROWS / 4; ++it) {
        if (i0 + it * 4 >= q_len) break;               // (uniform over the workgroup)
        const int i = i0 + it * 4 + wave;
        const bool valid = i < q_len;
        __syncthreads();
        if (valid && lane < dh) qb[wave * dh + lane] = qreg[it] * a.scale;
        __syncthreads();
        float sc[MAXKPL];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) {
            const int j = lane + 64 * u;
            sc[u] = -INFINITY;
            if (valid && j < k_len) {
                float acc = 0.f;
                const float* kr = Ks + j * ldr;
                const float* qr = qb + wave * dh;
                for (int d = 0; d < dh; ++d) acc += qr[d] * kr[d];
                acc += (a.causal && j > i) ? -10000.0f : mterm[u];
                sc[u] = acc;
                mx = fmaxf(mx, acc);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) {
            const int j = lane + 64 * u;
            if (valid && j < k_len) { sc[u] = expf(sc[u] - mx); sum += sc[u]; }
        }
        sum = wave_sum(sum);
        const float inv = valid ? 1.0f / sum : 0.f;
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) {
            const int j = lane + 64 * u;
            if (valid && j < k_len) {
                float p = sc[u] * inv;
                if (a.p_drop > 0.f)
                    p *= attn_drop_scale(seed, a.site, (u64)(s * a.H + h) * a.max_q + i, (uint32_t)j, a.p_drop, ik);
                pb[wave * a.max_k + j] = p;
            }
        }
        if (valid && lane == 0 && a.LSE) a.LSE[((size_t)s * a.H + h) * a.max_q + i] = mx + logf(sum);
        __syncthreads();
        if (valid && lane < dh) {
            float acc = 0.f;
            const float* pr = pb + wave * a.max_k;
            for (int j = 0; j < k_len; ++j) acc += pr[j] * Vs[j * ldr + lane];
            a.O[(size_t)(q_off + i) * a.ldo + h * dh + lane] = acc;
        }
    }
}

// pass A: per query row → dQ, delta.  Same LDS carve as forward plus a dO row buffer (4 × dh).
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs a) {
    extern __shared__ float smem[];
    const int sh = blockIdx.x, s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const int dh = a.dh, ldr = dh + 1;
    float* Ks = smem;
    float* Vs = Ks + a.max_k * ldr;
    float* qb = Vs + a.max_k * ldr;            // 4 × dh (scaled q)
    float* db = qb + 4 * dh;                   // 4 × dh (dO row)
    float* pb = db + 4 * dh;                   // 4 × max_k (dS row)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = blockIdx.y * QROWS;
    if (i0 >= q_len) return;
    for (int e = threadIdx.x; e < k_len * dh; e += 256) {
        const int j = e / dh, d = e - j * dh;
        Ks[j * ldr + d] = a.K[(size_t)(k_off + j) * a.ldk + h * dh + d];
        Vs[j * ldr + d] = a.V[(size_t)(k_off + j) * a.ldv + h * dh + d];
    }
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float ik = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    for (int it = 0; it < QROWS / 4; ++it) {
        const int i = i0 + it * 4 + wave;
        const bool valid = i < q_len;
        __syncthreads();
        float dlt = 0.f;
        if (valid && lane < dh) {
            const float dov = a.dO[(size_t)(q_off + i) * a.lddo + h * dh + lane];
            qb[wave * dh + lane] = a.Q[(size_t)(q_off + i) * a.ldq + h * dh + lane] * a.scale;
            db[wave * dh + lane] = dov;
            dlt = dov * a.O[(size_t)(q_off + i) * a.ldo + h * dh + lane];
        }
        dlt = wave_sum(dlt);
        __syncthreads();
        const float lse = valid ? a.LSE[((size_t)s * a.H + h) * a.max_q + i] : 0.f;
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) {
            const int j = lane + 64 * u;
            if (valid && j < k_len) {
                float sacc = 0.f, dp = 0.f;
                const float* kr = Ks + j * ldr;
                const float* vr = Vs + j * ldr;
                const float* qr = qb + wave * dh;
                const float* dr = db + wave * dh;
                for (int d = 0; d < dh; ++d) { sacc += qr[d] * kr[d]; dp += dr[d] * vr[d]; }
                sacc += mask_term(a, k_off, i, j);
                const float p = expf(sacc - lse);
                if (a.p_drop > 0.f)
                    dp *= attn_drop_scale(seed, a.site, (u64)(s * a.H + h) * a.max_q + i, (uint32_t)j, a.p_drop, ik);
                pb[wave * a.max_k + j] = p * (dp - dlt);
            }
        }
        if (valid && lane == 0) a.delta[((size_t)s * a.H + h) * a.max_q + i] = dlt;
        __syncthreads();
        if (valid && lane < dh) {
            float acc = 0.f;
            const float* pr = pb + wave * a.max_k;
            for (int j = 0; j < k_len; ++j) acc += pr[j] * Ks[j * ldr + lane];
            a.dQ[(size_t)(q_off + i) * a.lddq + h * dh + lane] = acc * a.scale;
        }
    }
}

// pass B: per key row → dK, dV.  LDS: Q (scaled) and dO of the whole sequence, per-wave k / v rows and p̃ / dS rows.
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnArgs a) {
    extern __shared__ float smem[];
    const int sh = blockIdx.x, s = sh / a.H, h = sh - s * a.H;
    const int q_off = a.seq[s], q_len = a.seq[a.n_seq + s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const int dh = a.dh, ldr = dh + 1;
    float* Qs = smem;
    float* Ds = Qs + a.max_q * ldr;
    float* kb = Ds + a.max_q * ldr;            // 4 × dh
    float* vb = kb + 4 * dh;                   // 4 × dh
    float* pb = vb + 4 * dh;                   // 4 × max_q  (p̃)
    float* sb = pb + 4 * a.max_q;              // 4 × max_q  (dS)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = blockIdx.y * QROWS;
    if (j0 >= k_len) return;
    for (int e = threadIdx.x; e < q_len * dh; e += 256) {
        const int i = e / dh, d = e - i * dh;
        Qs[i * ldr + d] = a.Q[(size_t)(q_off + i) * a.ldq + h * dh + d] * a.scale;
        Ds[i * ldr + d] = a.dO[(size_t)(q_off + i) * a.lddo + h * dh + d];
    }
    const u64 seed = a.p_drop > 0.f ? a.seed[0] : 0ull;
    const float ik = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const size_t stat0 = ((size_t)s * a.H + h) * a.max_q;
    for (int it = 0; it < QROWS / 4; ++it) {
        const int j = j0 + it * 4 + wave;
        const bool valid = j < k_len;
        __syncthreads();
        if (valid && lane < dh) {
            kb[wave * dh + lane] = a.K[(size_t)(k_off + j) * a.ldk + h * dh + lane];
            vb[wave * dh + lane] = a.V[(size_t)(k_off + j) * a.ldv + h * dh + lane];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXKPL; ++u) {
            const int i = lane + 64 * u;
            if (valid && i < q_len) {
                float sacc = 0.f, dp = 0.f;
                const float* qr = Qs + i * ldr;
                const float* dr = Ds + i * ldr;
                const float* kr = kb + wave * dh;
                const float* vr = vb + wave * dh;
                for (int d = 0; d < dh; ++d) { sacc += qr[d] * kr[d]; dp += dr[d] * vr[d]; }
                sacc += mask_term(a, k_off, i, j);
                const float p = expf(sacc - a.LSE[stat0 + i]);
                float dm = 1.0f;
                if (a.p_drop > 0.f) dm = attn_drop_scale(seed, a.site, (u64)(s * a.H + h) * a.max_q + i, (uint32_t)j, a.p_drop, ik);
                pb[wave * a.max_q + i] = p * dm;
                sb[wave * a.max_q + i] = p * (dp * dm - a.delta[stat0 + i]);
            }
        }
        __syncthreads();
        if (valid && lane < dh) {
            float accv = 0.f, acck = 0.f;
            const float* pr = pb + wave * a.max_q;
            const float* sr = sb + wave * a.max_q;
            for (int i = 0; i < q_len; ++i) {
                accv += pr[i] * Ds[i * ldr + lane];
                acck += sr[i] * Qs[i * ldr + lane];   // Qs already carries 1/sqrt(dh)
            }
            a.dV[(size_t)(k_off + j) * a.lddv + h * dh + lane] = accv;
            a.dK[(size_t)(k_off + j) * a.lddk + h * dh + lane] = acck;
        }
    }
}

static int set_lds(const void* fn, size_t bytes) { return svpc_raise_lds_once(fn, "attention"); }   // once per kernel symbol, process-wide table (api.cpp)

// ---- single-query attention (incremental decoding: one new token per sentence attends to its cached keys; translator.py:88-100
// under the causal mask).  One wave per (sequence, head), lane = head dimension; the keys/values of up to 32 positions are
// fetched together (all loads in flight), scores by wave reductions, softmax in registers.  fp32 throughout, forward only.
template <int CH>
__global__ __launch_bounds__(256) void attn_q1_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int sh = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (sh >= a.n_seq * a.H) return;
    const int s = sh / a.H, h = sh - s * a.H, dh = a.dh;
    const int q_off = a.seq[s], k_off = a.seq[2 * a.n_seq + s], k_len = a.seq[3 * a.n_seq + s];
    const bool on = lane < dh;
    const float q = on ? a.Q[(size_t)q_off * a.ldq + h * dh + lane] * a.scale : 0.f;
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int j0 = 0; j0 < k_len; j0 += CH) {
        float kr[CH], vr[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool ok = on && (j0 + j < k_len);
            kr[j] = ok ? a.K[(size_t)(k_off + j0 + j) * a.ldk + h * dh + lane] : 0.f;
            vr[j] = ok ? a.V[(size_t)(k_off + j0 + j) * a.ldv + h * dh + lane] : 0.f;
        }
        float sc[CH];
        float mx = m;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float d = wave_sum(q * kr[j]);
            if (j0 + j < k_len) {
                if (a.key_mask) d += (1.0f - a.key_mask[k_off + j0 + j]) * -10000.0f;
            } else d = -INFINITY;
            sc[j] = d;
            mx = fmaxf(mx, d);
        }
        const float corr = expf(m - mx);            // 0 on the first chunk (m = -inf)
        l *= corr; acc *= corr;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float p = expf(sc[j] - mx);       // 0 for padded positions
            l += p;
            acc += p * vr[j];
        }
        m = mx;
    }
    if (on) a.O[(size_t)q_off * a.ldo + h * dh + lane] = acc / l;
    if (a.LSE && lane == 0) a.LSE[((size_t)s * a.H + h) * a.max_q] = m + logf(l);
}

// ---- one decoding step of an attention block in ONE launch (incremental greedy decoding, translator.py:88-112 through the decoder
// layer model.py:620-663): per sentence t,  out[t] = LayerNorm( x[t] + Attention(q[t]; K, V of the sentence) ), optionally after
// APPENDING the token's new key / value row to the sentence's cache (the causal mask lets position pos see exactly the pos+1 rows
// written so far).  Replaces cache copy + attn_q1 + LayerNorm (self-attention) and attn_q1 + LayerNorm (cross-attention over the 1-3
// memory rows).  One workgroup per sentence, a thread owns 4 consecutive dimensions (16 lanes = one head of 64): every key row of the
// sentence is requested before the first is used (MAXK 16-byte loads per thread), scores by row-of-16 DPP sums, softmax in registers,
// the normalisation statistics by two block reductions (mean, then the centred second moment, as the reference's BertLayerNorm).
// fp32 throughout; forward only.
struct Q1LnArgs {
    const float* Q; int ldq; float* K; float* V; int ldkv; int k_stride; int n_keys;
    const float* newK; const float* newV; int ldnew;
    const float* X; int ldx; const float* gamma; const float* beta; float eps; float* O; int ldo;
    int T, D; float scale;
};
// 16 bytes of the key / value cache with the non-temporal hint: the cache (up to 104 MB per layer and iteration) is read once per
// iteration and must not push the decoder's weights (71 MB, re-read every iteration) out of the Infinity Cache
typedef float q1_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 q1_stream4(const float* p) {
    const q1_f4 v = __builtin_nontemporal_load(reinterpret_cast<const q1_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add_<0xB1>(v); v = dpp_add_<0x4E>(v); v = dpp_add_<0x141>(v); v = dpp_add_<0x140>(v);
    return v;
}
template <int MAXK>
__global__ __launch_bounds__(256) void attn_q1_ln_kernel(Q1LnArgs a) {
    __shared__ float red[2][4];
    const int t = blockIdx.x, tid = threadIdx.x, d0 = 4 * tid;
    const int nw = blockDim.x >> 6;
    const bool app = a.newK != nullptr;
    const int nc = app ? a.n_keys - 1 : a.n_keys;                   // rows read from the cache
    const size_t row0 = (size_t)t * a.k_stride;
    float4 q = *reinterpret_cast<const float4*>(a.Q + (size_t)t * a.ldq + d0);
    const float4 xr = *reinterpret_cast<const float4*>(a.X + (size_t)t * a.ldx + d0);
    const float4 gm = *reinterpret_cast<const float4*>(a.gamma + d0), bt = *reinterpret_cast<const float4*>(a.beta + d0);
    float4 kn = make_float4(0.f, 0.f, 0.f, 0.f), vn = kn;
    if (app) {
        kn = *reinterpret_cast<const float4*>(a.newK + (size_t)t * a.ldnew + d0);
        vn = *reinterpret_cast<const float4*>(a.newV + (size_t)t * a.ldnew + d0);
    }
    float4 kr[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j)      // (rows past the end re-read the last one — unconditional loads, all in flight; nc == 0 reads row 0 of the sentence's own cache block)
        kr[j] = q1_stream4(a.K + (row0 + min(j, max(nc - 1, 0))) * a.ldkv + d0);
    q.x *= a.scale; q.y *= a.scale; q.z *= a.scale; q.w *= a.scale;
    float sc[MAXK], sn = -INFINITY, mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const float d = row16_sum(q.x * kr[j].x + q.y * kr[j].y + q.z * kr[j].z + q.w * kr[j].w);
        sc[j] = j < nc ? d : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    float4 vr[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        vr[j] = q1_stream4(a.V + (row0 + min(j, max(nc - 1, 0))) * a.ldkv + d0);
    if (app) {
        sn = row16_sum(q.x * kn.x + q.y * kn.y + q.z * kn.z + q.w * kn.w);
        mx = fmaxf(mx, sn);
        *reinterpret_cast<float4*>(a.K + (row0 + nc) * a.ldkv + d0) = kn;
        *reinterpret_cast<float4*>(a.V + (row0 + nc) * a.ldkv + d0) = vn;
    }
    float l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const float p = expf(sc[j] - mx);              // 0 past the end
        l += p;
        acc.x += p * vr[j].x; acc.y += p * vr[j].y; acc.z += p * vr[j].z; acc.w += p * vr[j].w;
    }
    if (app) {
        const float p = expf(sn - mx);
        l += p;
        acc.x += p * vn.x; acc.y += p * vn.y; acc.z += p * vn.z; acc.w += p * vn.w;
    }
    const float il = 1.0f / l;
    float4 y = make_float4(acc.x * il + xr.x, acc.y * il + xr.y, acc.z * il + xr.z, acc.w * il + xr.w);
    // LayerNorm over the D values of the row (model.py:150-166: u = mean, s = mean((x-u)^2), (x-u)/sqrt(s+eps)·w + b)
    float ps = wave_sum((y.x + y.y) + (y.z + y.w));
    if ((tid & 63) == 0) red[0][tid >> 6] = ps;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nw; ++w) tot += red[0][w];
    const float mean = tot / (float)a.D;
    y.x -= mean; y.y -= mean; y.z -= mean; y.w -= mean;
    float pq = wave_sum((y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w));
    if ((tid & 63) == 0) red[1][tid >> 6] = pq;
    __syncthreads();
    float tq = 0.f;
    for (int w = 0; w < nw; ++w) tq += red[1][w];
    const float rstd = 1.0f / sqrtf(tq / (float)a.D + a.eps);
    float4 o = make_float4(y.x * rstd * gm.x + bt.x, y.y * rstd * gm.y + bt.y, y.z * rstd * gm.z + bt.z, y.w * rstd * gm.w + bt.w);
    *reinterpret_cast<float4*>(a.O + (size_t)t * a.ldo + d0) = o;
}

extern "C" {

// one query row per sequence (q_len == 1 for every sequence), no dropout, not causal-masked beyond k_len: forward only
int svpc_attn_q1_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                     const int* seq, int n_seq, int H, int dh, int max_k, const float* key_mask, float scale, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(dh <= 64, "attention: head dim must be <= 64");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.seq = seq;
    a.n_seq = n_seq; a.H = H; a.dh = dh; a.max_q = 1; a.key_mask = key_mask; a.scale = scale;
    // chunk of keys fetched and reduced together: 8 for the 1–3 memory slots of the cross-attention, 32 otherwise
    if (max_k <= 8) hipLaunchKernelGGL(attn_q1_kernel<8>, dim3(ceil_div(n_seq * H, 4)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_q1_kernel<32>, dim3(ceil_div(n_seq * H, 4)), dim3(256), 0, stream, a);
    return svpc_check_launch("attn_q1_fwd");
}

// 1 if svpc_attn_q1_ln_fwd takes the shape: heads of 64, D a multiple of 256 up to 1024, at most 32 key rows per sentence, 16-byte rows
int svpc_attn_q1_ln_supported(int D, int dh, int n_keys, int ldq, int ldkv, int ldnew, int ldx, int ldo) {
    if (dh != 64 || D % 256 != 0 || D > 1024 || n_keys < 1 || n_keys > 32) return 0;
    if ((ldq | ldkv | ldnew | ldx | ldo) & 3) return 0;
    return 1;
}
// out[t] = LayerNorm(X[t] + Attention(Q[t]; keys / values = rows t·k_stride … t·k_stride + n_keys − 1 of K / V)), T sentences, one query
// each; with newK / newV (both or neither; row t, leading dimension ldnew) the token's key / value row is FIRST stored as row
// t·k_stride + n_keys − 1 of K / V (the cache append of incremental decoding) — it is the last of the n_keys rows attended to.
int svpc_attn_q1_ln_fwd(const float* Q, int ldq, float* K, float* V, int ldkv, int k_stride, int n_keys, const float* newK, const float* newV,
                        int ldnew, const float* X, int ldx, const float* gamma, const float* beta, float eps, float* O, int ldo, int T, int D,
                        int dh, float scale, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(svpc_attn_q1_ln_supported(D, dh, n_keys, ldq, ldkv, newK ? ldnew : 0, ldx, ldo) == 1 && (newK == nullptr) == (newV == nullptr) &&
                     k_stride >= n_keys &&
                     ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)newK) | ((uintptr_t)newV) | ((uintptr_t)X) | ((uintptr_t)gamma) |
                       ((uintptr_t)beta) | ((uintptr_t)O)) & 15) == 0,
                 "attn_q1_ln: needs heads of 64, D % 256 == 0, D <= 1024, 1..32 key rows per sentence, 16-byte aligned rows");
    Q1LnArgs a{Q, ldq, K, V, ldkv, k_stride, n_keys, newK, newV, ldnew, X, ldx, gamma, beta, eps, O, ldo, T, D, scale};
    const int nc = newK ? n_keys - 1 : n_keys;
    const dim3 grid(T), block(D / 4);
    if (nc <= 4) hipLaunchKernelGGL(attn_q1_ln_kernel<4>, grid, block, 0, stream, a);
    else if (nc <= 8) hipLaunchKernelGGL(attn_q1_ln_kernel<8>, grid, block, 0, stream, a);
    else if (nc <= 16) hipLaunchKernelGGL(attn_q1_ln_kernel<16>, grid, block, 0, stream, a);
    else if (nc <= 24) hipLaunchKernelGGL(attn_q1_ln_kernel<24>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(attn_q1_ln_kernel<32>, grid, block, 0, stream, a);
    return svpc_check_launch("attn_q1_ln_fwd");
}

int svpc_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                  const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                  float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(dh <= 64, "attention: head dim must be <= 64");
    SVPC_REQUIRE(max_k <= 64 * MAXKPL && max_q <= 64 * MAXKPL, "attention: at most 256 keys/queries per sequence");
    SVPC_REQUIRE(p_drop <= 0.f || seed != nullptr, "attention: dropout needs a seed pointer");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.seq = seq;
    a.n_seq = n_seq; a.H = H; a.dh = dh; a.max_q = max_q; a.max_k = max_k; a.key_mask = key_mask; a.causal = causal;
    a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    const size_t lds = ((size_t)2 * max_k * (dh + 1) + 4 * dh + 4 * max_k) * sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "attention: K/V tile does not fit LDS");
    int rc = set_lds((const void*)attn_fwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(n_seq * H, ceil_div(max_q, QROWS)), dim3(256), lds, stream, a);
    return svpc_check_launch("attn_fwd");
}

// delta: scratch (n_seq, H, max_q).  dQ/dK/dV may alias column blocks of one packed gradient buffer.
int svpc_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                  const float* LSE, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                  float* delta, const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                  float scale, float p_drop, unsigned site, const u64* seed, hipStream_t stream) {
    if (n_seq == 0) return 0;
    SVPC_REQUIRE(dh <= 64, "attention: head dim must be <= 64");
    SVPC_REQUIRE(max_k <= 64 * MAXKPL && max_q <= 64 * MAXKPL, "attention: at most 256 keys/queries per sequence");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = const_cast<float*>(O); a.ldo = ldo;
    a.LSE = const_cast<float*>(LSE); a.seq = seq; a.n_seq = n_seq; a.H = H; a.dh = dh; a.max_q = max_q; a.max_k = max_k;
    a.key_mask = key_mask; a.causal = causal; a.scale = scale; a.p_drop = p_drop; a.site = site; a.seed = seed;
    a.dO = dO; a.lddo = lddo; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv; a.delta = delta;
    const size_t lds_q = ((size_t)2 * max_k * (dh + 1) + 8 * dh + 4 * max_k) * sizeof(float);
    const size_t lds_kv = ((size_t)2 * max_q * (dh + 1) + 8 * dh + 8 * max_q) * sizeof(float);
    SVPC_REQUIRE(lds_q <= 150 * 1024 && lds_kv <= 150 * 1024, "attention: tiles do not fit LDS");
    int rc = set_lds((const void*)attn_bwd_q_kernel, lds_q);
    if (rc) return rc;
    rc = set_lds((const void*)attn_bwd_kv_kernel, lds_kv);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(n_seq * H, ceil_div(max_q, QROWS)), dim3(256), lds_q, stream, a);
    rc = svpc_check_launch("attn_bwd_q");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(n_seq * H, ceil_div(max_k, QROWS)), dim3(256), lds_kv, stream, a);
    return svpc_check_launch("attn_bwd_kv");
}

}  // extern "C"
