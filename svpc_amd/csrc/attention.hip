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
        if (valid && lane < dh) qb[wave * dh + lane] = a.Q[(size_t)(q_off + i) * a.ldq + h * dh + lane] * a.scale;
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
                acc += mask_term(a, k_off, i, j);
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
