// Row-wise / element-wise kernels of the hot path (all HBM-bound; wave64 shuffle reductions, coalesced rows):
//   activation backward (+dropout) for the GEMM epilogues      (model.py:58-64 gelu, ReLU/sigmoid of :497,:552,:755-759)
//   span (weighted) mean + table add                           (ingredient pooling :125-139, [CLS]+step PE :1064,
//                                                               masked bag-of-words mean :1019-1021)
//   row normalise a/Σa (Eq. 1, :798), 3-way softmax (Eq. 3, :804)
//   BCE-sum (:871) and asymmetric loss rows (libs/ASL/src/loss_functions/losses.py:15-50)
//   LSTM cell (nn.LSTM gate order i,f,g,o; :865)
//   embedding-gradient scatter-add with padding row (nn.Embedding(padding_idx=0), :492,:519)
#include "common.h"

// ---- dz = dy * act'(.) * dropout ----------------------------------------------------------------------------
// `aux` is the pre-activation z for GELU, the activated (pre-dropout) output y for ReLU / sigmoid.
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux,
                                                      T* __restrict__ dz, size_t n, int act, float p, uint32_t site,
                                                      const u64* __restrict__ seed_ptr) {
    const u64 seed = p > 0.f ? seed_ptr[0] : 0ull;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float g = (float)dy[i];
        if (p > 0.f) g *= drop_scale(seed, site, i, p, ik);
        const float a = (float)aux[i];
        if (act == ACT_RELU) g = a > 0.f ? g : 0.f;
        else if (act == ACT_GELU) g *= gelu_erf_grad(a);
        else if (act == ACT_SIGMOID) g *= a * (1.0f - a);
        dz[i] = (T)g;
    }
}

// 16-byte form for the bf16 streams (8 elements per thread and iteration; n % 8 == 0, 16-byte aligned pointers)
__global__ __launch_bounds__(256) void act_bwd_bf16x8_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ aux, uint4* __restrict__ dz,
                                                             size_t n8, int act, float p, uint32_t site, const u64* __restrict__ seed_ptr) {
    const u64 seed = p > 0.f ? seed_ptr[0] : 0ull;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n8; q += (size_t)gridDim.x * 256) {
        const uint4 gv = dy[q], av = aux[q];
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, aw[4] = {av.x, av.y, av.z, av.w};
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g2[2] = {__uint_as_float(gw[j] << 16), __uint_as_float(gw[j] & 0xffff0000u)};
            const float a2[2] = {__uint_as_float(aw[j] << 16), __uint_as_float(aw[j] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float g = g2[e];
                if (p > 0.f) g *= drop_scale(seed, site, q * 8 + 2 * j + e, p, ik);
                const float a = a2[e];
                if (act == ACT_RELU) g = a > 0.f ? g : 0.f;
                else if (act == ACT_GELU) g *= gelu_erf_grad(a);
                else if (act == ACT_SIGMOID) g *= a * (1.0f - a);
                g2[e] = g;
            }
            union { __bf16 h[2]; uint32_t u; } pk;
            pk.h[0] = (__bf16)g2[0]; pk.h[1] = (__bf16)g2[1];
            ow[j] = pk.u;
        }
        dz[q] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                                  size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, size_t n, float p, uint32_t site,
                                                           const u64* __restrict__ seed_ptr) {
    const u64 seed = seed_ptr[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = drop_scale(seed, site, i, p, 1.0f);
}

// the keep mask (1 / 0) of the attention kernels' probability dropout: element (row, k) of a (n_rows, max_k) tensor, rows indexed
// (sequence·H + head)·max_q + query — the row-hash draw of common.h (test infrastructure hands it to the fp32 reference)
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(float* __restrict__ out, size_t n_rows, int max_k, float p, uint32_t site,
                                                                const u64* __restrict__ seed_ptr) {
    const u64 seed = seed_ptr[0];
    const size_t n = n_rows * (size_t)max_k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t row = i / (size_t)max_k;
        out[i] = attn_drop_scale(seed, site, row, (uint32_t)(i - row * (size_t)max_k), p, 1.0f);
    }
}

__global__ __launch_bounds__(256) void gumbel_noise_kernel(float* __restrict__ out, size_t n, uint32_t site,
                                                           const u64* __restrict__ seed_ptr) {
    const u64 seed = seed_ptr[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = gumbel_noise(seed, site, i);
}

__global__ void bump_seed_kernel(u64* seed) { seed[0] = seed[0] * 6364136223846793005ull + 1442695040888963407ull; }

// ---- deterministic full reduction: out[0] = scale * sum(x) (single workgroup, fixed order) -------------------
__global__ __launch_bounds__(256) void sum_all_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out, float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, size_t n, const float* __restrict__ v) {
    const float s = v[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = s;
}

// ---- rows: one wave per row ------------------------------------------------------------------------------------
// mode 0: y = a / sum(a)       mode 1: y = softmax(a)
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ a, float* __restrict__ y, int R, int C, int mode) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const float* ar = a + (size_t)r * C;
    float* yr = y + (size_t)r * C;
    if (mode == 0) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += ar[c];
        s = 1.0f / wave_sum(s);
        for (int c = lane; c < C; c += 64) yr[c] = ar[c] * s;
    } else {
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, ar[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(ar[c] - m);
        s = 1.0f / wave_sum(s);
        for (int c = lane; c < C; c += 64) yr[c] = expf(ar[c] - m) * s;
    }
}
// mode 0: da = (dy - sum(dy*y)) / sum(a)  [y = a/S]      mode 1: da = y (dy - sum(dy*y))
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ a, float* __restrict__ da, int R, int C, int mode) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const size_t o = (size_t)r * C;
    float dot = 0.f, s = 0.f;
    for (int c = lane; c < C; c += 64) { dot += dy[o + c] * y[o + c]; if (mode == 0) s += a[o + c]; }
    dot = wave_sum(dot);
    if (mode == 0) {
        s = 1.0f / wave_sum(s);
        for (int c = lane; c < C; c += 64) da[o + c] = (dy[o + c] - dot) * s;
    } else {
        for (int c = lane; c < C; c += 64) da[o + c] = y[o + c] * (dy[o + c] - dot);
    }
}

// ---- span weighted mean: out[g] = Σ_{r<len} w x / Σ w + add[add_idx[g]] ---------------------------------------
__global__ __launch_bounds__(256) void span_mean_fwd_kernel(const float* __restrict__ x, const int* __restrict__ starts,
                                                            const int* __restrict__ lens, const float* __restrict__ w,
                                                            const float* __restrict__ add, const int* __restrict__ add_idx,
                                                            float* __restrict__ out, int G, int D) {
    const int g = blockIdx.x;
    const int s = starts[g], l = lens[g];
    // eight rows of the span in flight per thread (clamped, unconditional loads; the weight of a row past the end is zeroed): a loop with
    // the load inside made every row of a 22-token sentence a memory round trip of its own (12–14 µs for 192 spans × 22 rows × 300 columns)
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = 0.f, wsum = 0.f;
        for (int r0 = 0; r0 < l; r0 += 8) {
            float xv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[(size_t)(s + min(r0 + u, l - 1)) * D + c];
            if (w) {                 // (one region for all eight weight loads, not a test around each)
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = w[s + min(r0 + u, l - 1)];
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wu = r0 + u < l ? wv[u] : 0.f;
                acc = fmaf(wu, xv[u], acc);
                wsum += wu;
            }
        }
        acc *= 1.0f / wsum;
        if (add) acc += add[(size_t)add_idx[g] * D + c];
        out[(size_t)g * D + c] = acc;
    }
}
// dx must be zero-initialised (rows outside every span get no gradient)
__global__ __launch_bounds__(256) void span_mean_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ starts,
                                                            const int* __restrict__ lens, const float* __restrict__ w,
                                                            float* __restrict__ dx, int G, int D) {
    const int g = blockIdx.x;
    const int s = starts[g], l = lens[g];
    float wsum = 0.f;
    if (w) { for (int r = 0; r < l; ++r) wsum += w[s + r]; } else wsum = (float)l;
    const float inv = 1.0f / wsum;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float d = dout[(size_t)g * D + c] * inv;
        for (int r = 0; r < l; ++r) dx[(size_t)(s + r) * D + c] = (w ? w[s + r] : 1.0f) * d;
    }
}

// ---- dtable[idx[r]] += dx[r]  (fp32 atomics, skipping the padding row) --------------------------------------
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ dx, const int* __restrict__ idx,
                                                               float* __restrict__ dtable, int R, int D, int pad_row) {
    const int r = blockIdx.x;
    const int t = idx[r];
    if (t == pad_row) return;
    for (int c = threadIdx.x; c < D; c += 256) atomicAdd(&dtable[(size_t)t * D + c], dx[(size_t)r * D + c]);
}

// ---- BCE rows (nn.BCELoss(sum): log clamped at -100) -----------------------------------------------------------
__global__ __launch_bounds__(256) void bce_rows_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                           const int* __restrict__ widths, float* __restrict__ out, int R, int C) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const int wdt = widths[r];
    float s = 0.f;
    for (int c = lane; c < wdt; c += 64) {
        const float pp = p[(size_t)r * C + c], yy = y[(size_t)r * C + c];
        s -= yy * fmaxf(logf(pp), -100.f) + (1.f - yy) * fmaxf(logf(1.f - pp), -100.f);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}
__global__ __launch_bounds__(256) void bce_rows_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ p,
                                                           const float* __restrict__ y, const int* __restrict__ widths,
                                                           float* __restrict__ dp, int R, int C) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const int wdt = widths[r];
    const float go = dout[r];
    for (int c = lane; c < C; c += 64) {
        float g = 0.f;
        if (c < wdt) {
            const float pp = p[(size_t)r * C + c], yy = y[(size_t)r * C + c];
            // the reference's BCE is torch's: its backward is (p - y) / max(p·(1-p), 1e-12) — NOT the derivative of the clamped forward
            // (aten binary_cross_entropy_backward); with a saturated probability (p = 7e-13, y = 1: test-sensitive weights) the two
            // differ by 37 % on that element, 2.5e-3 of a weight gradient (round 4)
            g = go * (pp - yy) / fmaxf((1.f - pp) * pp, 1e-12f);
        }
        dp[(size_t)r * C + c] = g;
    }
}

// ---- asymmetric loss rows (gamma- 4, gamma+ 1, clip .05, eps 1e-8; focal weights differentiated) --------------
__device__ __forceinline__ void asl_terms(float p, float y, float gneg, float gpos, float clip, float eps, float& loss, float& dldp) {
    const float pn_raw = 1.f - p + clip;
    const float pn = fminf(pn_raw, 1.f);
    const float dpn = pn_raw < 1.f ? -1.f : 0.f;               // d pn / d p
    const float lp = logf(fmaxf(p, eps)), ln = logf(fmaxf(pn, eps));
    const float dlp = p > eps ? 1.f / p : 0.f;
    const float dln = pn > eps ? dpn / pn : 0.f;
    const float ce = y * lp + (1.f - y) * ln;
    const float dce = y * dlp + (1.f - y) * dln;
    const float pt = p * y + pn * (1.f - y);
    const float dpt = y + dpn * (1.f - y);
    const float gam = gpos * y + gneg * (1.f - y);
    const float base = 1.f - pt;
    // base^gam and base^(gam-1): gamma is 1 (positives) or 4 (negatives) in the reference's loss (losses.py:15-22), which a
    // product does in 3 multiplies where the general powf costs ≈100 instructions — per element, twice, in a kernel that sits on
    // the step's dependent chain; other exponents take the general route
    float w, wm1;
    if (gam == 1.f) { w = base; wm1 = 1.f; }
    else if (gam == 4.f) { const float b2 = base * base; w = b2 * b2; wm1 = b2 * base; }
    else { w = powf(base, gam); wm1 = gam == 0.f ? 0.f : powf(base, gam - 1.f); }
    // d/dp base^gam = -gam * base^(gam-1) * dpt   (torch.pow backward; 0 where gam == 0)
    const float dw = gam == 0.f ? 0.f : -gam * wm1 * dpt;
    loss = -(ce * w);
    dldp = -(dce * w + ce * dw);
}
__global__ __launch_bounds__(256) void asl_rows_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                           const float* __restrict__ active, float* __restrict__ out, int R, int C,
                                                           float gneg, float gpos, float clip, float eps) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    float s = 0.f;
    if (active[r] != 0.f) {
        for (int c = lane; c < C; c += 64) {
            float l, d;
            asl_terms(p[(size_t)r * C + c], y[(size_t)r * C + c], gneg, gpos, clip, eps, l, d);
            s += l;
        }
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}
__global__ __launch_bounds__(256) void asl_rows_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ p,
                                                           const float* __restrict__ y, const float* __restrict__ active,
                                                           float* __restrict__ dp, int R, int C, float gneg, float gpos, float clip,
                                                           float eps) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    const float go = active[r] != 0.f ? dout[r] : 0.f;
    for (int c = lane; c < C; c += 64) {
        float l, d = 0.f;
        if (go != 0.f) asl_terms(p[(size_t)r * C + c], y[(size_t)r * C + c], gneg, gpos, clip, eps, l, d);
        dp[(size_t)r * C + c] = go * d;
    }
}
__global__ __launch_bounds__(256) void row_any_eq1_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C) {
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= R) return;
    float f = 0.f;
    for (int c = lane; c < C; c += 64) f = fmaxf(f, x[(size_t)r * C + c] == 1.0f ? 1.f : 0.f);
    f = wave_max(f);
    if (lane == 0) out[r] = f;
}
// ---- loss tail: the whole sum of the training loss in ONE launch (and its backward in one) -----------------------------------
// total = Σ cap_rows + [Σ_r BCE(e_p[r], align[r]) + Σ_{r: any(act[r]==1)} ASL(a_p[r], act[r])] + λ·[the same two sums for the
// re-simulator's r_e / r_a]   (reference: model.py:1110-1115 per-video sums, :1168-1188 total; BCE :869, ASL losses.py:15-50).
// The eager form was 15 launches forward (row kernels, five sum_all, adds) and ≈10 backward on a launch-bound chain.
struct LossTailArgs {
    const float* cap_rows; int n_cap;
    const float* e_p; const float* align; const int* widths; int R, Ce;
    const float* a_p; const float* act; int Ca;
    const float* r_e; const float* r_a; float lambda;
    float gneg, gpos, clip, eps;
    float* out;            // out[0] total, out[1] caption, out[2] entity, out[3] action, out[4] re-simulation (unweighted)
    const float* dout;     // backward: upstream gradient of the total (device scalar)
    float* d_cap; float* de_p; float* da_p; float* dr_e; float* dr_a;
};
constexpr int LT_CH = 8;      // columns per lane held in registers by the loss-tail kernels (action vocabulary ≤ 512)
__device__ __forceinline__ float bce_row_sum(const float* __restrict__ p, const float* __restrict__ y, int wdt, int lane) {
    float s = 0.f;
    for (int c = lane; c < wdt; c += 64) {
        const float pp = p[c], yy = y[c];
        s -= yy * fmaxf(logf(pp), -100.f) + (1.f - yy) * fmaxf(logf(1.f - pp), -100.f);
    }
    return wave_sum(s);
}
__device__ __forceinline__ float asl_row_sum(const float* __restrict__ p, const float* __restrict__ y, int C, int lane, const LossTailArgs& a) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        float l, d;
        asl_terms(p[c], y[c], a.gneg, a.gpos, a.clip, a.eps, l, d);
        s += l;
    }
    return wave_sum(s);
}
__device__ __forceinline__ bool row_has_one(const float* __restrict__ y, int C, int lane) {
    float f = 0.f;
    for (int c = lane; c < C; c += 64) f = fmaxf(f, y[c] == 1.0f ? 1.f : 0.f);
    return wave_max(f) != 0.f;
}
// Forward in ONE launch on many workgroups: blocks [0, ceil(R/4)) — a wave per row writes that row's three sums into `partial`;
// the remaining blocks sum 1,024 caption rows each.  The workgroup that draws the last ticket (agent-scope release / acquire around
// a relaxed counter, cdna_hip_programming.md Guideline 16) adds the partials in index order — deterministic — writes out[0..4] and
// resets the counter for the next launch (the counter lives in a buffer the caller zeroed once).
__global__ __launch_bounds__(256) void loss_tail_fwd_kernel(LossTailArgs a, int row_blocks, float* __restrict__ partial, int* __restrict__ counter) {
    __shared__ float red[4];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* prow = partial;                                   // [R][3]
    float* pcap = partial + (size_t)3 * a.R;                 // [cap blocks]
    if ((int)blockIdx.x < row_blocks) {
        const int r = blockIdx.x * 4 + wave;
        if (r < a.R) {
            const int wdt = a.widths[r];
            const float* yr = a.align + (size_t)r * a.Ce;
            const float* ar = a.act + (size_t)r * a.Ca;
            float ent = 0.f, actl = 0.f, re = 0.f;
            if (a.Ca <= 64 * LT_CH) {
                // every load of the row in flight at once (targets, both probability rows): the row test, then the sums, otherwise
                // form a chain of dependent memory round trips
                float yv[LT_CH], pv[LT_CH], rv[LT_CH];
#pragma unroll
                for (int k = 0; k < LT_CH; ++k) {
                    const int c = lane + 64 * k;
                    const bool in = c < a.Ca;
                    yv[k] = in ? ar[c] : 0.f;
                    pv[k] = (in && a.a_p) ? a.a_p[(size_t)r * a.Ca + c] : 0.5f;
                    rv[k] = (in && a.r_a) ? a.r_a[(size_t)r * a.Ca + c] : 0.5f;
                }
                if (a.e_p) ent = bce_row_sum(a.e_p + (size_t)r * a.Ce, yr, wdt, lane);
                if (a.r_e) re = bce_row_sum(a.r_e + (size_t)r * a.Ce, yr, wdt, lane);
                float f = 0.f;
#pragma unroll
                for (int k = 0; k < LT_CH; ++k) f = fmaxf(f, yv[k] == 1.0f ? 1.f : 0.f);
                if (wave_max(f) != 0.f) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int k = 0; k < LT_CH; ++k) {
                        if (lane + 64 * k < a.Ca) {
                            float l, d;
                            if (a.a_p) { asl_terms(pv[k], yv[k], a.gneg, a.gpos, a.clip, a.eps, l, d); s1 += l; }
                            if (a.r_a) { asl_terms(rv[k], yv[k], a.gneg, a.gpos, a.clip, a.eps, l, d); s2 += l; }
                        }
                    }
                    actl = wave_sum(s1);
                    re += wave_sum(s2);
                }
            } else {
                const bool on = row_has_one(ar, a.Ca, lane);
                if (a.e_p) ent = bce_row_sum(a.e_p + (size_t)r * a.Ce, yr, wdt, lane);
                if (a.a_p && on) actl = asl_row_sum(a.a_p + (size_t)r * a.Ca, ar, a.Ca, lane, a);
                if (a.r_e) re = bce_row_sum(a.r_e + (size_t)r * a.Ce, yr, wdt, lane);
                if (a.r_a && on) re += asl_row_sum(a.r_a + (size_t)r * a.Ca, ar, a.Ca, lane, a);
            }
            if (lane == 0) { prow[3 * r] = ent; prow[3 * r + 1] = actl; prow[3 * r + 2] = re; }
        }
    } else {
        const int cb = blockIdx.x - row_blocks;
        float c = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = cb * 1024 + k * 256 + threadIdx.x;
            if (i < a.n_cap) c += a.cap_rows[i];
        }
        c = block_sum_256(c, red);
        if (threadIdx.x == 0) pcap[cb] = c;
    }
    // ---- publish, draw a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (wave != 0) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, tc = 0.f;
    for (int r = lane; r < a.R; r += 64) { t0 += prow[3 * r]; t1 += prow[3 * r + 1]; t2 += prow[3 * r + 2]; }
    const int ncb = (int)gridDim.x - row_blocks;
    for (int i = lane; i < ncb; i += 64) tc += pcap[i];
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); tc = wave_sum(tc);
    if (lane == 0) {
        a.out[1] = tc; a.out[2] = t0; a.out[3] = t1; a.out[4] = t2;
        a.out[0] = ((tc + t0) + t1) + a.lambda * t2;          // the reference's order of additions (model.py:1188)
        __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void loss_tail_bce_bwd(const LossTailArgs& a, int r, int wdt, const float* __restrict__ yr, float g, int lane) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* p = which ? a.r_e : a.e_p;
        float* dp = which ? a.dr_e : a.de_p;
        const float go = which ? g * a.lambda : g;
        if (!p || !dp) continue;
        for (int c = lane; c < a.Ce; c += 64) {
            float v = 0.f;
            if (c < wdt) {
                const float pp = p[(size_t)r * a.Ce + c], yy = yr[c];
                v = go * (pp - yy) / fmaxf((1.f - pp) * pp, 1e-12f);      // (torch's BCE backward: see bce_rows_bwd_kernel)
            }
            dp[(size_t)r * a.Ce + c] = v;
        }
    }
}
// backward: blocks [0, ceil(R/4)) — one wave per row writes every probability gradient of its row; the remaining blocks fill d_cap
__global__ __launch_bounds__(256) void loss_tail_bwd_kernel(LossTailArgs a, int row_blocks) {
    const float g = a.dout[0];
    if ((int)blockIdx.x >= row_blocks) {
        const int i = ((int)blockIdx.x - row_blocks) * 256 + threadIdx.x;
        if (a.d_cap && i < a.n_cap) a.d_cap[i] = g;
        return;
    }
    const int lane = threadIdx.x & 63, r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= a.R) return;
    const int wdt = a.widths[r];
    const float* yr = a.align + (size_t)r * a.Ce;
    const float* ar = a.act + (size_t)r * a.Ca;
    if (a.Ca <= 64 * LT_CH) {       // all loads of the row first (see the forward)
        float yv[LT_CH], pv[LT_CH], rv[LT_CH];
#pragma unroll
        for (int k = 0; k < LT_CH; ++k) {
            const int c = lane + 64 * k;
            const bool in = c < a.Ca;
            yv[k] = in ? ar[c] : 0.f;
            pv[k] = (in && a.a_p && a.da_p) ? a.a_p[(size_t)r * a.Ca + c] : 0.5f;
            rv[k] = (in && a.r_a && a.dr_a) ? a.r_a[(size_t)r * a.Ca + c] : 0.5f;
        }
        loss_tail_bce_bwd(a, r, wdt, yr, g, lane);
        float f = 0.f;
#pragma unroll
        for (int k = 0; k < LT_CH; ++k) f = fmaxf(f, yv[k] == 1.0f ? 1.f : 0.f);
        const bool on = wave_max(f) != 0.f;
#pragma unroll
        for (int k = 0; k < LT_CH; ++k) {
            const int c = lane + 64 * k;
            if (c < a.Ca) {
                float l, d1 = 0.f, d2 = 0.f;
                if (on && a.a_p && a.da_p) asl_terms(pv[k], yv[k], a.gneg, a.gpos, a.clip, a.eps, l, d1);
                if (on && a.r_a && a.dr_a) asl_terms(rv[k], yv[k], a.gneg, a.gpos, a.clip, a.eps, l, d2);
                if (a.a_p && a.da_p) a.da_p[(size_t)r * a.Ca + c] = g * d1;
                if (a.r_a && a.dr_a) a.dr_a[(size_t)r * a.Ca + c] = g * a.lambda * d2;
            }
        }
        return;
    }
    const bool on = row_has_one(ar, a.Ca, lane);
    loss_tail_bce_bwd(a, r, wdt, yr, g, lane);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* p = which ? a.r_a : a.a_p;
        float* dp = which ? a.dr_a : a.da_p;
        const float go = on ? (which ? g * a.lambda : g) : 0.f;
        if (!p || !dp) continue;
        for (int c = lane; c < a.Ca; c += 64) {
            float l, d = 0.f;
            if (go != 0.f) asl_terms(p[(size_t)r * a.Ca + c], ar[c], a.gneg, a.gpos, a.clip, a.eps, l, d);
            dp[(size_t)r * a.Ca + c] = go * d;
        }
    }
}
__global__ __launch_bounds__(256) void clamp_labels_kernel(const int* __restrict__ in, int* __restrict__ out, int n, int vocab, int unk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] >= vocab ? unk : in[i];
}

// ---- LSTM cell: gates = gx + gh (i, f, g, o); inactive rows pass (h, c) through ---------------------------------
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                            const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                                            const float* __restrict__ active, float* __restrict__ h,
                                                            float* __restrict__ c, float* __restrict__ gates_act, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float gi = sigmoidf_(gx[g0] + gh[g0]);
    const float gf = sigmoidf_(gx[g0 + D] + gh[g0 + D]);
    const float gg = tanhf(gx[g0 + 2 * D] + gh[g0 + 2 * D]);
    const float go = sigmoidf_(gx[g0 + 3 * D] + gh[g0 + 3 * D]);
    gates_act[g0] = gi; gates_act[g0 + D] = gf; gates_act[g0 + 2 * D] = gg; gates_act[g0 + 3 * D] = go;
    const float cn = gf * c_prev[i] + gi * gg;
    const float hn = go * tanhf(cn);
    const float a = active[n];
    c[i] = a * cn + (1.f - a) * c_prev[i];
    h[i] = a * hn + (1.f - a) * h_prev[i];
}
// dgates is the gradient of the pre-activation gate sums (shared by gx and gh)
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                                            const float* __restrict__ gates_act, const float* __restrict__ c_prev,
                                                            const float* __restrict__ active, float* __restrict__ dgates,
                                                            float* __restrict__ dc_prev, float* __restrict__ dh_prev, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float gi = gates_act[g0], gf = gates_act[g0 + D], gg = gates_act[g0 + 2 * D], go = gates_act[g0 + 3 * D];
    const float a = active[n];
    const float dhn = a * dh[i], dcn_in = a * dc[i];
    const float cn = gf * c_prev[i] + gi * gg;
    const float tc = tanhf(cn);
    const float dcn = dcn_in + dhn * go * (1.f - tc * tc);
    dgates[g0] = dcn * gg * gi * (1.f - gi);
    dgates[g0 + D] = dcn * c_prev[i] * gf * (1.f - gf);
    dgates[g0 + 2 * D] = dcn * gi * (1.f - gg * gg);
    dgates[g0 + 3 * D] = dhn * tc * go * (1.f - go);
    dc_prev[i] = dcn * gf + (1.f - a) * dc[i];
    dh_prev[i] = (1.f - a) * dh[i];
}

// ---- input staging (reference: recursive_caption_dataset.py:389-416 feature windowing, :536-575 collate, train.py:91 H2D) ------------
// dst[r][:] = idx[r] >= 0 ? src[idx[r]][:] : 0  — the per-clip frame windows of a batch gathered out of an HBM-resident feature
// bank straight into the (S, N, Lv+Lt, F) layout the model consumes; HBM-bound (one read + one write of the gathered rows).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, int rows, int width4) {
    const int r = blockIdx.x;
    const int s = idx[r];
    float4* o = reinterpret_cast<float4*>(dst) + (size_t)r * width4;
    if (s < 0) {
        for (int c = threadIdx.x; c < width4; c += 256) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float4* in = reinterpret_cast<const float4*>(src) + (size_t)s * width4;
        for (int c = threadIdx.x; c < width4; c += 256) o[c] = in[c];
    }
}
// video half of input_ids / input_mask for every clip: [CLS] [VID]×n [SEP] [PAD]…  (n = valid frames; n < 0 marks a padded step:
// all PAD, mask 0).  ids / mask are (clips, L) with the text half [Lv, L) left untouched.
__global__ __launch_bounds__(256) void video_tokens_kernel(const int* __restrict__ n_valid, long long* __restrict__ ids,
                                                           float* __restrict__ mask, int clips, int Lv, int L, int cls_id, int vid_id,
                                                           int sep_id, int pad_id) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= clips * Lv) return;
    const int c = i / Lv, p = i - c * Lv, n = n_valid[c];
    int tok = pad_id; float m = 0.f;
    if (n >= 0) {
        if (p == 0) { tok = cls_id; m = 1.f; }
        else if (p <= n) { tok = vid_id; m = 1.f; }
        else if (p == n + 1) { tok = sep_id; m = 1.f; }
    }
    ids[(size_t)c * L + p] = tok;
    mask[(size_t)c * L + p] = m;
}

// ---- on-device training metrics (reference: src/train.py:32-49) — counters live in HBM, read back once per logging interval -------
// counters[0] += #rows with label != ignore ; counters[1] += #rows whose first-index arg-max equals the label  (cal_performance)
__global__ __launch_bounds__(256) void metric_argmax_kernel(const float* __restrict__ scores, int ld, int C,
                                                            const long long* __restrict__ labels, int ignore, double* __restrict__ counters) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int r = blockIdx.x;
    const long long lab = labels[r];
    if (lab == ignore) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < C; v += 256) {
        const float sc = scores[(size_t)r * ld + v];
        if (sc > best) { best = sc; bi = v; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < bi)) { best = sval[w]; bi = sidx[w]; }
        atomicAdd(&counters[0], 1.0);
        if ((long long)bi == lab) atomicAdd(&counters[1], 1.0);
    }
}
// counters[0] += Σ gold[prob > 0.5] ; counters[1] += Σ gold ; counters[2] += #(prob > 0.5)   (calculate_f1)
__global__ __launch_bounds__(256) void metric_f1_kernel(const float* __restrict__ prob, const float* __restrict__ gold, size_t n,
                                                        double* __restrict__ counters) {
    __shared__ float red[4];
    float c = 0.f, g = 0.f, p = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const bool on = prob[i] > 0.5f;
        const float gv = gold[i];
        c += on ? gv : 0.f; g += gv; p += on ? 1.f : 0.f;
    }
    c = block_sum_256(c, red); g = block_sum_256(g, red); p = block_sum_256(p, red);
    if (threadIdx.x == 0) { atomicAdd(&counters[0], (double)c); atomicAdd(&counters[1], (double)g); atomicAdd(&counters[2], (double)p); }
}

// Sequence forms used by the fused BiLSTM recurrence: the step's input-projection rows are gathered in place
// (gx_all[rows[n]]), and the backward adds the gradient arriving from the layer above (dh_out) to the recurrent one.
__global__ __launch_bounds__(256) void lstm_cell_fwd_idx_kernel(const float* __restrict__ gx_all, const int* __restrict__ rows,
                                                                const float* __restrict__ gh, const float* __restrict__ c_prev,
                                                                const float* __restrict__ h_prev, const float* __restrict__ active,
                                                                float* __restrict__ h, float* __restrict__ c,
                                                                float* __restrict__ gates_act, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float* gx = gx_all + (size_t)rows[n] * 4 * D + d;
    const float gi = sigmoidf_(gx[0] + gh[g0]);
    const float gf = sigmoidf_(gx[D] + gh[g0 + D]);
    const float gg = tanhf(gx[2 * D] + gh[g0 + 2 * D]);
    const float go = sigmoidf_(gx[3 * D] + gh[g0 + 3 * D]);
    gates_act[g0] = gi; gates_act[g0 + D] = gf; gates_act[g0 + 2 * D] = gg; gates_act[g0 + 3 * D] = go;
    const float cn = gf * c_prev[i] + gi * gg;
    const float hn = go * tanhf(cn);
    const float a = active[n];
    c[i] = a * cn + (1.f - a) * c_prev[i];
    h[i] = a * hn + (1.f - a) * h_prev[i];
}
__global__ __launch_bounds__(256) void lstm_cell_bwd_seq_kernel(const float* __restrict__ dh_out, const float* __restrict__ dh_rec,
                                                                const float* __restrict__ dc, const float* __restrict__ gates_act,
                                                                const float* __restrict__ c_prev, const float* __restrict__ active,
                                                                float* __restrict__ dgates, float* __restrict__ dc_prev,
                                                                float* __restrict__ dh_prev, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float gi = gates_act[g0], gf = gates_act[g0 + D], gg = gates_act[g0 + 2 * D], go = gates_act[g0 + 3 * D];
    const float a = active[n];
    const float dht = dh_out[i] + dh_rec[i];
    const float dhn = a * dht, dcn_in = a * dc[i];
    const float cn = gf * c_prev[i] + gi * gg;
    const float tc = tanhf(cn);
    const float dcn = dcn_in + dhn * go * (1.f - tc * tc);
    dgates[g0] = dcn * gg * gi * (1.f - gi);
    dgates[g0 + D] = dcn * c_prev[i] * gf * (1.f - gf);
    dgates[g0 + 2 * D] = dcn * gi * (1.f - gg * gg);
    dgates[g0 + 3 * D] = dhn * tc * go * (1.f - go);
    dc_prev[i] = dcn * gf + (1.f - a) * dc[i];
    dh_prev[i] = (1.f - a) * dht;
}

// Both directions of the BiLSTM at one time step in one launch: blockIdx.y = direction, per-direction pointers in the struct.
struct LstmPair {
    const float* gx[2]; const int* rows[2]; const float* gh[2]; const float* c_prev[2]; const float* h_prev[2]; const float* active;
    float* h[2]; float* c[2]; float* gates[2];
    // backward
    const float* dh_out[2]; const float* dh_rec[2]; const float* dc[2]; float* dgates[2]; float* dc_prev[2]; float* dh_prev[2];
    const float* dh_parts[2]; int n_parts;      // optional k-parts of the recurrent dgrad (n_parts slabs of N·D floats per direction)
};
__global__ __launch_bounds__(256) void lstm_pair_fwd_kernel(LstmPair a, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float* gx = a.gx[z] + (size_t)a.rows[z][n] * 4 * D + d;
    const float* gh = a.gh[z];
    const float gi = sigmoidf_(gx[0] + gh[g0]);
    const float gf = sigmoidf_(gx[D] + gh[g0 + D]);
    const float gg = tanhf(gx[2 * D] + gh[g0 + 2 * D]);
    const float go = sigmoidf_(gx[3 * D] + gh[g0 + 3 * D]);
    float* ga = a.gates[z];
    ga[g0] = gi; ga[g0 + D] = gf; ga[g0 + 2 * D] = gg; ga[g0 + 3 * D] = go;
    const float cp = a.c_prev[z][i];
    const float cn = gf * cp + gi * gg;
    const float hn = go * tanhf(cn);
    const float act = a.active[n];
    a.c[z][i] = act * cn + (1.f - act) * cp;
    a.h[z][i] = act * hn + (1.f - act) * a.h_prev[z][i];
}
__global__ __launch_bounds__(256) void lstm_pair_bwd_kernel(LstmPair a, int N, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    const size_t g0 = (size_t)n * 4 * D + d;
    const float* ga = a.gates[z];
    const float gi = ga[g0], gf = ga[g0 + D], gg = ga[g0 + 2 * D], go = ga[g0 + 3 * D];
    const float act = a.active[n];
    float dht = a.dh_out[z][i] + a.dh_rec[z][i];
    for (int p = 0; p < a.n_parts; ++p) dht += a.dh_parts[z][(size_t)p * N * D + i];       // part order: deterministic
    const float dci = a.dc[z][i], cp = a.c_prev[z][i];
    const float dhn = act * dht, dcn_in = act * dci;
    const float cn = gf * cp + gi * gg;
    const float tc = tanhf(cn);
    const float dcn = dcn_in + dhn * go * (1.f - tc * tc);
    float* dg = a.dgates[z];
    dg[g0] = dcn * gg * gi * (1.f - gi);
    dg[g0 + D] = dcn * cp * gf * (1.f - gf);
    dg[g0 + 2 * D] = dcn * gi * (1.f - gg * gg);
    dg[g0 + 3 * D] = dhn * tc * go * (1.f - go);
    a.dc_prev[z][i] = dcn * gf + (1.f - act) * dci;
    a.dh_prev[z][i] = (1.f - act) * dht;
}

// ---- greedy pick (src/translator.py:104-112): per sentence j take row j*lt+pos of the score matrix, suppress the UNK
// column (-1e10), first-index argmax over the row's C_j classes; the emitted stream keeps the extended id, the model
// side sees UNK for copied out-of-vocabulary words (id >= C_j - X_j).
__global__ __launch_bounds__(256) void greedy_pick_kernel(const float* __restrict__ scores, int ld, const int* __restrict__ row_c,
                                                          const int* __restrict__ row_x, int lt, int pos, int unk,
                                                          int* __restrict__ next_ext, int* __restrict__ next_model,
                                                          int* __restrict__ text_out, int* __restrict__ ext_out, int ld_out, int col) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int j = blockIdx.x, r = j * lt + pos;
    const int C = row_c[r], X = row_x[r];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < C; v += 256) {
        const float sc = v == unk ? -1e10f : scores[(size_t)r * ld + v];
        if (sc > best) { best = sc; bi = v; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        best = sval[0]; bi = sidx[0];
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < bi)) { best = sval[w]; bi = sidx[w]; }
        const int mod = bi >= C - X ? unk : bi;
        next_ext[j] = bi;
        next_model[j] = mod;
        if (text_out) {                  // the picked token is also the sentence's entry at position `col` of the two id matrices
            text_out[(size_t)j * ld_out + col] = mod;
            ext_out[(size_t)j * ld_out + col] = bi;
        }
    }
}

static inline unsigned grid1d(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }


// ---- rows between storage kinds in ONE launch (data movement: hi + lo of a split row is exact in fp32, fp32 → split is the two roundings
// that define the planes).  dst[r][:] = convert(src[idx ? idx[r] : r][:]); kinds: 0 fp32, 1 bf16, 2 split (lo plane `lo` columns behind).
// Replaces the ATen chains of the few places where rows change domain (decoder memory rows fp32 → split: 4 launches; decoder output
// split → fp32: 3; the [CLS] rows of the clip stream gathered as fp32: 5).
template <int SK, int DK>
__global__ __launch_bounds__(256) void rows_move_kernel(const void* __restrict__ src, int lds_, int slo, const int* __restrict__ idx,
                                                        void* __restrict__ dst, int ldd, int dlo, int W) {
    const int r = blockIdx.x;
    const size_t so = (size_t)(idx ? idx[r] : r) * lds_, dofs = (size_t)r * ldd;
    for (int c = threadIdx.x; c < W; c += 256) {
        float v;
        if constexpr (SK == 0) v = reinterpret_cast<const float*>(src)[so + c];
        else {
            const __bf16* b = reinterpret_cast<const __bf16*>(src);
            v = (float)b[so + c];
            if constexpr (SK == 2) v += (float)b[so + slo + c];
        }
        if constexpr (DK == 0) reinterpret_cast<float*>(dst)[dofs + c] = v;
        else {
            __bf16* b = reinterpret_cast<__bf16*>(dst);
            const __bf16 h = (__bf16)v;
            b[dofs + c] = h;
            if constexpr (DK == 2) b[dofs + dlo + c] = (__bf16)(v - (float)h);
        }
    }
}
// table[idx[r]][:] += rows[r][:] for a dense bf16 table and fp32 rows (fp32 add, one rounding; distinct idx: no atomics needed)
__global__ __launch_bounds__(256) void scatter_add_rows_bf16_kernel(const float* __restrict__ rows, const int* __restrict__ idx,
                                                                    __bf16* __restrict__ table, int ldt, int W) {
    const int r = blockIdx.x;
    const size_t to = (size_t)idx[r] * ldt;
    for (int c = threadIdx.x; c < W; c += 256) table[to + c] = (__bf16)((float)table[to + c] + rows[(size_t)r * W + c]);
}

// ---- rows of two (R', W) fp32 tables through two index lists in one launch (the BiLSTM's two directions, model.py:1022-1024: outputs are
// picked from the time-major states of both directions and summed; backward scatters the one output gradient into both and gathers both
// gate gradients).  MODE 0: out[r] = a[ia[r]] + b[ib[r]];  MODE 1: a[ia[r]] = b[ib[r]] = out[r] (distinct indices);  MODE 2: oa[r] = a[ia[r]],
// ob[r] = b[ib[r]] (out = oa, out2 = ob)
template <int MODE>
__global__ __launch_bounds__(256) void pair_rows_kernel(float* __restrict__ a, const int* __restrict__ ia, float* __restrict__ b,
                                                        const int* __restrict__ ib, float* __restrict__ out, float* __restrict__ out2, int W) {
    const int r = blockIdx.x;
    float* ar = a + (size_t)ia[r] * W;
    float* br = b + (size_t)ib[r] * W;
    float* o = out + (size_t)r * W;
    for (int c = threadIdx.x * 4; c < W; c += 1024) {
        if (MODE == 0) {
            const float4 x = *reinterpret_cast<const float4*>(ar + c), y = *reinterpret_cast<const float4*>(br + c);
            *reinterpret_cast<float4*>(o + c) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        } else if (MODE == 1) {
            const float4 v = *reinterpret_cast<const float4*>(o + c);
            *reinterpret_cast<float4*>(ar + c) = v;
            *reinterpret_cast<float4*>(br + c) = v;
        } else {
            *reinterpret_cast<float4*>(o + c) = *reinterpret_cast<const float4*>(ar + c);
            *reinterpret_cast<float4*>(out2 + (size_t)r * W + c) = *reinterpret_cast<const float4*>(br + c);
        }
    }
}

// out[r] = inv[r] >= 0 ? src[inv[r]] : 0 — packed rows back into a padded layout, zeros elsewhere, in one launch
__global__ __launch_bounds__(256) void rows_expand_kernel(const float* __restrict__ src, const int* __restrict__ inv, float* __restrict__ out, int W) {
    const int r = blockIdx.x, i = inv[r];
    float* o = out + (size_t)r * W;
    if (i < 0) { for (int c = threadIdx.x; c < W; c += 256) o[c] = 0.f; return; }
    const float* q = src + (size_t)i * W;
    for (int c = threadIdx.x; c < W; c += 256) o[c] = q[c];
}

// out = bf16(a + b) for a bf16 (may be null: a plain cast) and b fp32: the two gradients of a stream tensor that is also read as fp32 rows
__global__ __launch_bounds__(256) void add_cast_bf16_kernel(const __bf16* __restrict__ a, const float* __restrict__ b, __bf16* __restrict__ out,
                                                            size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (__bf16)((a ? (float)a[i] : 0.f) + b[i]);
}

extern "C" {

int svpc_rows_expand(const float* src, const int* inv, float* out, int n_rows, int W, hipStream_t s) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(rows_expand_kernel, dim3(n_rows), dim3(256), 0, s, src, inv, out, W);
    return svpc_check_launch("rows_expand");
}
int svpc_add_cast_bf16(const void* a, const float* b, void* out, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_cast_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, s, (const __bf16*)a, b, (__bf16*)out, n);
    return svpc_check_launch("add_cast_bf16");
}

// mode 0 gather + add, 1 scatter to both, 2 gather both (see pair_rows_kernel); W % 4 == 0, 16-byte aligned rows
int svpc_pair_rows(float* a, const int* ia, float* b, const int* ib, float* out, float* out2, int R, int W, int mode, hipStream_t s) {
    if (R == 0) return 0;
    SVPC_REQUIRE(W % 4 == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out) | ((uintptr_t)out2)) & 15) == 0 && mode >= 0 && mode <= 2 &&
                     (mode != 2 || out2 != nullptr),
                 "pair_rows: rows of W % 4 == 0 floats, 16-byte aligned; mode 0..2 (2 needs a second output)");
    if (mode == 0) hipLaunchKernelGGL(pair_rows_kernel<0>, dim3(R), dim3(256), 0, s, a, ia, b, ib, out, out2, W);
    else if (mode == 1) hipLaunchKernelGGL(pair_rows_kernel<1>, dim3(R), dim3(256), 0, s, a, ia, b, ib, out, out2, W);
    else hipLaunchKernelGGL(pair_rows_kernel<2>, dim3(R), dim3(256), 0, s, a, ia, b, ib, out, out2, W);
    return svpc_check_launch("pair_rows");
}

int svpc_act_bwd_t(const void* dy, const void* aux, void* dz, int dt, size_t n, int act, float p, unsigned site, const u64* seed,
                   hipStream_t s) {
    if (n == 0) return 0;
    SVPC_REQUIRE(p <= 0.f || seed != nullptr, "act_bwd: dropout needs a seed pointer");
    if (dt == 0) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid1d(n)), dim3(256), 0, s, (const float*)dy, (const float*)aux, (float*)dz, n,
                                    act, p, site, seed);
    else if (n % 8 == 0 && ((((uintptr_t)dy) | ((uintptr_t)aux) | ((uintptr_t)dz)) & 15) == 0)
        hipLaunchKernelGGL(act_bwd_bf16x8_kernel, dim3(grid1d(n / 8)), dim3(256), 0, s, (const uint4*)dy, (const uint4*)aux, (uint4*)dz, n / 8, act,
                           p, site, seed);
    else hipLaunchKernelGGL(act_bwd_kernel<__bf16>, dim3(grid1d(n)), dim3(256), 0, s, (const __bf16*)dy, (const __bf16*)aux, (__bf16*)dz, n,
                            act, p, site, seed);
    return svpc_check_launch("act_bwd");
}
int svpc_act_bwd(const float* dy, const float* aux, float* dz, size_t n, int act, float p, unsigned site, const u64* seed,
                 hipStream_t s) {
    return svpc_act_bwd_t(dy, aux, dz, 0, n, act, p, site, seed, s);
}
int svpc_rows_move(const void* src, int src_kind, int ld_src, int lo_src, const int* idx, void* dst, int dst_kind, int ld_dst, int lo_dst,
                   int R, int W, hipStream_t s) {
    if (R == 0 || W == 0) return 0;
    SVPC_REQUIRE(src_kind >= 0 && src_kind <= 2 && dst_kind >= 0 && dst_kind <= 2, "rows_move: storage kinds are 0 (fp32), 1 (bf16), 2 (split)");
#define RM(SK, DK) if (src_kind == SK && dst_kind == DK) hipLaunchKernelGGL((rows_move_kernel<SK, DK>), dim3(R), dim3(256), 0, s, src, ld_src, lo_src, idx, dst, ld_dst, lo_dst, W)
    RM(0, 0); RM(0, 1); RM(0, 2); RM(1, 0); RM(1, 1); RM(1, 2); RM(2, 0); RM(2, 1); RM(2, 2);
#undef RM
    return svpc_check_launch("rows_move");
}
int svpc_scatter_add_rows_bf16(const float* rows, const int* idx, void* table, int ld_table, int R, int W, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_bf16_kernel, dim3(R), dim3(256), 0, s, rows, idx, (__bf16*)table, ld_table, W);
    return svpc_check_launch("scatter_add_rows_bf16");
}
int svpc_add(const float* a, const float* b, float* c, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_kernel, dim3(grid1d(n)), dim3(256), 0, s, a, b, c, n);
    return svpc_check_launch("add");
}
int svpc_dropout_mask(float* out, size_t n, float p, unsigned site, const u64* seed, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid1d(n)), dim3(256), 0, s, out, n, p, site, seed);
    return svpc_check_launch("dropout_mask");
}
int svpc_attn_dropout_mask(float* out, size_t n_rows, int max_k, float p, unsigned site, const u64* seed, hipStream_t s) {
    if (n_rows == 0 || max_k <= 0) return 0;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(grid1d(n_rows * (size_t)max_k)), dim3(256), 0, s, out, n_rows, max_k, p, site, seed);
    return svpc_check_launch("attn_dropout_mask");
}
int svpc_gumbel_noise(float* out, size_t n, unsigned site, const u64* seed, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gumbel_noise_kernel, dim3(grid1d(n)), dim3(256), 0, s, out, n, site, seed);
    return svpc_check_launch("gumbel_noise");
}
int svpc_bump_seed(u64* seed, hipStream_t s) {
    hipLaunchKernelGGL(bump_seed_kernel, dim3(1), dim3(1), 0, s, seed);
    return svpc_check_launch("bump_seed");
}
int svpc_sum_all(const float* x, size_t n, float* out, float scale, hipStream_t s) {
    hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, s, x, n, out, scale);
    return svpc_check_launch("sum_all");
}
int svpc_fill_from(float* x, size_t n, const float* v, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid1d(n)), dim3(256), 0, s, x, n, v);
    return svpc_check_launch("fill_from");
}
int svpc_rownorm_fwd(const float* a, float* y, int R, int C, int mode, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(rownorm_fwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, a, y, R, C, mode);
    return svpc_check_launch("rownorm_fwd");
}
int svpc_rownorm_bwd(const float* dy, const float* y, const float* a, float* da, int R, int C, int mode, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, dy, y, a, da, R, C, mode);
    return svpc_check_launch("rownorm_bwd");
}
int svpc_span_mean_fwd(const float* x, const int* starts, const int* lens, const float* w, const float* add, const int* add_idx,
                       float* out, int G, int D, hipStream_t s) {
    if (G == 0) return 0;
    hipLaunchKernelGGL(span_mean_fwd_kernel, dim3(G), dim3(256), 0, s, x, starts, lens, w, add, add_idx, out, G, D);
    return svpc_check_launch("span_mean_fwd");
}
int svpc_span_mean_bwd(const float* dout, const int* starts, const int* lens, const float* w, float* dx, int G, int D,
                       hipStream_t s) {
    if (G == 0) return 0;
    hipLaunchKernelGGL(span_mean_bwd_kernel, dim3(G), dim3(256), 0, s, dout, starts, lens, w, dx, G, D);
    return svpc_check_launch("span_mean_bwd");
}
int svpc_scatter_add_rows(const float* dx, const int* idx, float* dtable, int R, int D, int pad_row, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(R), dim3(256), 0, s, dx, idx, dtable, R, D, pad_row);
    return svpc_check_launch("scatter_add_rows");
}
int svpc_bce_rows_fwd(const float* p, const float* y, const int* widths, float* out, int R, int C, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(bce_rows_fwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, p, y, widths, out, R, C);
    return svpc_check_launch("bce_rows_fwd");
}
int svpc_bce_rows_bwd(const float* dout, const float* p, const float* y, const int* widths, float* dp, int R, int C, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(bce_rows_bwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, dout, p, y, widths, dp, R, C);
    return svpc_check_launch("bce_rows_bwd");
}
int svpc_asl_rows_fwd(const float* p, const float* y, const float* active, float* out, int R, int C, float gneg, float gpos,
                      float clip, float eps, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(asl_rows_fwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, p, y, active, out, R, C, gneg, gpos, clip, eps);
    return svpc_check_launch("asl_rows_fwd");
}
int svpc_asl_rows_bwd(const float* dout, const float* p, const float* y, const float* active, float* dp, int R, int C, float gneg,
                      float gpos, float clip, float eps, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(asl_rows_bwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, dout, p, y, active, dp, R, C, gneg, gpos, clip, eps);
    return svpc_check_launch("asl_rows_bwd");
}
static LossTailArgs loss_tail_args(const float* cap_rows, int n_cap, const float* e_p, const float* align, const int* widths, int R, int Ce,
                                   const float* a_p, const float* act, int Ca, const float* r_e, const float* r_a, float lambda,
                                   float gneg, float gpos, float clip, float eps) {
    LossTailArgs a{};
    a.cap_rows = cap_rows; a.n_cap = n_cap; a.e_p = e_p; a.align = align; a.widths = widths; a.R = R; a.Ce = Ce; a.a_p = a_p; a.act = act;
    a.Ca = Ca; a.r_e = r_e; a.r_a = r_a; a.lambda = lambda; a.gneg = gneg; a.gpos = gpos; a.clip = clip; a.eps = eps;
    return a;
}
int svpc_loss_tail_ws_floats(int n_cap, int R) { return 3 * R + ceil_div(n_cap > 0 ? n_cap : 1, 1024) + 4; }
// workspace: svpc_loss_tail_ws_floats() floats of scratch; counter: one int the caller zeroed ONCE (the kernel leaves it at zero)
int svpc_loss_tail_fwd(const float* cap_rows, int n_cap, const float* e_p, const float* align, const int* widths, int R, int Ce,
                       const float* a_p, const float* act, int Ca, const float* r_e, const float* r_a, float lambda, float gneg,
                       float gpos, float clip, float eps, float* out5, float* workspace, int* counter, hipStream_t s) {
    SVPC_REQUIRE(R == 0 || (align && act && widths), "loss_tail: targets missing");
    SVPC_REQUIRE(workspace && counter, "loss_tail: workspace and counter required");
    LossTailArgs a = loss_tail_args(cap_rows, n_cap, e_p, align, widths, R, Ce, a_p, act, Ca, r_e, r_a, lambda, gneg, gpos, clip, eps);
    a.out = out5;
    const int row_blocks = ceil_div(R, 4), cap_blocks = ceil_div(n_cap > 0 ? n_cap : 1, 1024);
    hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(row_blocks + cap_blocks), dim3(256), 0, s, a, row_blocks, workspace, counter);
    return svpc_check_launch("loss_tail_fwd");
}
int svpc_loss_tail_bwd(const float* dout, int n_cap, const float* e_p, const float* align, const int* widths, int R, int Ce,
                       const float* a_p, const float* act, int Ca, const float* r_e, const float* r_a, float lambda, float gneg,
                       float gpos, float clip, float eps, float* d_cap, float* de_p, float* da_p, float* dr_e, float* dr_a,
                       hipStream_t s) {
    LossTailArgs a = loss_tail_args(nullptr, n_cap, e_p, align, widths, R, Ce, a_p, act, Ca, r_e, r_a, lambda, gneg, gpos, clip, eps);
    a.dout = dout; a.d_cap = d_cap; a.de_p = de_p; a.da_p = da_p; a.dr_e = dr_e; a.dr_a = dr_a;
    const int row_blocks = ceil_div(R, 4), cap_blocks = d_cap ? ceil_div(n_cap, 256) : 0;
    if (row_blocks + cap_blocks == 0) return 0;
    hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(row_blocks + cap_blocks), dim3(256), 0, s, a, row_blocks);
    return svpc_check_launch("loss_tail_bwd");
}
int svpc_row_any_eq1(const float* x, float* out, int R, int C, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(row_any_eq1_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, s, x, out, R, C);
    return svpc_check_launch("row_any_eq1");
}
// ---- token staging: up to GC_MAX (gather by int32 row index, or identity) + cast segments in one launch.  The training forward reads
// ids / masks / labels of all (step, video) rows (int64 or fp32, as the reference's loader hands them: train.py:91-112) and needs the
// clip rows' ids and masks and the sentence rows' ids, masks and labels as int32 / fp32 (model.py:1038-1042, 925-1015): nine cast and
// index_select launches otherwise.  dtype codes: 0 fp32, 1 int64, 2 int32.
constexpr int GC_MAX = 8;
struct GcSeg { const void* src; const int* idx; void* dst; int src_dt, dst_dt, n, block0; };
struct GcArgs { int n; GcSeg s[GC_MAX]; u64* bump; };
__global__ __launch_bounds__(256) void gather_cast_multi_kernel(GcArgs a) {
    // (the step's seed advances here when the staging is the first launch of the step: one launch fewer than a bump_seed of its own;
    // nothing in this launch reads the seed)
    if (a.bump && blockIdx.x == 0 && threadIdx.x == 0) a.bump[0] = a.bump[0] * 6364136223846793005ull + 1442695040888963407ull;
    int si = 0;
    while (si + 1 < a.n && (int)blockIdx.x >= a.s[si + 1].block0) ++si;
    const GcSeg& g = a.s[si];
    const int i = ((int)blockIdx.x - g.block0) * 256 + (int)threadIdx.x;
    if (i >= g.n) return;
    const size_t j = g.idx ? (size_t)g.idx[i] : (size_t)i;
    if (g.dst_dt == 0) {
        const float v = g.src_dt == 0 ? reinterpret_cast<const float*>(g.src)[j]
                      : g.src_dt == 1 ? (float)reinterpret_cast<const long long*>(g.src)[j] : (float)reinterpret_cast<const int*>(g.src)[j];
        reinterpret_cast<float*>(g.dst)[i] = v;
    } else {
        const int v = g.src_dt == 0 ? (int)reinterpret_cast<const float*>(g.src)[j]
                    : g.src_dt == 1 ? (int)reinterpret_cast<const long long*>(g.src)[j] : reinterpret_cast<const int*>(g.src)[j];
        reinterpret_cast<int*>(g.dst)[i] = v;
    }
}
struct HostGcSeg { const void* src; const int* idx; void* dst; int src_dt, dst_dt, n; };
int svpc_gather_cast_multi_seed(const void* segments, int n, u64* bump_seed, hipStream_t s);
int svpc_gather_cast_multi(const void* segments, int n, hipStream_t s) { return svpc_gather_cast_multi_seed(segments, n, nullptr, s); }
// the same, advancing the dropout / Gumbel seed of the step (svpc_bump_seed's update) in the same launch when bump_seed is not null
int svpc_gather_cast_multi_seed(const void* segments, int n, u64* bump_seed, hipStream_t s) {
    SVPC_REQUIRE(n >= 0 && n <= GC_MAX, "gather_cast_multi: at most 8 segments per launch");
    const HostGcSeg* h = reinterpret_cast<const HostGcSeg*>(segments);
    GcArgs a{};
    a.bump = bump_seed;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        SVPC_REQUIRE(h[i].src_dt >= 0 && h[i].src_dt <= 2 && (h[i].dst_dt == 0 || h[i].dst_dt == 2) && h[i].n >= 0,
                     "gather_cast_multi: dtypes are 0 fp32 / 1 int64 / 2 int32 (destination fp32 or int32)");
        if (h[i].n == 0) continue;
        GcSeg& g = a.s[a.n++];
        g.src = h[i].src; g.idx = h[i].idx; g.dst = h[i].dst; g.src_dt = h[i].src_dt; g.dst_dt = h[i].dst_dt; g.n = h[i].n; g.block0 = blocks;
        blocks += ceil_div(h[i].n, 256);
    }
    if (blocks == 0) {
        if (bump_seed) return svpc_bump_seed(bump_seed, s);
        return 0;
    }
    hipLaunchKernelGGL(gather_cast_multi_kernel, dim3(blocks), dim3(256), 0, s, a);
    return svpc_check_launch("gather_cast_multi");
}
int svpc_clamp_labels(const int* in, int* out, int n, int vocab, int unk, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(clamp_labels_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, out, n, vocab, unk);
    return svpc_check_launch("clamp_labels");
}
int svpc_greedy_pick_append(const float* scores, int ld, const int* row_c, const int* row_x, int n_sent, int lt, int pos, int unk,
                            int* next_ext, int* next_model, int* text_out, int* ext_out, int ld_out, int col, hipStream_t s) {
    if (n_sent == 0) return 0;
    SVPC_REQUIRE((text_out == nullptr) == (ext_out == nullptr) && (text_out == nullptr || (col >= 0 && col < ld_out)),
                 "greedy_pick_append: both id matrices or neither, column inside the matrices");
    hipLaunchKernelGGL(greedy_pick_kernel, dim3(n_sent), dim3(256), 0, s, scores, ld, row_c, row_x, lt, pos, unk, next_ext, next_model,
                       text_out, ext_out, ld_out, col);
    return svpc_check_launch("greedy_pick");
}
int svpc_greedy_pick(const float* scores, int ld, const int* row_c, const int* row_x, int n_sent, int lt, int pos, int unk,
                     int* next_ext, int* next_model, hipStream_t s) {
    return svpc_greedy_pick_append(scores, ld, row_c, row_x, n_sent, lt, pos, unk, next_ext, next_model, nullptr, nullptr, 0, 0, s);
}
int svpc_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, const float* h_prev, const float* active, float* h,
                       float* c, float* gates_act, int N, int D, hipStream_t s) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, s, gx, gh, c_prev, h_prev, active, h, c,
                       gates_act, N, D);
    return svpc_check_launch("lstm_cell_fwd");
}
int svpc_metric_argmax(const float* scores, int ld, int R, int C, const long long* labels, int ignore, double* counters, hipStream_t s) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(metric_argmax_kernel, dim3(R), dim3(256), 0, s, scores, ld, C, labels, ignore, counters);
    return svpc_check_launch("metric_argmax");
}
int svpc_metric_f1(const float* prob, const float* gold, size_t n, double* counters, hipStream_t s) {
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256);
    hipLaunchKernelGGL(metric_f1_kernel, dim3(blocks), dim3(256), 0, s, prob, gold, n, counters);
    return svpc_check_launch("metric_f1");
}
int svpc_gather_rows_f32(const float* src, const int* idx, float* dst, int rows, int width, hipStream_t s) {
    if (rows == 0) return 0;
    SVPC_REQUIRE(width % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "gather_rows: rows must be 16-byte aligned float4 multiples");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, src, idx, dst, rows, width / 4);
    return svpc_check_launch("gather_rows");
}
int svpc_video_tokens(const int* n_valid, long long* ids, float* mask, int clips, int Lv, int L, int cls_id, int vid_id, int sep_id,
                      int pad_id, hipStream_t s) {
    if (clips == 0) return 0;
    hipLaunchKernelGGL(video_tokens_kernel, dim3(ceil_div(clips * Lv, 256)), dim3(256), 0, s, n_valid, ids, mask, clips, Lv, L, cls_id,
                       vid_id, sep_id, pad_id);
    return svpc_check_launch("video_tokens");
}
int svpc_lstm_cell_fwd_idx(const float* gx_all, const int* rows, const float* gh, const float* c_prev, const float* h_prev,
                           const float* active, float* h, float* c, float* gates_act, int N, int D, hipStream_t s) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_idx_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, s, gx_all, rows, gh, c_prev, h_prev, active, h,
                       c, gates_act, N, D);
    return svpc_check_launch("lstm_cell_fwd_idx");
}
int svpc_lstm_cell_bwd_seq(const float* dh_out, const float* dh_rec, const float* dc, const float* gates_act, const float* c_prev,
                           const float* active, float* dgates, float* dc_prev, float* dh_prev, int N, int D, hipStream_t s) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_bwd_seq_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, s, dh_out, dh_rec, dc, gates_act, c_prev, active,
                       dgates, dc_prev, dh_prev, N, D);
    return svpc_check_launch("lstm_cell_bwd_seq");
}
// pair forms: arrays of 2 pointers (direction 0 = forward in time, 1 = reverse); `active` is shared (the same videos are alive)
int svpc_lstm_pair_fwd(const float* const* gx, const int* const* rows, const float* const* gh, const float* const* c_prev,
                       const float* const* h_prev, const float* active, float* const* h, float* const* c, float* const* gates, int N, int D,
                       hipStream_t s) {
    if (N == 0) return 0;
    LstmPair a{};
    for (int z = 0; z < 2; ++z) {
        a.gx[z] = gx[z]; a.rows[z] = rows[z]; a.gh[z] = gh[z]; a.c_prev[z] = c_prev[z]; a.h_prev[z] = h_prev[z]; a.h[z] = h[z];
        a.c[z] = c[z]; a.gates[z] = gates[z];
    }
    a.active = active;
    hipLaunchKernelGGL(lstm_pair_fwd_kernel, dim3(ceil_div(N * D, 256), 2), dim3(256), 0, s, a, N, D);
    return svpc_check_launch("lstm_pair_fwd");
}
int svpc_lstm_pair_bwd_parts(const float* const* dh_out, const float* const* dh_rec, const float* const* dh_parts, int n_parts,
                             const float* const* dc, const float* const* gates, const float* const* c_prev, const float* active,
                             float* const* dgates, float* const* dc_prev, float* const* dh_prev, int N, int D, hipStream_t s);
int svpc_lstm_pair_bwd(const float* const* dh_out, const float* const* dh_rec, const float* const* dc, const float* const* gates,
                       const float* const* c_prev, const float* active, float* const* dgates, float* const* dc_prev,
                       float* const* dh_prev, int N, int D, hipStream_t s) {
    return svpc_lstm_pair_bwd_parts(dh_out, dh_rec, nullptr, 0, dc, gates, c_prev, active, dgates, dc_prev, dh_prev, N, D, s);
}
// the same with the recurrent dgrad of the previous launch arriving as n_parts k-part slabs per direction (dh_parts[z]: n_parts × N·D)
int svpc_lstm_pair_bwd_parts(const float* const* dh_out, const float* const* dh_rec, const float* const* dh_parts, int n_parts,
                             const float* const* dc, const float* const* gates, const float* const* c_prev, const float* active,
                             float* const* dgates, float* const* dc_prev, float* const* dh_prev, int N, int D, hipStream_t s) {
    if (N == 0) return 0;
    LstmPair a{};
    a.n_parts = dh_parts ? n_parts : 0;
    for (int z = 0; z < 2 && a.n_parts; ++z) a.dh_parts[z] = dh_parts[z];
    for (int z = 0; z < 2; ++z) {
        a.dh_out[z] = dh_out[z]; a.dh_rec[z] = dh_rec[z]; a.dc[z] = dc[z]; a.gates[z] = const_cast<float*>(gates[z]);
        a.c_prev[z] = c_prev[z]; a.dgates[z] = dgates[z]; a.dc_prev[z] = dc_prev[z]; a.dh_prev[z] = dh_prev[z];
    }
    a.active = active;
    hipLaunchKernelGGL(lstm_pair_bwd_kernel, dim3(ceil_div(N * D, 256), 2), dim3(256), 0, s, a, N, D);
    return svpc_check_launch("lstm_pair_bwd");
}
int svpc_lstm_cell_bwd(const float* dh, const float* dc, const float* gates_act, const float* c_prev, const float* active,
                       float* dgates, float* dc_prev, float* dh_prev, int N, int D, hipStream_t s) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, s, dh, dc, gates_act, c_prev, active, dgates,
                       dc_prev, dh_prev, N, D);
    return svpc_check_launch("lstm_cell_bwd");
}

}  // extern "C"
