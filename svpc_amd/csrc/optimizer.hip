// Fused multi-tensor training-step tail: global-norm clip (train.py:141-142) → BertAdam (optimization.py:284-331:
// per-tensor clip to 1.0, m/v update without bias correction, decay added to the update, scheduled lr) → EMA
// (optimization.py:196-203).  The ≈190 parameter tensors are walked through a chunk table (tensor id, offset) so the
// whole step is 3 launches instead of ≈10³ eager ops; per-tensor and global norms are reduced in a fixed order.
// HBM-bound: 4 reads + 3 writes of the parameter bytes (+2 with EMA, +½ for the bf16 weight shadow).
#include "common.h"

// shadow: optional bf16 copy of the parameter (operand of the direct-to-LDS GEMMs), rewritten by the Adam kernel; shadow_lo: optional
// second plane bf16(p - shadow) of the bf16x3 mode's split weights (gemm_p8x3.hip)
struct TensorMeta { float* p; float* g; float* m; float* v; float* ema; long long n; float wd; int pad; __bf16* shadow; __bf16* shadow_lo; };

constexpr int OPT_CHUNK = 16384;

// partial[chunk] = Σ g² over the chunk
__global__ __launch_bounds__(256) void opt_sumsq_kernel(const TensorMeta* __restrict__ meta, const int* __restrict__ chunk_tid,
                                                        const long long* __restrict__ chunk_start, float* __restrict__ partial) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const TensorMeta t = meta[chunk_tid[c]];
    const long long s = chunk_start[c];
    const long long e = s + OPT_CHUNK < t.n ? s + OPT_CHUNK : t.n;
    float acc = 0.f;
    if ((e - s) == OPT_CHUNK && ((((unsigned long long)(t.g + s)) & 15ull) == 0)) {
        // a whole chunk: 16 independent 16-byte loads per thread, all requested before the first is consumed
        const float4* g4 = reinterpret_cast<const float4*>(t.g + s);
        float4 v[OPT_CHUNK / 4 / 256];
#pragma unroll
        for (int k = 0; k < OPT_CHUNK / 4 / 256; ++k) v[k] = g4[threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < OPT_CHUNK / 4 / 256; ++k) acc += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
    } else {
        for (long long i = s + threadIdx.x; i < e; i += 256) { const float g = t.g[i]; acc += g * g; }
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[c] = acc;
}
// norms_sq[t] = Σ partial over the tensor's chunks; norms_sq[n_tensors] = Σ_t norms_sq[t]
__global__ __launch_bounds__(256) void opt_norms_kernel(const float* __restrict__ partial, const int* __restrict__ tensor_chunk_off,
                                                        int n_tensors, float* __restrict__ norms_sq) {
    __shared__ float red[4];
    float tot = 0.f;
    for (int t = threadIdx.x; t < n_tensors; t += 256) {
        float s = 0.f;
        for (int c = tensor_chunk_off[t]; c < tensor_chunk_off[t + 1]; ++c) s += partial[c];
        norms_sq[t] = s;
        tot += s;
    }
    tot = block_sum_256(tot, red);
    if (threadIdx.x == 0) norms_sq[n_tensors] = tot;
}
// hyper: [lr_scheduled, ema_decay(<0 → off), max_global_norm(<=0 → off), max_tensor_norm(<=0 → off), b1, b2, eps]
__global__ __launch_bounds__(256) void opt_adam_kernel(const TensorMeta* __restrict__ meta, const int* __restrict__ chunk_tid,
                                                       const long long* __restrict__ chunk_start, const float* __restrict__ norms_sq,
                                                       int n_tensors, const float* __restrict__ hyper) {
    const int c = blockIdx.x;
    const int tid = chunk_tid[c];
    const TensorMeta t = meta[tid];
    const float lr = hyper[0], ema_decay = hyper[1], max_g = hyper[2], max_t = hyper[3], b1 = hyper[4], b2 = hyper[5], eps = hyper[6];
    float coef = 1.0f;
    if (max_g > 0.f) coef = fminf(1.0f, max_g / (sqrtf(norms_sq[n_tensors]) + 1e-6f));   // clip_grad_norm_ (global)
    if (max_t > 0.f) {
        const float tn = sqrtf(norms_sq[tid]) * coef;                                        // norm after the global clip
        coef *= fminf(1.0f, max_t / (tn + 1e-6f));                                           // BertAdam per-tensor clip
    }
    const long long s = chunk_start[c];
    const long long e = s + OPT_CHUNK < t.n ? s + OPT_CHUNK : t.n;
    // a whole chunk of 16-byte aligned state: 16-byte accesses, the loads of four groups (16–20 per thread) requested before the first
    // is consumed.  (Element by element every iteration is a memory round trip of its own: the compiler cannot move the next
    // iteration's loads above this one's stores — the state arrays may alias for all it knows — and a lane moves 4 bytes per access.)
    const bool use_ema = t.ema && ema_decay >= 0.f;
    const unsigned long long al = (unsigned long long)(t.g + s) | (unsigned long long)(t.m + s) | (unsigned long long)(t.v + s) |
                                  (unsigned long long)(t.p + s) | (use_ema ? (unsigned long long)(t.ema + s) : 0ull) |
                                  ((t.shadow ? (unsigned long long)(t.shadow + s) : 0ull) << 1) |
                                  ((t.shadow && t.shadow_lo ? (unsigned long long)(t.shadow_lo + s) : 0ull) << 1);
    if ((e - s) == OPT_CHUNK && (al & 15ull) == 0) {
        constexpr int NB = 4, PER = OPT_CHUNK / 4 / 256 / NB;      // 4 batches of 4 float4 per thread
        float4* __restrict__ g4 = reinterpret_cast<float4*>(t.g + s);
        float4* __restrict__ m4 = reinterpret_cast<float4*>(t.m + s);
        float4* __restrict__ v4 = reinterpret_cast<float4*>(t.v + s);
        float4* __restrict__ p4 = reinterpret_cast<float4*>(t.p + s);
        float4* __restrict__ e4 = reinterpret_cast<float4*>(t.ema + s);
        uint2* __restrict__ sh = reinterpret_cast<uint2*>(t.shadow + s);
        uint2* __restrict__ sl = reinterpret_cast<uint2*>(t.shadow_lo + s);
        for (int b = 0; b < NB; ++b) {
            float4 g[PER], m[PER], v[PER], p[PER], em[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + 256 * (b * PER + k);
                g[k] = g4[i]; m[k] = m4[i]; v[k] = v4[i]; p[k] = p4[i];
                if (use_ema) em[k] = e4[i];
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + 256 * (b * PER + k);
                float gg[4] = {g[k].x, g[k].y, g[k].z, g[k].w}, mm[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
                float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, pp[4] = {p[k].x, p[k].y, p[k].z, p[k].w};
                float ee[4] = {0.f, 0.f, 0.f, 0.f};
                if (use_ema) { ee[0] = em[k].x; ee[1] = em[k].y; ee[2] = em[k].z; ee[3] = em[k].w; }
                __bf16 hi[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {                 // the same operations in the same order as the scalar form below
                    const float gj = gg[j] * coef;
                    mm[j] = b1 * mm[j] + (1.f - b1) * gj;
                    vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
                    float upd = mm[j] / (sqrtf(vv[j]) + eps);
                    if (t.wd > 0.f) upd += t.wd * pp[j];
                    pp[j] -= lr * upd;
                    hi[j] = (__bf16)pp[j];
                    lo[j] = (__bf16)(pp[j] - (float)hi[j]);
                    if (use_ema) ee[j] = (1.f - ema_decay) * pp[j] + ema_decay * ee[j];
                }
                m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
                v4[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
                if (t.shadow) {
                    union { __bf16 h[4]; uint2 u; } ph, pl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ph.h[j] = hi[j]; pl.h[j] = lo[j]; }
                    sh[i] = ph.u;
                    if (t.shadow_lo) sl[i] = pl.u;
                }
                if (use_ema) e4[i] = make_float4(ee[0], ee[1], ee[2], ee[3]);
            }
        }
        return;
    }
    for (long long i = s + threadIdx.x; i < e; i += 256) {
        const float g = t.g[i] * coef;
        const float m = b1 * t.m[i] + (1.f - b1) * g;
        const float v = b2 * t.v[i] + (1.f - b2) * g * g;
        float p = t.p[i];
        float upd = m / (sqrtf(v) + eps);
        if (t.wd > 0.f) upd += t.wd * p;
        p -= lr * upd;
        t.m[i] = m; t.v[i] = v; t.p[i] = p;
        if (t.shadow) {
            const __bf16 ph = (__bf16)p;
            t.shadow[i] = ph;
            if (t.shadow_lo) t.shadow_lo[i] = (__bf16)(p - (float)ph);
        }
        if (t.ema && ema_decay >= 0.f) t.ema[i] = (1.f - ema_decay) * p + ema_decay * t.ema[i];
    }
}
__global__ __launch_bounds__(256) void opt_zero_kernel(const TensorMeta* __restrict__ meta, const int* __restrict__ chunk_tid,
                                                       const long long* __restrict__ chunk_start) {
    const int c = blockIdx.x;
    const TensorMeta t = meta[chunk_tid[c]];
    const long long s = chunk_start[c];
    const long long e = s + OPT_CHUNK < t.n ? s + OPT_CHUNK : t.n;
    for (long long i = s + threadIdx.x; i < e; i += 256) t.g[i] = 0.f;
}

extern "C" {

int svpc_opt_chunk(void) { return OPT_CHUNK; }
int svpc_opt_meta_bytes(void) { return (int)sizeof(TensorMeta); }

// norms_sq: n_tensors+1 floats (last = global Σ g²); partial: n_chunks floats
int svpc_opt_step(const void* meta, const int* chunk_tid, const long long* chunk_start, const int* tensor_chunk_off, int n_tensors,
                  int n_chunks, float* partial, float* norms_sq, const float* hyper, hipStream_t s) {
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(n_chunks), dim3(256), 0, s, (const TensorMeta*)meta, chunk_tid, chunk_start, partial);
    int rc = svpc_check_launch("opt_sumsq");
    if (rc) return rc;
    hipLaunchKernelGGL(opt_norms_kernel, dim3(1), dim3(256), 0, s, partial, tensor_chunk_off, n_tensors, norms_sq);
    rc = svpc_check_launch("opt_norms");
    if (rc) return rc;
    hipLaunchKernelGGL(opt_adam_kernel, dim3(n_chunks), dim3(256), 0, s, (const TensorMeta*)meta, chunk_tid, chunk_start, norms_sq,
                       n_tensors, hyper);
    return svpc_check_launch("opt_adam");
}
int svpc_opt_zero_grad(const void* meta, const int* chunk_tid, const long long* chunk_start, int n_chunks, hipStream_t s) {
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(opt_zero_kernel, dim3(n_chunks), dim3(256), 0, s, (const TensorMeta*)meta, chunk_tid, chunk_start);
    return svpc_check_launch("opt_zero");
}

}  // extern "C"
