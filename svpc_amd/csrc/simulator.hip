// Entity-state recurrence of the visual simulator / textual re-simulator (reference: src/rtransformer/model.py:792-820,
// Eqs. (2)-(7)).  Everything that depends only on the step vector (action selector, verb mixture f̄, ĥ = ReLU(W1 v),
// q = W2[ĥ;a], c = softmax(W3 ĥ), w = W4 f̄) is hoisted into batched GEMMs by the host; what is left is sequential in
// the step index t and tiny, so ONE workgroup per video walks its S_b steps with the entity matrix E (E_b × D) resident
// in LDS — no launches inside the recurrence:
//     e  = sigmoid(E q_t)                     alpha = c0·e + c1·e_prev          ē = (alpha/Σalpha)ᵀ E
//     k  = ReLU(w_t · ē)                      E ← alpha kᵀ + (1-alpha) ⊙ E      e_prev ← e
// Backward walks t in reverse with the gradient of E carried in registers (thread-owned columns), recomputing alpha, k
// from the saved e / ē / E_t; the 2·E_b+1 per-step scalars (dα pieces, dw) are reduced once per step through LDS.
#include "common.h"
#include <stdlib.h>

constexpr int SIM_EMAX = 32;

struct SimArgs {
    const float* q; const float* c; const float* w4f; const float* E0;
    const int* step_off; const int* step_len; const int* ent_off; const int* ent_len;
    float* e_out; float* ebar; float* eall; int e_max; int D;
    // backward
    const float* de; const float* debar; const float* deall;
    float* dq; float* dc; float* dw4f; float* dE0;
};

__global__ __launch_bounds__(1024) void sim_recur_fwd_kernel(SimArgs a) {
    extern __shared__ float smem[];
    const int b = blockIdx.x, D = a.D, em = a.e_max;
    const int s0 = a.step_off[b], S = a.step_len[b], e0 = a.ent_off[b], E = a.ent_len[b];
    float* Es = smem;                 // em × D
    float* ev = Es + (size_t)em * D;  // 32
    float* prev = ev + SIM_EMAX;      // 32
    float* al = prev + SIM_EMAX;      // 32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NT = blockDim.x, NW = blockDim.x >> 6;
    // (eight pieces of the initial entity matrix in flight per thread: with the load inside a one-piece loop body every piece was a
    // memory round trip of its own — E of them in a row before the recurrence starts)
    for (int i0 = 0; i0 < E * D; i0 += 8 * NT) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = a.E0[(size_t)e0 * D + min(i0 + u * NT + (int)threadIdx.x, E * D - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * NT + (int)threadIdx.x;
            if (i < E * D) Es[i] = v8[u];
        }
    }
    float my_prev = 0.f;              // thread e < 32: e_{t-1}[e] (read and replaced by the same thread: no LDS, no barrier)
    (void)prev;
    // The recurrence is a chain of dependent steps on ONE workgroup per video: what a step costs is its exposed latency.  The step
    // vector q_t (and c_t, w_t) do not depend on the state, so the values of step t+1 are requested at the top of step t and land under
    // its arithmetic (D <= 768: 12 values per lane; wider rows read q in the loop as before).
    constexpr int QN = 12;
    const bool pre = D <= 64 * QN;
    float qn[QN], cn0 = 0.f, cn1 = 0.f, wn = 0.f;
    auto fetch = [&](int jn) {
        const float* qp = a.q + (size_t)jn * D;
#pragma unroll
        for (int k = 0; k < QN; ++k) qn[k] = (lane + 64 * k) < D ? qp[lane + 64 * k] : 0.f;
        cn0 = a.c[(size_t)jn * 3]; cn1 = a.c[(size_t)jn * 3 + 1]; wn = a.w4f[jn];
    };
    if (pre && S > 0) fetch(s0);
    __syncthreads();
    for (int t = 0; t < S; ++t) {
        const int j = s0 + t;
        const float* qj = a.q + (size_t)j * D;
        float qc[QN];
#pragma unroll
        for (int k = 0; k < QN; ++k) qc[k] = qn[k];
        const float pc0 = cn0, pc1 = cn1, pw = wn;
        if (pre && t + 1 < S) fetch(j + 1);
        for (int e = wave; e < E; e += NW) {
            float dot = 0.f;
            if (pre) {
#pragma unroll
                for (int k = 0; k < QN; ++k)
                    if (lane + 64 * k < D) dot += Es[(size_t)e * D + lane + 64 * k] * qc[k];
            } else {
                for (int d = lane; d < D; d += 64) dot += Es[(size_t)e * D + d] * qj[d];
            }
            dot = wave_sum(dot);
            if (lane == 0) ev[e] = sigmoidf_(dot);
        }
        __syncthreads();
        const float c0 = pre ? pc0 : a.c[(size_t)j * 3], c1 = pre ? pc1 : a.c[(size_t)j * 3 + 1];
        if (threadIdx.x < SIM_EMAX) {
            const int e = threadIdx.x;
            const float ev_e = e < E ? ev[e] : 0.f;
            al[e] = e < E ? c0 * ev_e + c1 * my_prev : 0.f;
            my_prev = ev_e;
            if (e < em) a.e_out[(size_t)j * em + e] = ev_e;
        }
        __syncthreads();
        float Z = 0.f;
        for (int e = 0; e < E; ++e) Z += al[e];
        const float invZ = 1.0f / Z;
        const float w = pre ? pw : a.w4f[j];
        for (int d = threadIdx.x; d < D; d += NT) {
            float eb = 0.f;
            for (int e = 0; e < E; ++e) eb += al[e] * Es[(size_t)e * D + d];
            eb *= invZ;
            a.ebar[(size_t)j * D + d] = eb;
            const float k = fmaxf(w * eb, 0.f);
            for (int e = 0; e < em; ++e) {
                float v = 0.f;
                if (e < E) {
                    v = al[e] * k + (1.f - al[e]) * Es[(size_t)e * D + d];
                    Es[(size_t)e * D + d] = v;
                }
                a.eall[((size_t)j * em + e) * D + d] = v;
            }
        }
        __syncthreads();          // the state is updated (and al / ev read) before the next step's dot products overwrite ev
    }
}

// NT threads, CPT = columns per thread (D <= NT*CPT).  The recurrence is a chain of S_b dependent steps per video and only
// n_videos workgroups exist, so the step latency is what counts: a wide workgroup (one column per thread at D = 768) keeps the
// per-thread serial work and the number of outstanding loads per thread small.
// The state before step t (saved by the forward) and the upstream gradient of the state after step t are E_b·D contiguous floats
// each.  DMA = true (D % 256 == 0): both are brought into LDS by global_load_lds, double-buffered — the copy for step t-1 is issued
// at the top of step t and lands under its arithmetic, so no global load sits inside the per-entity loops (conditional loads
// there serialise into one memory round trip per entity).  DMA = false: the same images are staged synchronously.
typedef const void __attribute__((address_space(1))) * sim_gptr;
typedef void __attribute__((address_space(3))) * sim_lptr;

template <int NT>
__device__ __forceinline__ void sim_dma(const float* __restrict__ src, float* __restrict__ dst, int n_floats, int wave, int lane) {
    const int pieces = n_floats >> 8;                 // 1 KiB = 256 floats per wave-instruction
    for (int i = wave; i < pieces; i += NT / 64)
        __builtin_amdgcn_global_load_lds((sim_gptr)(src + (size_t)i * 256 + lane * 4), (sim_lptr)(dst + (size_t)i * 256), 16, 0, 0);
}

// Barrier between two LDS phases of a step.  __syncthreads() is a workgroup fence + barrier, and with global_load_lds copies in flight the
// fence waits for EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)): the next step's images and operands, requested at the
// top of the step so that they land under its arithmetic, were waited for at the first barrier behind them — a memory round trip on
// the chain of every step.  The phases inside a step exchange data through ordinary ds_write / ds_read only: lgkmcnt(0) is enough.
__device__ __forceinline__ void sim_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 2: DMA double-buffered (4 images in LDS) · MODE 1: both images staged synchronously (2 images) · MODE 0: only the state is
// staged, the upstream gradient is read from HBM inside the loops (1 image: up to 32 entities × 768) · MODE 3 (the 1-image case when
// D % 256 == 0: 17-32 entities at D = 768, what the reference's batches with up to 31 ingredients need): the state image by DMA and the
// thread's column of the upstream gradient in registers, every entity requested at the top of the step — ONE memory round trip per step
// where MODE 0 made one per entity inside the loop (conditional loads serialise) plus a staged copy: 850 → ≈120 µs at 16 steps × 31 entities
// EM = compile-time bound on the entities per video (16 or 32): the per-entity scalars and gradient columns are register arrays
// unrolled to EM, so the smaller bound halves the register pressure and the predicated code when the batch allows it.
template <int CPT, int NT, int MODE, int EM>
__global__ __launch_bounds__(NT) void sim_recur_bwd_kernel(SimArgs a) {
    constexpr int NW = NT / 64;
    constexpr bool DMA = MODE == 2;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int b = blockIdx.x, D = a.D, em = a.e_max;
    const int s0 = a.step_off[b], S = a.step_len[b], e0 = a.ent_off[b], E = a.ent_len[b];
    const int img = ((em * D + 255) / 256) * 256;  // floats per image, 1-KiB aligned
    float* Ebuf = smem;                            // DMA: [2][img] state BEFORE step t;  else [img]
    float* Ubuf = Ebuf + (DMA ? 2 : 1) * img;      // same shape: upstream gradient of the state AFTER step t (zeros if absent)
    constexpr bool ONE = MODE == 0 || MODE == 3;      // one image: the state only
    float* red = Ubuf + (ONE ? 0 : (DMA ? 2 : 1)) * img;       // (2*32+1) × NW wave partials
    float* sc = red + (2 * EM + 1) * NW;     // scalars: ds[32]
    float* dprev = sc + EM;                  // 32: gradient flowing into e_{t-1} through "prev"
    float* tot = dprev + EM;                 // 2·EM + 1 block totals of the wave partials in `red`
    // the saved e rows and the upstream gradient of e, per step: e_t is needed at steps t (as e) and t+1 (as e_prev), so a ring of three
    // rows holds them; thread e < EM requests row t-2 (and de of t-1) at the top of step t and stores it at the end of the step.  Read
    // from global memory at the top of every step (3·E loads behind branches, then a wait for everything outstanding — the prefetched
    // operands of the next step included) they put a memory round trip on the chain of every step.
    float* eoR = tot + 2 * EM + 1;           // [3][EM]
    float* deR = eoR + 3 * EM;               // [2][EM]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < EM && S > 0) {
        const int e = threadIdx.x;
        const bool ok = e < E;
        const int ec = min(e, em - 1);
        eoR[((S - 1) % 3) * EM + e] = ok ? a.e_out[(size_t)(s0 + S - 1) * em + ec] : 0.f;
        eoR[((S + 1) % 3) * EM + e] = (ok && S > 1) ? a.e_out[(size_t)(s0 + max(S - 2, 0)) * em + ec] : 0.f;       // (S-2) mod 3
        deR[((S - 1) & 1) * EM + e] = (ok && a.de) ? a.de[(size_t)(s0 + S - 1) * em + ec] : 0.f;
    }

    float dE[EM][CPT];
#pragma unroll
    for (int e = 0; e < EM; ++e)
#pragma unroll
        for (int u = 0; u < CPT; ++u) dE[e][u] = 0.f;
    float my_dprev = 0.f;                    // wave 0, lane e: gradient flowing into e_{t-1}[e] through "prev"
    (void)dprev; (void)tot;
    if (!ONE && !a.deall)
        for (int i = threadIdx.x; i < (DMA ? 2 : 1) * img; i += NT) Ubuf[i] = 0.f;

    auto state_before = [&](int t) -> const float* {
        return t == 0 ? a.E0 + (size_t)e0 * D : a.eall + (size_t)(s0 + t - 1) * em * D;
    };
    if (DMA && S > 0) {
        sim_dma<NT>(state_before(S - 1), Ebuf, E * D, wave, lane);
        if (a.deall) sim_dma<NT>(a.deall + (size_t)(s0 + S - 1) * em * D, Ubuf, E * D, wave, lane);
    }

    // Per-step operands that do not depend on the recurrence (c_t, w_t, the saved e rows, ē_t, q_t, the upstream gradients of e and ē):
    // the values of step t-1 are requested at the top of step t and land under its arithmetic — the chain of S_b dependent steps
    // is what this kernel costs, so no load may sit on it.
    float n_c0 = 0.f, n_c1 = 0.f, n_w = 0.f, n_eb[CPT], n_deb[CPT], n_q[CPT];     // (the 3·E saved-e scalars too: spills at 168 VGPRs, measured slower)
    auto fetch = [&](int tt) {
        const int jj = s0 + tt;
        n_c0 = a.c[(size_t)jj * 3]; n_c1 = a.c[(size_t)jj * 3 + 1]; n_w = a.w4f[jj];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int d = threadIdx.x + NT * u;
            n_eb[u] = d < D ? a.ebar[(size_t)jj * D + d] : 0.f;
            n_deb[u] = (d < D && a.debar) ? a.debar[(size_t)jj * D + d] : 0.f;
            n_q[u] = d < D ? a.q[(size_t)jj * D + d] : 0.f;
        }
    };
    if (S > 0) fetch(S - 1);
    float upv[MODE == 3 ? EM : 1][CPT];
#pragma unroll
    for (int e = 0; e < (MODE == 3 ? EM : 1); ++e)
#pragma unroll
        for (int u = 0; u < CPT; ++u) upv[e][u] = 0.f;
    // the results of a step (dq row, dc, dw) leave at the START of the next step: a store issued at the end of a step is still on its way
    // when the next step's top waits for its images (vmcnt counts stores too) — held back one step, everything the wait sees is a step old
    float held_dq[CPT], held_dc0 = 0.f, held_dc1 = 0.f, held_dw = 0.f;
    int held_j = -1;
#pragma unroll
    for (int u = 0; u < CPT; ++u) held_dq[u] = 0.f;
    auto flush_held = [&]() {
        if (held_j < 0) return;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int d = threadIdx.x + NT * u;
            if (d < D) a.dq[(size_t)held_j * D + d] = held_dq[u];
        }
        if (threadIdx.x == 0) {
            a.dc[(size_t)held_j * 3] = held_dc0; a.dc[(size_t)held_j * 3 + 1] = held_dc1; a.dc[(size_t)held_j * 3 + 2] = 0.f;
            a.dw4f[held_j] = held_dw;
        }
    };
    for (int t = S - 1; t >= 0; --t) {
        const int j = s0 + t;
        const int cur = DMA ? ((S - 1 - t) & 1) : 0;
        float* Es = Ebuf + cur * img;
        const float* Us = Ubuf + cur * img;
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step t have landed
            __syncthreads();                                       // everyone's have; everyone is done with the other buffers
            flush_held();
            if (t > 0) {
                sim_dma<NT>(state_before(t - 1), Ebuf + (cur ^ 1) * img, E * D, wave, lane);
                if (a.deall) sim_dma<NT>(a.deall + (size_t)(j - 1) * em * D, Ubuf + (cur ^ 1) * img, E * D, wave, lane);
            }
        } else if (MODE == 3) {
            __syncthreads();                                       // everyone is done with the image of step t+1
            sim_dma<NT>(state_before(t), Es, E * D, wave, lane);
            if (a.deall) {
#pragma unroll
                for (int e = 0; e < EM; ++e)
#pragma unroll
                    for (int u = 0; u < CPT; ++u)                   // (rows / columns past the end re-read the last: unconditional, all in flight)
                        upv[e][u] = a.deall[((size_t)j * em + max(min(e, E - 1), 0)) * D + min((int)threadIdx.x + NT * u, D - 1)];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            __syncthreads();
            const float* sb = state_before(t);
            for (int i = threadIdx.x; i < E * D; i += NT) {
                Es[i] = sb[i];
                if (MODE != 0 && a.deall) Ubuf[i] = a.deall[(size_t)j * em * D + i];
            }
            __syncthreads();
        }
        const float c0 = n_c0, c1 = n_c1, w = n_w;
        float al[EM], ebv[CPT], debv[CPT], qv[CPT];
        float Z = 0.f;
        const float* er = eoR + (t % 3) * EM;
        const float* pr = eoR + ((t + 2) % 3) * EM;        // (t-1) mod 3
        const float* dr = deR + (t & 1) * EM;
        const float pmul = t > 0 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            al[e] = e < E ? c0 * er[e] + c1 * pmul * pr[e] : 0.f;
            Z += al[e];
        }
        // rows for the next steps (thread e < EM; clamped addresses, no branch around the loads): e of step t-2, de of step t-1
        float nx_e = 0.f, nx_de = 0.f;
        if (threadIdx.x < EM) {
            const int ec = min((int)threadIdx.x, em - 1);
            nx_e = a.e_out[(size_t)(s0 + max(t - 2, 0)) * em + ec];
            nx_de = a.de ? a.de[(size_t)(s0 + max(t - 1, 0)) * em + ec] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) { ebv[u] = n_eb[u]; debv[u] = n_deb[u]; qv[u] = n_q[u]; }
        if (t > 0) fetch(t - 1);
        const float invZ = 1.0f / Z;
        // per-thread partials of A1[e], dab[e], dw
        float pA[EM], pB[EM];
        float pw = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e) { pA[e] = 0.f; pB[e] = 0.f; }
        float debar_l[CPT], kq[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int d = threadIdx.x + NT * u;
            debar_l[u] = 0.f; kq[u] = 0.f;
            if (d < D) {
                const float eb = ebv[u];
                const float k = fmaxf(w * eb, 0.f);
                float dk = 0.f;
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    if (e < E) {
                        const float up = MODE == 3 ? upv[MODE == 3 ? e : 0][u]
                                                   : (MODE != 0 ? Us[(size_t)e * D + d] : (a.deall ? a.deall[((size_t)j * em + e) * D + d] : 0.f));
                        const float g = dE[e][u] + up;
                        const float Ev = Es[(size_t)e * D + d];
                        pA[e] += g * (k - Ev);
                        dk += g * al[e];
                        dE[e][u] = g * (1.f - al[e]);
                    }
                }
                const float dpre = k > 0.f ? dk : 0.f;
                pw += dpre * eb;
                const float deb = dpre * w + debv[u];
                debar_l[u] = deb;
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    if (e < E) {
                        pB[e] += deb * Es[(size_t)e * D + d];
                        dE[e][u] += al[e] * invZ * deb;
                    }
                }
                kq[u] = qv[u];
            }
        }
        // block reduction of 2·EM + 1 scalars: one reduce-scatter butterfly over the wave for 32 of them at a time (lane l ends with value
        // l >> 1), the wave partials to LDS
        if constexpr (EM == 16) {
            float v32[32];
#pragma unroll
            for (int e = 0; e < 16; ++e) { v32[e] = pA[e]; v32[16 + e] = pB[e]; }
            const float u_ = wave_reduce_scatter32(v32, lane);
            if ((lane & 1) == 0) red[(lane >> 1) * NW + wave] = u_;            // rows 0..15: A1[e], rows 16..31: dab[e]
        } else {
            const float ua = wave_reduce_scatter32(pA, lane), ub = wave_reduce_scatter32(pB, lane);
            if ((lane & 1) == 0) { red[(lane >> 1) * NW + wave] = ua; red[(EM + (lane >> 1)) * NW + wave] = ub; }
        }
        pw = wave_sum(pw);
        if (lane == 0) red[(2 * EM) * NW + wave] = pw;
        sim_lds_barrier();
        // wave 0 finishes the scalar algebra, lane e for entity e (every thread used to do all of it redundantly: ≈200 vector instructions
        // per wave and step on a kernel that is bound by instruction issue — 12 waves on 4 SIMDs): the NW wave partials of A1[e] and dab[e]
        // in wave order, the three sums over the entities as wave reductions, ds[e] to LDS for everyone
        float dc0 = 0.f, dc1 = 0.f, dw = 0.f;
        if (wave == 0) {
            const int e = lane;
            const bool on = e < E;                       // (E <= EM <= 32 < 64)
            const int ec = min(e, EM - 1);
            float A1 = 0.f, dabv = 0.f, dwv = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NW; ++w_) {
                A1 += red[ec * NW + w_];
                dabv += red[(EM + ec) * NW + w_];
                dwv += red[(2 * EM) * NW + w_];
            }
            const float ev_e = er[ec], pv_e = pmul * pr[ec];
            const float al_e = on ? c0 * ev_e + c1 * pv_e : 0.f;
            const float mix = wave_sum(on ? dabv * al_e * invZ : 0.f);
            const float dal_e = on ? A1 + (dabv - mix) * invZ : 0.f;
            dc0 = wave_sum(dal_e * ev_e);
            dc1 = wave_sum(dal_e * pv_e);
            dw = dwv;
            const float de_tot = c0 * dal_e + dr[ec] + my_dprev;
            if (e < EM) sc[e] = on ? de_tot * ev_e * (1.f - ev_e) : 0.f;
            my_dprev = on ? c1 * dal_e : 0.f;            // gradient flowing into e_{t-1} through "prev": this lane's own entity
        }
        sim_lds_barrier();
        float dsv[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) dsv[e] = sc[e];
        if (!DMA) flush_held();            // (the other modes stage synchronously: nothing to hide the stores from)
        held_dc0 = dc0; held_dc1 = dc1; held_dw = dw; held_j = j;
        // dq and the last piece of dE
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int d = threadIdx.x + NT * u;
            if (d < D) {
                float dqv = 0.f;
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    if (e < E) {
                        dqv += dsv[e] * Es[(size_t)e * D + d];
                        dE[e][u] += dsv[e] * kq[u];
                    }
                }
                held_dq[u] = dqv;
            }
        }
        if (threadIdx.x < EM) {             // (the slots written here were last read at step t+1; the barrier at the top of step t-1 publishes them)
            const bool ok = (int)threadIdx.x < E;
            eoR[((t + 1) % 3) * EM + threadIdx.x] = (ok && t >= 2) ? nx_e : 0.f;          // (t-2) mod 3
            deR[((t + 1) & 1) * EM + threadIdx.x] = (ok && t >= 1) ? nx_de : 0.f;         // (t-1) & 1
        }
    }
    flush_held();
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int d = threadIdx.x + NT * u;
        if (d < D) {
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) a.dE0[((size_t)e0 + e) * D + d] = dE[e][u];
        }
    }
}

// ---- the backward for 17-32 entities per video at D = NT ∈ {256, 512, 768, 1024} (one column per thread, one state image in LDS:
// the reference's batches carry up to 31 ingredients).  Same arithmetic and summation order as sim_recur_bwd_kernel; what differs is
// where things live.  The generic kernel keeps nine 32-entry per-entity arrays in registers (α, e, e_prev, de, the two partial sums, dαβ,
// dα, ds — unrolled to EM): at 768 threads (168 VGPRs) that is 540 bytes of scratch per lane, touched inside every step of the dependent
// chain, and with one image its loads of the upstream gradient sat inside the entity loop, one memory round trip per entity (850 µs for
// 16 steps × 31 entities).  Here: (1) the per-entity scalars live in LDS (written by the first 32 threads from values requested one
// step ahead, read back as broadcasts), (2) the two per-entity partial sums are wave-reduced as they are produced, (3) the scalar
// algebra is done once by thread e instead of redundantly by all, (4) the state image comes by LDS-DMA and the thread's column of the
// upstream gradient — all entities — is requested at the top of the step: one memory round trip per step.  Registers per thread: the
// carried gradient column dE[32] and that upstream column.
template <int NT>
__global__ __launch_bounds__(NT) void sim_recur_bwd_lean_kernel(SimArgs a) {
    constexpr int NW = NT / 64, EM = SIM_EMAX;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int b = blockIdx.x, D = a.D, em = a.e_max;            // D == NT (host check)
    const int s0 = a.step_off[b], S = a.step_len[b], e0 = a.ent_off[b], E = a.ent_len[b];
    const int img = ((em * D + 255) / 256) * 256;
    float* Es = smem;
    float* red = Es + img;                   // (2·EM + 1) × NW wave partials
    float* dprev = red + (2 * EM + 1) * NW;  // EM: gradient flowing into e_{t-1} through "prev"
    float* tot = dprev + EM;                 // 2·EM + 1 block totals
    float* al_s = tot + (2 * EM + 1);        // EM each: α, e, e_prev, upstream de, dα, ds
    float* ev_s = al_s + EM;
    float* pv_s = ev_s + EM;
    float* dv_s = pv_s + EM;
    float* dal_s = dv_s + EM;
    float* ds_s = dal_s + EM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = tid;
    float dE[EM], upv[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) { dE[e] = 0.f; upv[e] = 0.f; }
    if (tid < EM) dprev[tid] = 0.f;
    auto state_before = [&](int t) -> const float* {
        return t == 0 ? a.E0 + (size_t)e0 * D : a.eall + (size_t)(s0 + t - 1) * em * D;
    };
    // operands of step t-1 that do not depend on the recurrence, requested during step t
    float n_c0 = 0.f, n_c1 = 0.f, n_w = 0.f, n_eb = 0.f, n_deb = 0.f, n_q = 0.f, n_ev = 0.f, n_pv = 0.f, n_dv = 0.f;
    const int ecl = min(tid, max(E - 1, 0));
    auto fetch = [&](int tt) {
        const int jj = s0 + tt;
        n_c0 = a.c[(size_t)jj * 3]; n_c1 = a.c[(size_t)jj * 3 + 1]; n_w = a.w4f[jj];
        n_eb = a.ebar[(size_t)jj * D + d];
        n_deb = a.debar ? a.debar[(size_t)jj * D + d] : 0.f;
        n_q = a.q[(size_t)jj * D + d];
        // (every thread loads — clamped — so that the loads are not predicated; threads ≥ E do not use them)
        n_ev = a.e_out[(size_t)jj * em + ecl];
        n_pv = a.e_out[(size_t)max(jj - 1, 0) * em + ecl];
        n_dv = a.de ? a.de[(size_t)jj * em + ecl] : 0.f;
    };
    if (S > 0) fetch(S - 1);
    for (int t = S - 1; t >= 0; --t) {
        const int j = s0 + t;
        __syncthreads();                                       // everyone is done with the image and the scalars of step t+1
        sim_dma<NT>(state_before(t), Es, E * D, wave, lane);
        if (a.deall) {
#pragma unroll
            for (int e = 0; e < EM; ++e)                        // (rows past the end re-read the last: unconditional, all in flight)
                upv[e] = a.deall[((size_t)j * em + max(min(e, E - 1), 0)) * D + d];
        }
        const float c0 = n_c0, c1 = n_c1, w = n_w, eb = n_eb, debv = n_deb, qv = n_q;
        if (tid < EM) {
            const bool on = tid < E;
            const float ev = on ? n_ev : 0.f, pv = (on && t > 0) ? n_pv : 0.f;
            ev_s[tid] = ev; pv_s[tid] = pv; dv_s[tid] = on ? n_dv : 0.f;
            al_s[tid] = on ? c0 * ev + c1 * pv : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t > 0) fetch(t - 1);
        const float k = fmaxf(w * eb, 0.f);
        float dk = 0.f, Z = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            if (e < E) {
                const float al = al_s[e];
                const float g = dE[e] + upv[e];
                const float Ev = Es[(size_t)e * D + d];
                const float ra = wave_sum(g * (k - Ev));
                if (lane == 0) red[e * NW + wave] = ra;
                dk += g * al;
                dE[e] = g * (1.f - al);
                Z += al;
            }
        }
        const float invZ = 1.0f / Z;
        const float dpre = k > 0.f ? dk : 0.f;
        const float deb = dpre * w + debv;
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            if (e < E) {
                const float rb = wave_sum(deb * Es[(size_t)e * D + d]);
                if (lane == 0) red[(EM + e) * NW + wave] = rb;
                dE[e] += al_s[e] * invZ * deb;
            }
        }
        const float pw = wave_sum(dpre * eb);
        if (lane == 0) red[(2 * EM) * NW + wave] = pw;
        __syncthreads();
        if (tid < 2 * EM + 1) {                                // wave order: deterministic
            const float* rp = red + tid * NW;
            float t_ = rp[0];
#pragma unroll
            for (int w_ = 1; w_ < NW; ++w_) t_ += rp[w_];
            tot[tid] = t_;
        }
        __syncthreads();
        if (tid < EM) {
            float mix = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) mix += tot[EM + e] * al_s[e] * invZ;
            float dal = 0.f, dsv = 0.f;
            if (tid < E) {
                dal = tot[tid] + (tot[EM + tid] - mix) * invZ;
                const float ev = ev_s[tid];
                const float de_tot = c0 * dal + dv_s[tid] + dprev[tid];
                dsv = de_tot * ev * (1.f - ev);
            }
            dal_s[tid] = dal; ds_s[tid] = dsv;
            dprev[tid] = tid < E ? c1 * dal : 0.f;
        }
        __syncthreads();
        if (tid == 0) {
            float dc0 = 0.f, dc1 = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) { dc0 += dal_s[e] * ev_s[e]; dc1 += dal_s[e] * pv_s[e]; }
            a.dc[(size_t)j * 3] = dc0; a.dc[(size_t)j * 3 + 1] = dc1; a.dc[(size_t)j * 3 + 2] = 0.f;
            a.dw4f[j] = tot[2 * EM];
        }
        float dqv = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            if (e < E) {
                const float dsv = ds_s[e];
                dqv += dsv * Es[(size_t)e * D + d];
                dE[e] += dsv * qv;
            }
        }
        a.dq[(size_t)j * D + d] = dqv;
    }
#pragma unroll
    for (int e = 0; e < EM; ++e)
        if (e < E) a.dE0[((size_t)e0 + e) * D + d] = dE[e];
}

static int sim_set_lds(const void* fn, size_t bytes) { return svpc_raise_lds_once(fn, "sim_recur"); }   // once per kernel symbol, process-wide table (api.cpp)


// ------------------------------------------------------------------------------------------------ the simulator's two tiny heads
// c = softmax(ĥ·W3ᵀ + b3) (three choice weights per step, model.py:801) and w = f̄·W4ᵀ + b4 (one scalar per step, :804-805) — projections
// onto 3 and 1 outputs.  As GEMM launches they fell on the generic kernel (N not a multiple of 4): 10-16 µs each forward, and backward a
// dgrad + wgrad + column sum + finalize apiece plus the softmax backward: 9 launches (≈ 75 µs) per simulator for ≈ 1 MFLOP.  Here: one
// launch forward, one backward (a wave per step row; the weight / bias gradients leave as per-workgroup partial sums for the table-driven
// finalizer).  fp32 arithmetic.
struct SimHeadArgs {
    const float* hh; const float* fb; const float* W3; const float* b3; const float* W4; const float* b4;
    float* c; float* w;            // (T, 3), (T)
    int T, D, Wd;
    // backward
    const float* dc; const float* dw; float* dhh; float* dfb; float* part3; float* part4; int rows_per_wg;
};
constexpr int SH_NPL = 16;          // D <= 1024
constexpr int SH_WPL = 8;           // word-vector width <= 512
constexpr int SH_ROWS = 8;          // step rows per backward workgroup (two per wave): T = 192 → 24 workgroups; at 32 rows the six workgroups took 54 µs

__global__ __launch_bounds__(256) void sim_heads_fwd_kernel(SimHeadArgs a) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.T) return;
    const float* hr = a.hh + (size_t)r * a.D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int d = lane; d < a.D; d += 64) {
        const float h = hr[d];
        s0 += h * a.W3[d]; s1 += h * a.W3[a.D + d]; s2 += h * a.W3[2 * a.D + d];
    }
    const float* fr = a.fb + (size_t)r * a.Wd;
    for (int d = lane; d < a.Wd; d += 64) s3 += fr[d] * a.W4[d];
    s0 = wave_sum(s0) + a.b3[0]; s1 = wave_sum(s1) + a.b3[1]; s2 = wave_sum(s2) + a.b3[2]; s3 = wave_sum(s3) + a.b4[0];
    if (lane == 0) {
        const float m = fmaxf(s0, fmaxf(s1, s2));
        const float e0 = expf(s0 - m), e1 = expf(s1 - m), e2 = expf(s2 - m), inv = 1.0f / (e0 + e1 + e2);
        a.c[(size_t)r * 3] = e0 * inv; a.c[(size_t)r * 3 + 1] = e1 * inv; a.c[(size_t)r * 3 + 2] = e2 * inv;
        a.w[r] = s3;
    }
}

// grid: ceil(T / rows_per_wg) workgroups of 256 threads; wave v takes rows r0 + v, r0 + v + 4, …;  partial sums of one workgroup:
// part3[g][0 : 3D] = dW3, part3[g][3D : 3D + 3] = db3;  part4[g][0 : Wd] = dW4, part4[g][Wd] = db4
__global__ __launch_bounds__(256) void sim_heads_bwd_kernel(SimHeadArgs a) {
    __shared__ float red[4][3 * 64 * SH_NPL / 4];       // cross-wave sums, reused per chunk (3 × 64·NPL/4 floats per wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.x;
    const int D = a.D, Wd = a.Wd;
    const int r0 = g * a.rows_per_wg, r1 = min(a.T, r0 + a.rows_per_wg);
    float aw3[3][SH_NPL], aw4[SH_WPL], ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < SH_NPL; ++i) { aw3[0][i] = aw3[1][i] = aw3[2][i] = 0.f; }
#pragma unroll
    for (int i = 0; i < SH_WPL; ++i) aw4[i] = 0.f;
    float w3v[3][SH_NPL], w4v[SH_WPL];
#pragma unroll
    for (int i = 0; i < SH_NPL; ++i) {
        const int d = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 3; ++j) w3v[j][i] = d < D ? a.W3[(size_t)j * D + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SH_WPL; ++i) { const int d = lane + 64 * i; w4v[i] = d < Wd ? a.W4[d] : 0.f; }
    for (int r = r0 + wave; r < r1; r += 4) {
        const float c0 = a.c[(size_t)r * 3], c1 = a.c[(size_t)r * 3 + 1], c2 = a.c[(size_t)r * 3 + 2];
        const float g0 = a.dc ? a.dc[(size_t)r * 3] : 0.f, g1 = a.dc ? a.dc[(size_t)r * 3 + 1] : 0.f, g2 = a.dc ? a.dc[(size_t)r * 3 + 2] : 0.f;
        const float dot = c0 * g0 + c1 * g1 + c2 * g2;
        const float d0 = c0 * (g0 - dot), d1 = c1 * (g1 - dot), d2 = c2 * (g2 - dot);      // softmax backward → gradient of the three logits
        const float dwr = a.dw ? a.dw[r] : 0.f;
        ab[0] += d0; ab[1] += d1; ab[2] += d2; ab[3] += dwr;
        const float* hr = a.hh + (size_t)r * D;
        float* dh = a.dhh + (size_t)r * D;
#pragma unroll
        for (int i = 0; i < SH_NPL; ++i) {
            const int d = lane + 64 * i;
            if (d < D) {
                const float h = hr[d];
                aw3[0][i] += d0 * h; aw3[1][i] += d1 * h; aw3[2][i] += d2 * h;
                dh[d] = d0 * w3v[0][i] + d1 * w3v[1][i] + d2 * w3v[2][i];
            }
        }
        const float* fr = a.fb + (size_t)r * Wd;
        float* df = a.dfb + (size_t)r * Wd;
#pragma unroll
        for (int i = 0; i < SH_WPL; ++i) {
            const int d = lane + 64 * i;
            if (d < Wd) { aw4[i] += dwr * fr[d]; df[d] = dwr * w4v[i]; }
        }
    }
    // the four waves' sums → one partial row per workgroup (wave order: deterministic), in chunks of four column groups through LDS
    float* p3 = a.part3 + (size_t)g * (3 * D + 3);
    float* p4 = a.part4 + (size_t)g * (Wd + 1);
    for (int i0 = 0; i0 < SH_NPL; i0 += SH_NPL / 4) {
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < SH_NPL / 4; ++ii)
#pragma unroll
            for (int j = 0; j < 3; ++j) red[wave][(j * (SH_NPL / 4) + ii) * 64 + lane] = aw3[j][i0 + ii];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int ii = 0; ii < SH_NPL / 4; ++ii) {
                const int d = lane + 64 * (i0 + ii);
                if (d < D) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int o = (j * (SH_NPL / 4) + ii) * 64 + lane;
                        p3[(size_t)j * D + d] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SH_WPL; ++i) red[wave][i * 64 + lane] = aw4[i];
    {   // (ab is wave-uniform: every lane accumulated the same rows)
        const float abv = lane == 0 ? ab[0] : (lane == 1 ? ab[1] : (lane == 2 ? ab[2] : ab[3]));
        if (lane < 4) red[wave][SH_WPL * 64 + lane] = abv;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < SH_WPL; ++i) {
            const int d = lane + 64 * i;
            if (d < Wd) { const int o = i * 64 + lane; p4[d] = red[0][o] + red[1][o] + red[2][o] + red[3][o]; }
        }
        if (lane < 4) {
            const int o = SH_WPL * 64 + lane;
            const float v = red[0][o] + red[1][o] + red[2][o] + red[3][o];
            if (lane < 3) p3[3 * (size_t)D + lane] = v; else p4[Wd] = v;
        }
    }
}

extern "C" {

// the simulator's choice softmax (D → 3) and verb scalar (W → 1) in one launch (model.py:801, :804-805); backward in one launch too: dhh / dfb
// (T, D) / (T, Wd) fully written, part3 (groups, 3·D + 3) = per-workgroup [dW3 | db3], part4 (groups, Wd + 1) = [dW4 | db4] for the
// finalizer; groups = svpc_sim_heads_groups(T)
int svpc_sim_heads_groups(int T) { return T <= 0 ? 0 : ceil_div(T, SH_ROWS); }
int svpc_sim_heads_fwd(const float* hh, const float* fb, const float* W3, const float* b3, const float* W4, const float* b4, float* c, float* w,
                       int T, int D, int Wd, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(D >= 1 && Wd >= 1, "sim_heads: empty rows");
    SimHeadArgs a{};
    a.hh = hh; a.fb = fb; a.W3 = W3; a.b3 = b3; a.W4 = W4; a.b4 = b4; a.c = c; a.w = w; a.T = T; a.D = D; a.Wd = Wd;
    hipLaunchKernelGGL(sim_heads_fwd_kernel, dim3(ceil_div(T, 4)), dim3(256), 0, stream, a);
    return svpc_check_launch("sim_heads_fwd");
}
int svpc_sim_heads_bwd(const float* hh, const float* fb, const float* W3, const float* W4, const float* c, const float* dc, const float* dw,
                       float* dhh, float* dfb, float* part3, float* part4, int T, int D, int Wd, hipStream_t stream) {
    if (T == 0) return 0;
    SVPC_REQUIRE(D <= 64 * SH_NPL && Wd <= 64 * SH_WPL, "sim_heads: hidden size <= 1024, word-vector width <= 512");
    SimHeadArgs a{};
    a.hh = hh; a.fb = fb; a.W3 = W3; a.W4 = W4; a.c = const_cast<float*>(c); a.dc = dc; a.dw = dw; a.dhh = dhh; a.dfb = dfb; a.part3 = part3;
    a.part4 = part4; a.T = T; a.D = D; a.Wd = Wd; a.rows_per_wg = SH_ROWS;
    hipLaunchKernelGGL(sim_heads_bwd_kernel, dim3(svpc_sim_heads_groups(T)), dim3(256), 0, stream, a);
    return svpc_check_launch("sim_heads_bwd");
}

int svpc_sim_recur_fwd(const float* q, const float* c, const float* w4f, const float* E0, const int* step_off, const int* step_len,
                       const int* ent_off, const int* ent_len, int n_videos, int e_max, int D, float* e_out, float* ebar,
                       float* eall, hipStream_t stream) {
    if (n_videos == 0) return 0;
    SVPC_REQUIRE(e_max >= 1 && e_max <= SIM_EMAX, "sim_recur: at most 32 entities per video");
    const size_t lds = ((size_t)e_max * D + 3 * SIM_EMAX) * sizeof(float);
    SVPC_REQUIRE(lds <= 150 * 1024, "sim_recur: entity state does not fit LDS");
    SimArgs a{};
    a.q = q; a.c = c; a.w4f = w4f; a.E0 = E0; a.step_off = step_off; a.step_len = step_len; a.ent_off = ent_off;
    a.ent_len = ent_len; a.e_out = e_out; a.ebar = ebar; a.eall = eall; a.e_max = e_max; a.D = D;
    int rc = sim_set_lds((const void*)sim_recur_fwd_kernel, lds);
    if (rc) return rc;
    const int nt_f = D > 512 ? 768 : (D > 256 ? 512 : 256);
    hipLaunchKernelGGL(sim_recur_fwd_kernel, dim3(n_videos), dim3(nt_f), lds, stream, a);
    return svpc_check_launch("sim_recur_fwd");
}

int svpc_sim_recur_bwd(const float* q, const float* c, const float* w4f, const float* E0, const int* step_off, const int* step_len,
                       const int* ent_off, const int* ent_len, int n_videos, int e_max, int D, const float* e_out,
                       const float* ebar, const float* eall, const float* de, const float* debar, const float* deall, float* dq,
                       float* dc, float* dw4f, float* dE0, hipStream_t stream) {
    if (n_videos == 0) return 0;
    SVPC_REQUIRE(e_max >= 1 && e_max <= SIM_EMAX, "sim_recur: at most 32 entities per video");
    SVPC_REQUIRE(D <= 1024, "sim_recur: hidden size must be <= 1024");
    const int nt = D > 768 ? 1024 : (D > 512 ? 768 : (D > 256 ? 512 : 256));      // one column per thread up to D = 1024
    const size_t img = (((size_t)e_max * D + 255) / 256) * 256;
    const int EMv = e_max <= 16 ? 16 : 32;
    const size_t tail = ((2 * EMv + 1) * (nt / 64) + 2 * EMv + (2 * EMv + 1) + 5 * EMv) * sizeof(float);
    const size_t budget = 150 * 1024;
    const bool aligned = ((((uintptr_t)E0) | ((uintptr_t)eall) | ((uintptr_t)deall)) & 15) == 0;
    int mode = 0;
    if (D % 256 == 0 && aligned && 4 * img * sizeof(float) + tail <= budget) mode = 2;
    else if (2 * img * sizeof(float) + tail <= budget) mode = 1;
    else if (D % 256 == 0 && aligned) mode = 3;
    const size_t lds = (mode == 2 ? 4 : (mode == 1 ? 2 : 1)) * img * sizeof(float) + tail;
    SVPC_REQUIRE(lds <= budget, "sim_recur: entity state does not fit LDS");
    SimArgs a{};
    a.q = q; a.c = c; a.w4f = w4f; a.E0 = E0; a.step_off = step_off; a.step_len = step_len; a.ent_off = ent_off;
    a.ent_len = ent_len; a.e_out = const_cast<float*>(e_out); a.ebar = const_cast<float*>(ebar);
    a.eall = const_cast<float*>(eall); a.e_max = e_max; a.D = D; a.de = de; a.debar = debar; a.deall = deall;
    a.dq = dq; a.dc = dc; a.dw4f = dw4f; a.dE0 = dE0;
    int rc;
    static int lean_env = -1;
    if (lean_env < 0) { const char* e = getenv("SVPC_SIM_LEAN"); lean_env = e ? atoi(e) : 1; }
    if (((mode == 3 && EMv == 32 && lean_env) || (lean_env == 2 && D % 256 == 0 && aligned)) && nt == D) {      // (2: experiments — every entity count)
        const size_t lds_lean = img * sizeof(float) + ((2 * SIM_EMAX + 1) * (nt / 64) + SIM_EMAX + (2 * SIM_EMAX + 1) + 6 * SIM_EMAX) * sizeof(float);
        SVPC_REQUIRE(lds_lean <= budget, "sim_recur: entity state does not fit LDS");
#define SIM_LEAN_GO(NTV)                                                                                             \
    do {                                                                                                             \
        rc = sim_set_lds((const void*)sim_recur_bwd_lean_kernel<NTV>, lds_lean); if (rc) return rc;                   \
        hipLaunchKernelGGL((sim_recur_bwd_lean_kernel<NTV>), dim3(n_videos), dim3(NTV), lds_lean, stream, a);         \
    } while (0)
        if (nt == 256) SIM_LEAN_GO(256);
        else if (nt == 512) SIM_LEAN_GO(512);
        else if (nt == 768) SIM_LEAN_GO(768);
        else SIM_LEAN_GO(1024);
#undef SIM_LEAN_GO
        return svpc_check_launch("sim_recur_bwd");
    }
#define SIM_BWD_GO2(NTV, MV, EMV)                                                                                  \
    do {                                                                                                           \
        rc = sim_set_lds((const void*)sim_recur_bwd_kernel<1, NTV, MV, EMV>, lds); if (rc) return rc;               \
        hipLaunchKernelGGL((sim_recur_bwd_kernel<1, NTV, MV, EMV>), dim3(n_videos), dim3(NTV), lds, stream, a);     \
    } while (0)
#define SIM_BWD_GO(NTV, MV) do { if (EMv == 16) SIM_BWD_GO2(NTV, MV, 16); else SIM_BWD_GO2(NTV, MV, 32); } while (0)
#define SIM_BWD_NT(NTV) do { if (mode == 2) SIM_BWD_GO(NTV, 2); else if (mode == 1) SIM_BWD_GO(NTV, 1); else if (mode == 3) SIM_BWD_GO(NTV, 3); else SIM_BWD_GO(NTV, 0); } while (0)
    if (nt == 256) SIM_BWD_NT(256);
    else if (nt == 512) SIM_BWD_NT(512);
    else if (nt == 768) SIM_BWD_NT(768);
    else SIM_BWD_NT(1024);
#undef SIM_BWD_NT
#undef SIM_BWD_GO
#undef SIM_BWD_GO2
    return svpc_check_launch("sim_recur_bwd");
}

}  // extern "C"
