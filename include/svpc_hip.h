/* svpc_hip.h — C-ABI of the MI355X (gfx950) kernel library behind svpc_amd's StateAwareRecursiveTransformer.
 *
 * The reference (awkrail/svpc) has no FFI / plugin layer: its hot path is eager PyTorch inside
 * src/rtransformer/model.py.  The drop-in boundary is therefore the Python module API; this
 * header is what sits directly beneath it.  Each entry point replaces the group of eager ops cited next to it
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: every pointer is a device (HBM) pointer unless noted; tensors are row-major fp32 / int32;
 * `stream` is the HIP stream to launch on (svpc_amd passes PyTorch's current stream); nothing is allocated
 * and nothing synchronises inside (graph-capture safe); return 0 = ok, non-zero = error with a message in
 * svpc_last_error().  Dropout / Gumbel draws are counter-based: (`seed` device word, `site`, element index).
 */
#ifndef SVPC_HIP_H
#define SVPC_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* svpc_stream_t; /* == hipStream_t */
typedef unsigned long long svpc_u64;

#define SVPC_ACT_NONE 0
#define SVPC_ACT_RELU 1
#define SVPC_ACT_GELU 2    /* exact erf GELU, model.py:58-64 */
#define SVPC_ACT_SIGMOID 3

const char* svpc_last_error(void);
int svpc_abi_version(void);

/* ---- LayerNorm family: BertLayerNorm model.py:143-156 and its fused uses :229-233, :285-289, :493-499, :548-562,
 *      :650, :659, :889-891.  y = post_drop(LN(pre_drop(x[src_rows[r]]) + res[r])·gamma+beta) + add1[r%mod1] + add2[idx2[r]] */
int svpc_ln_fwd(const float* x, const int* src_rows, const float* res, const float* gamma, const float* beta, float* y,
                float* mean, float* rstd, int R, int D, float eps, float p_pre, unsigned site_pre, float p_post,
                unsigned site_post, const svpc_u64* seed, const float* add1, int mod1, const float* add2, const int* idx2,
                svpc_stream_t stream);
/* typed forms (dtype codes 0 = fp32, 1 = bf16; x_dt: x/dx, y_dt: residual/y/dy/dh; (0,0), (0,1), (1,1)); statistics stay fp32 */
int svpc_ln_fwd_t(const void* x, int x_dt, const int* src_rows, const void* res, const float* gamma, const float* beta, void* y,
                  int y_dt, float* mean, float* rstd, int R, int D, float eps, float p_pre, unsigned site_pre, float p_post,
                  unsigned site_post, const svpc_u64* seed, const float* add1, int mod1, const float* add2, const int* idx2,
                  svpc_stream_t stream);
int svpc_ln_bwd_t(const void* dy, const void* x, int x_dt, int y_dt, const int* src_rows, const void* res, const float* gamma,
                  const float* mean, const float* rstd, void* dh, void* dx, float* dgamma, float* dbeta, int accumulate,
                  float* workspace, int R, int D, float p_pre, unsigned site_pre, float p_post, unsigned site_post,
                  const svpc_u64* seed, svpc_stream_t stream);
/* the same in two calls, so that the parameter-gradient tail can run on another stream: partial = svpc_ln_bwd_groups(R)·2D floats */
int svpc_ln_bwd_rows_t(const void* dy, const void* x, int x_dt, int y_dt, const int* src_rows, const void* res, const float* gamma,
                       const float* mean, const float* rstd, void* dh, void* dx, float* partial, int R, int D, float p_pre,
                       unsigned site_pre, float p_post, unsigned site_post, const svpc_u64* seed, svpc_stream_t stream);
int svpc_ln_param_grads(const float* partial, int R, int D, float* dgamma, float* dbeta, int accumulate, svpc_stream_t stream);
/* strided / split forms (bf16x3 mode).  dtype code 2 = "split": a row stores an fp32-like value as TWO bf16 planes, hi = bf16(v) at
 * column c and lo = bf16(v - hi) at column lo + c (hi + lo is exact in fp32, 16-17 significant bits; the bytes of fp32).  ldx / ldr /
 * ldy: row strides in elements (0 = dense); lox / lor / loy: column offsets of the lo planes.  Forward combinations (x_dt, y_dt):
 * the typed ones plus (0,2), (2,2), (2,0).  The backward reads the hi planes of split rows in place as dtype 1 with their leading
 * dimension (ldx, ldr); dy / dh / dx stay dense.  Same reference lines as svpc_ln_fwd_t. */
int svpc_ln_fwd_s(const void* x, int x_dt, int ldx, int lox, const int* src_rows, const void* res, int ldr, int lor, const float* gamma,
                  const float* beta, void* y, int y_dt, int ldy, int loy, float* mean, float* rstd, int R, int D, float eps, float p_pre,
                  unsigned site_pre, float p_post, unsigned site_post, const svpc_u64* seed, const float* add1, int mod1,
                  const float* add2, const int* idx2, svpc_stream_t stream);
int svpc_ln_bwd_rows_s(const void* dy, const void* x, int x_dt, int ldx, int y_dt, const int* src_rows, const void* res, int ldr,
                       const float* gamma, const float* mean, const float* rstd, void* dh, void* dx, float* partial, int R, int D,
                       float p_pre, unsigned site_pre, float p_post, unsigned site_post, const svpc_u64* seed, svpc_stream_t stream);
int svpc_bucket_colsum_t(const void* x, int x_dt, int ldx, const int* idx, int R, int C, int K, float* out, int accumulate,
                         float* workspace, svpc_stream_t stream);
/* deferred reduction tails: the first stage of a plain column sum into a caller-owned partial buffer, and ONE launch that adds the
 * column sums of up to svpc_multi_finalize_max() partial buffers (bias gradients; LayerNorm [dgamma ; dbeta] with split = D) into
 * their arena targets: out(c) += Σ_g partial[g][c], c < split → out0[c], else out1[c - split].  `entries` is a HOST array. */
typedef struct svpc_finalize_entry { const float* partial; float* out0; float* out1; int groups, ncols, split; } svpc_finalize_entry;
int svpc_colsum_partial_t(const void* x, int x_dt, int ldx, int R, int C, float* partial, svpc_stream_t stream);
/* first stages of up to 48 plain column sums (dt: 0 = fp32, 1 = bf16 input) in one launch; partial_i = svpc_colsum_chunks(R_i) × C_i */
typedef struct svpc_colsum_entry { const void* x; float* partial; int dt, ldx, R, C; } svpc_colsum_entry;
int svpc_multi_colsum(const svpc_colsum_entry* entries, int n, svpc_stream_t stream);
int svpc_multi_finalize_max(void);
int svpc_multi_finalize(const svpc_finalize_entry* entries, int n, svpc_stream_t stream);
int svpc_ln_bwd_groups(int R); /* workspace floats needed by svpc_ln_bwd = (groups + 1) * 2 * D */
/* rows of `partial` that svpc_ln_bwd_rows_s writes when dh == dx == NULL (parameter gradients only: the first LayerNorm over the frame
 * features, model.py:548) — fewer, longer workgroups than the rows pass; svpc_ln_param_grads_g sums a given number of partial rows */
int svpc_ln_param_only_groups(int R);
int svpc_ln_param_grads_g(const float* partial, int groups, int D, float* dgamma, float* dbeta, int accumulate, svpc_stream_t stream);
int svpc_ln_bwd(const float* dy, const float* x, const int* src_rows, const float* res, const float* gamma,
                const float* mean, const float* rstd, float* dh, float* dx, float* dgamma, float* dbeta, int accumulate,
                float* workspace, int R, int D, float p_pre, unsigned site_pre, float p_post, unsigned site_post,
                const svpc_u64* seed, svpc_stream_t stream);
/* out[k][c] (+)= Σ_{r: idx[r]==k} x[r][c]  — bias gradients (K=1, idx NULL) and the token-type table gradient (model.py:890) */
int svpc_colsum_chunks(int R); /* workspace floats = chunks * K * C */
int svpc_bucket_colsum(const float* x, int ldx, const int* idx, int R, int C, int K, float* out, int accumulate,
                       float* workspace, svpc_stream_t stream);

/* ---- fp32 MFMA GEMM with fused epilogue: every nn.Linear on the path (model.py:195-197, :230, :259, :281, :496, :551,
 *      :706, :738, :755-770, :846-850, :859-860, nn.LSTM projections :865) and their dgrad / wgrad products.
 *      C[M,N] = epi(Σ_k A(m,k)·B(n,k)); a_kc/b_kc: 1 = k contiguous ([rows][ld]), 0 = k strided ([K][ld]). */
int svpc_gemm_f32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N,
                  int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                  float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* same contract on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): fp32 operands are rounded to bf16 while staged, fp32 accumulate */
int svpc_gemm_bf16(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N,
                   int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                   float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* typed form (dtype codes 0 = fp32, 1 = bf16): (f32,f32→f32) any shape; (bf16,f32→bf16) NT/NN and (bf16,bf16→f32) TN for
 * interior-only shapes — the bf16 activation stream of the clip encoder */
int svpc_gemm_mx(const void* A, int a_dt, int lda, int a_kc, const void* B, int b_dt, int ldb, int b_kc, void* C, int c_dt, int ldc,
                 void* Z, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed,
                 int accumulate, float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* bf16 × bf16 form with direct-to-LDS operand staging (global_load_lds into an LDS ring, counted vmcnt): any M, N (see
 * svpc_gemm_glds_supported); weights come from the optimizer's bf16 shadow arena.  C bf16 (c_dt 1) or fp32 (c_dt 0).  Launches of
 * at least 200 tiles of 256x256 with a k-contiguous operand run on the two-group ping-pong kernel; bf16 outputs are written as whole
 * 128-byte lines. */
int svpc_gemm_glds_supported(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K);
int svpc_gemm_glds(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, int M, int N,
                   int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                   float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* the same with an addend R of C's type and leading dimension: C = epi(A·B) + R.  Lets the dgrad of a sub-layer's first projection
 * absorb the residual-path gradient (reference: the `+` of model.py:229-233 / :285-289 in backward) instead of a separate add kernel */
int svpc_gemm_glds_r(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, const void* R,
                     int M, int N, int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                     float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* … and with an activation-backward factor G (of C's type and leading dimension): C = epi(A·B) ⊙ gact'(G) + R, where G is what the
 * forward of activation `gact` kept (the pre-activation z for GELU, the output y for ReLU / sigmoid).  The dgrad of the projection
 * that FOLLOWS an activation (reference: BertOutput.dense after BertIntermediate's gelu, model.py:255-289) writes the gradient of the
 * pre-activation directly; the separate svpc_act_bwd pass over the stream disappears.  G needs a non-accumulating bf16 output. */
int svpc_gemm_glds_rg(const void* A, int lda, int a_kc, const void* B, int ldb, int b_kc, void* C, int c_dt, int ldc, void* Z, const void* R,
                      const void* G, int gact, int M, int N, int K, const float* bias, int act, float p_drop, unsigned site,
                      const svpc_u64* seed, int accumulate, float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* the 8-phase 256x256x64 form of the same product for the FORWARD projections of a bf16 activation stream (both operands
 * k-contiguous bf16, bf16 output, optional bias / ReLU / GELU / pre-activation copy Z; K % 64 == 0, N % 8 == 0, 16-byte aligned rows):
 * C = act(A[M,K]·B[N,K]^T + bias).  svpc_gemm_glds routes its stream-sized forward launches here; the direct entry exists for tests
 * and tools.  reference: every nn.Linear of the clip encoder / decoder forward (model.py:195-197,230,259,281,551). */
int svpc_gemm_p8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* Z, int M, int N, int K, const float* bias, int act,
                 svpc_stream_t stream);
/* bf16x3 form of the same projections (the ≤1e-4-parity throughput mode): split operands (see svpc_ln_fwd_s), three-term product
 *   A·Bᵀ ≈ A_lo·B_hiᵀ + A_hi·B_loᵀ + A_hi·B_hiᵀ   — one bf16 GEMM with a 3K-deep contraction on the svpc_gemm_p8 structure, fp32
 * accumulate, ≈2⁻¹⁷ per operand instead of 2⁻⁹ at 3× (not 16×, as the f32 MFMA) the bf16 matrix work.  A [M][lda]: hi plane at column
 * 0, lo plane at column a_lo; B_hi [N][ldb] with B_lo = B_hi + b_lo elements (same layout: the two planes of the weight shadow);
 * C [M][ldc] written split (hi at column c, lo at c_lo + c); Z (optional, only with an activation): plain bf16 [M][ldz] pre-activation
 * copy for the bf16 backward.  K % 64 == 0, N % 8 == 0, every plane 16-byte aligned; act in {none, relu, gelu}. */
int svpc_gemm_p8x3_supported(int lda, int a_lo, int ldb, int ldc, int c_lo, int ldz, int M, int N, int K);
int svpc_gemm_p8x3(const void* A, int lda, int a_lo, const void* B, int ldb, long long b_lo, void* C, int ldc, int c_lo, void* Z, int ldz,
                   int M, int N, int K, const float* bias, int act, svpc_stream_t stream);
/* the same contract on 128x128 tiles (4 waves, two workgroups per CU): the decoder's projections (4,224 sentence rows, 576 memory
 * rows; model.py:620-663), where 256x256 tiles leave most of the 256 CUs idle */
int svpc_gemm_s4x3(const void* A, int lda, int a_lo, const void* B, int ldb, long long b_lo, void* C, int ldc, int c_lo, void* Z, int ldz,
                   int M, int N, int K, const float* bias, int act, svpc_stream_t stream);
/* bf16 GEMM with a k-STRIDED B operand on the same 8-phase template — the input gradients (dgrad) of the stream projections:
 *   C[M,N] = (A[M,K] · B[K,N]) ⊙ gact'(G[M,N]) + R[M,N],  A = dz (k-contiguous), B = the weight matrix as stored (W[out = K][in = N]),
 * G (optional) what the forward of the activation in front of this projection's input kept, R (optional) a parked residual-path gradient;
 * all bf16, C / G / R share ldc.  K % 64 == 0, N % 8 == 0.  B fragments come from a [64 k-rows][128 columns] LDS image through
 * ds_read_b64_tr_b16.  reference: the backward of every nn.Linear of the clip encoder (model.py:195-197,230,259,281,551). */
/* the same dgrad on 128x128 tiles, 8 waves, up to four 32-KiB stages: the decoder's projections with M = 4,224 or 576 rows, where a
   256x256 tiling leaves most CUs idle. gemm_s4t.hip; replaces the backward of the nn.Linear layers of BertDecoderLayerNoMemoryUntied,
   src/rtransformer/model.py:620-663.  C = A·B + R with R optional; no activation factor. */
int svpc_gemm_s4t_supported(int lda, int ldb, int ldc, int M, int N, int K);
int svpc_gemm_s4t(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* R, int M, int N, int K, svpc_stream_t stream);
int svpc_gemm_p8t_supported(int lda, int ldb, int ldc, int M, int N, int K);
int svpc_gemm_p8t(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* G, int gact, const void* R, int M, int N, int K,
                  svpc_stream_t stream);
/* fp32-operand form with direct-to-LDS staging (deep LDS ring, operands rounded to bf16 when the MFMA fragments are built): the
 * latency-bound GEMMs of the decoder :620-694, step-wise encoder :594-617, simulators :742-823, BiLSTM :1017-1025, LM head
 * :697-739 and their dgrad/wgrad.  Any M, N (edges clamped); K % 32 == 0; k-strided operands need rows % 4 == 0. */
int svpc_gemm_l32_supported(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K);
int svpc_gemm_l32_preferred(int a_kc, int b_kc, int lda, int ldb, int M, int N, int K);   /* supported AND measured faster */
int svpc_gemm_l32(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, int M, int N, int K,
                  const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate, float* workspace,
                  size_t workspace_bytes, svpc_stream_t stream);
/* the same with an fp32 addend R of C's leading dimension, C = epi(A·B) + R (fp32 twin of svpc_gemm_glds_r: the residual-path
 * gradient of a LayerNorm joins the dgrad of the projection that consumes the residual tensor — step-wise encoder, model.py:565-591) */
int svpc_gemm_l32_r(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R, int M,
                    int N, int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                    float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* ... and with the activation-backward factor, C = (A·B) * gact'(G) + R: G in fp32 and C's layout is what the forward of activation gact
 * kept, z for GELU, y for ReLU or sigmoid. The dgrad of the projection that consumes an activated tensor writes the gradient of the
 * pre-activation directly: the FFN of the step-wise encoder, model.py:565-591. No bias, activation or dropout in this form. */
int svpc_gemm_l32_rg(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, const float* R, const float* G,
                     int gact, int M, int N, int K, int accumulate, float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* the same contract with bf16x3 products: every fp32 operand value enters as hi + lo bf16 terms when the MFMA fragments are built,
 * three MFMAs per product — the forward arithmetic of the ≤1e-4-parity throughput mode for every projection in fp32 storage */
int svpc_gemm_l32_x3(const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, float* C, int ldc, float* Z, const float* R, int M,
                     int N, int K, const float* bias, int act, float p_drop, unsigned site, const svpc_u64* seed, int accumulate,
                     float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* grouped weight (+ bias) gradients of up to svpc_gemm_group_wgrad_max() independent linears in one launch:
 *   dw[n_out, n_in] += dzᵀ · x,   db[n_out] += Σ_rows dz   (db may be NULL)       — the wgrad half of every nn.Linear backward
 * `problems` is a HOST array of svpc_wgrad_problem; any row count (the partial last k-tile is zero-sourced), n_out % 4 == 0, n_in % 4 == 0,
 * 16-byte aligned operands */
/* grouped small GEMMs of one layout in one launch (fp32 operands, bf16 MFMA): C_p = (accumulate ? C_p : 0) + Σ_k A_p(m,k)·B_p(n,k);
 * `problems` is a HOST array, at most 32 entries, K % 32 == 0 — e.g. the recurrent projections of both LSTM directions at one step */
typedef struct svpc_gemm_problem { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc; } svpc_gemm_problem;
int svpc_gemm_group(const svpc_gemm_problem* problems, int n, int a_kc, int b_kc, int accumulate, svpc_stream_t stream);
int svpc_gemm_group_x3(const svpc_gemm_problem* problems, int n, int a_kc, int b_kc, int accumulate, svpc_stream_t stream); /* bf16x3 products */
typedef struct svpc_wgrad_problem {
    const float* dz; const float* x; float* dw; float* db;
    int n_out, n_in, rows, ld_dz, ld_x, ld_dw;
} svpc_wgrad_problem;
int svpc_gemm_group_wgrad_max(void);
/* the bf16-stream table (svpc_gemm_group_wgrad_bf16_ws below) on the 8-phase 256x256x64 template with BOTH operands read k-strided through
 * ds_read_b64_tr_b16 (gemm_p8w.hip): problems at least 256 wide both ways; same balance (deep tiles of later rounds cut into k-parts →
 * fp32 slabs in `workspace`, added in part order by a fix-up launch).  _ok: 1 if every problem of the table qualifies. */
int svpc_gemm_group_wgrad_bf16_p8_ok(const void* problems, int n);
int svpc_gemm_group_wgrad_bf16_p8(const void* problems, int n, float* workspace, size_t workspace_bytes, svpc_stream_t stream);
/* the same for the bf16 activation streams (dz, x bf16; dw fp32; db must be NULL; n_out % 8 == 0, n_in % 8 == 0, any row count):
 * every 128² tile runs its whole k-loop — no split-K slabs, no reduce launches */
int svpc_gemm_group_wgrad_bf16(const svpc_wgrad_problem* problems, int n, svpc_stream_t stream);
/* the same with scratch for load balancing: deep tiles dealt after the first round of workgroups are cut into k-parts that write
 * fp32 slabs into `workspace`; a second launch adds a tile's slabs in part order into dW (deterministic) */
int svpc_gemm_group_wgrad_bf16_ws(const svpc_wgrad_problem* problems, int n, float* workspace, size_t workspace_bytes,
                                  svpc_stream_t stream);
int svpc_gemm_group_wgrad(const svpc_wgrad_problem* problems, int n, svpc_stream_t stream);
/* dz = dy · act'(aux) · dropout  (aux = pre-activation for GELU, activated output for ReLU / sigmoid) */
int svpc_act_bwd(const float* dy, const float* aux, float* dz, size_t n, int act, float p, unsigned site, const svpc_u64* seed,
                 svpc_stream_t stream);

int svpc_act_bwd_t(const void* dy, const void* aux, void* dz, int dt, size_t n, int act, float p, unsigned site, const svpc_u64* seed,
                   svpc_stream_t stream);

/* ---- attention core: BertSelfAttention.forward model.py:194-219 (scale, additive -10000 mask, softmax, dropout, PV).
 *      seq = int32[4][n_seq]: q_off, q_len, k_off, k_len.  LSE: (n_seq, H, max_q). */
int svpc_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                  const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                  float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* single-query form for incremental greedy decoding (src/translator.py:88-100: position i attends to positions ≤ i): every sequence
 * has exactly one query row; fp32, forward only */
int svpc_attn_q1_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                     const int* seq, int n_seq, int H, int dh, int max_k, const float* key_mask, float scale, svpc_stream_t stream);
/* one decoding step of an attention block of the decoder layer (src/rtransformer/model.py:620-663 evaluated for ONE new position per
 * sentence, as src/translator.py:88-112 needs): O[t] = LayerNorm(X[t] + Attention(Q[t]; rows t·k_stride … t·k_stride+n_keys−1 of K / V)),
 * heads of dh = 64, fp32.  newK / newV (both or neither): the token's own key / value row, stored as row t·k_stride+n_keys−1 of the cache
 * before it is attended to (incremental self-attention); without them K / V are only read (cross-attention over the memory rows). */
int svpc_attn_q1_ln_supported(int D, int dh, int n_keys, int ldq, int ldkv, int ldnew, int ldx, int ldo);
int svpc_attn_q1_ln_fwd(const float* Q, int ldq, float* K, float* V, int ldkv, int k_stride, int n_keys, const float* newK, const float* newV,
                        int ldnew, const float* X, int ldx, const float* gamma, const float* beta, float eps, float* O, int ldo, int T, int D,
                        int dh, float scale, svpc_stream_t stream);
int svpc_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                  const float* LSE, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                  float* delta, const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                  float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);

/* MFMA (bf16 operands, fp32 softmax/accumulate) form of the same core for ≤128×128 (queries×keys) per sequence, dh 32/64 */
int svpc_attn_mfma_supported(int dh, int max_q, int max_k, int ldq, int ldk, int ldv, int dt /* 0 fp32, 1 bf16: the leading dimensions are
                             checked in 16-byte units of that element type */);
int svpc_attn_mfma_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, float* LSE,
                       const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                       float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
int svpc_attn_mfma_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                       const float* LSE, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                       const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                       float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);

int svpc_attn_mfma_fwd_t(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int dt, float* LSE,
                         const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                         float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* bf16x3 forward over split-stored Q / K / V / O (hi planes at the pointers with the rows' leading dimensions, lo planes q_lo / k_lo /
 * v_lo / o_lo elements behind them): Sᵀ and Oᵀ as three-term split-bf16 products, fp32 softmax, the library's dropout draws; non-causal,
 * ≤128 rows per sequence, head dim 32 / 64.  The backward is svpc_attn_mfma_bwd_t on the hi planes.  model.py:194-219 (clip encoder). */
int svpc_attn_stream_x3_fwd(const void* Q, int ldq, int q_lo, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, void* O,
                            int ldo, int o_lo, float* LSE, const int* seq, int n_seq, int H, int dh, int max_q, int max_k,
                            const float* key_mask, float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* one fp32 query per sequence against bf16 (k_lo = v_lo = 0) or split keys / values, forward and backward — the [CLS]-only last
 * clip-encoder layer of the training forward (model.py:1062-1064; core :194-219): a wave per (sequence, head), exact fp32 arithmetic on
 * the values read, no LDS; <= 128 keys, head dim 32 / 64.  Backward: dQ fp32, dK / dV dense bf16 rows (every key row is written). */
int svpc_attn_q1s_supported(int dh, int max_k, int ldq, int ldk, int ldv, int k_lo, int v_lo);
int svpc_attn_q1s_fwd(const float* Q, int ldq, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, float* O, int ldo, float* LSE,
                      const int* seq, int n_seq, int H, int dh, int max_k, const float* key_mask, float scale, float p_drop, unsigned site,
                      const svpc_u64* seed, svpc_stream_t stream);
int svpc_attn_q1s_bwd(const float* Q, int ldq, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, const float* dO, int lddo,
                      float* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int* seq, int n_seq, int H, int dh, int max_k,
                      const float* key_mask, float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* general bf16x3 forward: sequences of <= 32 queries and keys (the decoder's causal self-attention and its memory cross-attention,
 * model.py:620-663) run one wave per (sequence, head) and may be causal; longer ones take the stream kernel above (non-causal) */
int svpc_attn_x3_fwd(const void* Q, int ldq, int q_lo, const void* K, int ldk, int k_lo, const void* V, int ldv, int v_lo, void* O, int ldo,
                     int o_lo, float* LSE, const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal,
                     float scale, float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* Sequences of 33-104 rows (the clip encoder's 100 x 100 x 64) of the bf16 / split streams take a PERSISTENT forward inside
 * svpc_attn_mfma_fwd_t / svpc_attn_x3_fwd (attention_pipe.hip: a loader wave keeps the K / V planes of the next two (sequence, head)
 * pairs in flight by LDS-DMA while four waves compute the current one; same arithmetic, dropout draws and LSE).  This switch selects
 * it (1, default), the one-workgroup-per-pair kernels (0), or just reports (negative); returns the previous setting.  A/B use only. */
int svpc_attn_pipe_enable(int on);
/* timing experiments on that kernel (tools/dbg/pipe_phases.py, pipe_stamps.py): bits 1 no LDS-DMA, 2 no arithmetic, 4 no O stores,
 * 8 no Q loads, 16 s_memtime stamps of workgroups 0-15 into buf (16 x 128 u64 of device memory).  0 / NULL = production. */
int svpc_attn_pipe_debug(int bits, void* buf);
int svpc_attn_mfma_bwd_t(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo, int dt,
                         const float* LSE, const void* dO, int lddo, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv,
                         const int* seq, int n_seq, int H, int dh, int max_q, int max_k, const float* key_mask, int causal, float scale,
                         float p_drop, unsigned site, const svpc_u64* seed, svpc_stream_t stream);

/* ---- simulator recurrence: EntitiyReasoningNetwork.forward model.py:792-820 (Eqs. 2-7), one workgroup per video */
int svpc_sim_recur_fwd(const float* q, const float* c, const float* w4f, const float* E0, const int* step_off, const int* step_len,
                       const int* ent_off, const int* ent_len, int n_videos, int e_max, int D, float* e_out, float* ebar,
                       float* eall, svpc_stream_t stream);
int svpc_sim_recur_bwd(const float* q, const float* c, const float* w4f, const float* E0, const int* step_off, const int* step_len,
                       const int* ent_off, const int* ent_len, int n_videos, int e_max, int D, const float* e_out,
                       const float* ebar, const float* eall, const float* de, const float* debar, const float* deall, float* dq,
                       float* dc, float* dw4f, float* dE0, svpc_stream_t stream);

/* the simulator's two tiny heads in one launch each way: c = softmax(hh·W3^T + b3) (three choice weights per step, src/rtransformer/model.py:801)
 * and w = fb·W4^T + b4 (one scalar per step, :804-805).  Backward: dhh (T, D) and dfb (T, Wd) fully written; the weight / bias gradients as
 * per-workgroup partial sums part3 (groups, 3·D + 3) = [dW3 | db3], part4 (groups, Wd + 1) = [dW4 | db4], groups = svpc_sim_heads_groups(T),
 * for svpc_multi_finalize.  dc / dw may be NULL (no gradient). */
int svpc_sim_heads_groups(int T);
int svpc_sim_heads_fwd(const float* hh, const float* fb, const float* W3, const float* b3, const float* W4, const float* b4, float* c, float* w,
                       int T, int D, int Wd, svpc_stream_t stream);
int svpc_sim_heads_bwd(const float* hh, const float* fb, const float* W3, const float* W4, const float* c, const float* dc, const float* dw,
                       float* dhh, float* dfb, float* part3, float* part4, int T, int D, int Wd, svpc_stream_t stream);

/* ---- decoder cross-attention over the <= 3 memory rows of a sentence + residual LayerNorm, one launch forward and one backward per
 * layer: src/rtransformer/model.py:657-658 (BertDecoderLayerNoMemoryUntied.forward), attention core :194-219, BertLayerNorm :143-156.
 *   y = LayerNorm(x1 + MHA(query rows q; keys / values = the sentence's nm memory rows))
 * q, x1: T·lt rows; k, v: T·nm rows (the layer's key / value column blocks of the memory projection).  dt codes: 0 fp32, 1 bf16, 2 split (two
 * bf16 planes, the lo plane lo* columns behind).  probs (T·lt, H, 4), mean / rstd (T·lt) are saved for the backward, which returns the
 * gradients of the query rows and of the residual rows (dense, dg_dt), of the key / value rows (dense, dkv_dt, row stride ld_dkv) and the
 * per-sentence partial sums of [dgamma | dbeta] (T, 2D) for svpc_multi_finalize. */
int svpc_cross_attn_ln_supported(int D, int H, int lt, int nm);
int svpc_cross_attn_ln_fwd(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                           int kv_dt, int ld_kv, int lokv, const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy,
                           float* probs, float* mean, float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                           const svpc_u64* seed, svpc_stream_t stream);
int svpc_cross_attn_ln_bwd(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                           int kv_dt, int ld_kv, int lokv, const float* gamma, const float* probs, const float* mean, const float* rstd,
                           const void* dy, int dy_dt, int lddy, void* dq, void* dres, int dg_dt, int lddg, void* dk, void* dv, int dkv_dt,
                           int ld_dkv, float* part_ln, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                           const svpc_u64* seed, svpc_stream_t stream);
/* the same over RAGGED sentences — the decoder run over the valid tokens only (nothing a pad token computes reaches the loss: model.py:630-640
 * masks pad keys, the caption loss ignores pad labels): sentence s owns the rows [row_off[s], row_off[s] + row_len[s]) of q, x1, y, probs, mean,
 * rstd, dy, dq, dres; row_len[s] <= lt, the padded length (register bound of the kernels, stride of the dropout rows); NULL, NULL = uniform */
int svpc_cross_attn_ln_fwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* beta, float eps, void* y, int y_dt, int ldy, int loy,
                             float* probs, float* mean, float* rstd, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const svpc_u64* seed, const int* row_off, const int* row_len, svpc_stream_t stream);
int svpc_cross_attn_ln_bwd_r(const void* q, int q_dt, int ldq, int loq, const void* x1, int x_dt, int ldx, int lox, const void* k, const void* v,
                             int kv_dt, int ld_kv, int lokv, const float* gamma, const float* probs, const float* mean, const float* rstd,
                             const void* dy, int dy_dt, int lddy, void* dq, void* dres, int dg_dt, int lddg, void* dk, void* dv, int dkv_dt,
                             int ld_dkv, float* part_ln, int T, int lt, int nm, int D, int H, float scale, float p_drop, unsigned site,
                             const svpc_u64* seed, const int* row_off, const int* row_len, svpc_stream_t stream);

/* ---- pointer-generator + caption loss: model.py:896-923, :37-55 */
int svpc_ptr_attn_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, float* att, int T,
                      int lt, int e_max, int D, svpc_stream_t stream);
/* … with the generation gate p_gen = sigmoid([dec ; att]·w + b) (src/rtransformer/model.py:905-908) of every row computed in the same launch
 * (w: 2·D floats, b: one; att may be null): what a decoding iteration needs of the pointer, lt = 1 */
int svpc_ptr_attn_pgen_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, float* att,
                           const float* pgen_w, const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, svpc_stream_t stream);
int svpc_ptr_attn_bwd(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                      const float* datt, float* ddec, float* dproj, float* dbank, int T, int lt, int e_max, int D,
                      svpc_stream_t stream);
/* training form: pointer attention AND the generation gate of every row in one launch, forward and backward (model.py:899-908).  The
 * attended vector feeds only the gate, so it never reaches HBM; backward returns the per-step partial sums of [d pgen_w | d pgen_b] in
 * wpart (T rows of 2·D + 1 floats) for the table-driven finalizer (svpc_multi_finalize). */
int svpc_ptr_attn_gate_fwd(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, const float* pgen_w,
                           const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, svpc_stream_t stream);
int svpc_ptr_attn_gate_bwd(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                           const float* pgen, const float* dpgen, const float* pgen_w, float* ddec, float* dproj, float* dbank,
                           float* wpart, int T, int lt, int e_max, int D, svpc_stream_t stream);
/* the same two over RAGGED sentences — the head run over the valid tokens only (model.pack_text_rows): sentence j owns the rows
 * [row_off[j], row_off[j] + row_len[j]) of dec, pi, pgen (and of ddec, dpi, dpgen), row_len[j] <= lt (the padded length); NULL, NULL = uniform */
int svpc_ptr_attn_gate_fwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, float* pi, const float* pgen_w,
                             const float* pgen_b, float* pgen, int T, int lt, int e_max, int D, const int* row_off, const int* row_len,
                             svpc_stream_t stream);
int svpc_ptr_attn_gate_bwd_r(const float* dec, const float* proj, const float* bank, const int* step_ne, const float* pi, const float* dpi,
                             const float* pgen, const float* dpgen, const float* pgen_w, float* ddec, float* dproj, float* dbank,
                             float* wpart, int T, int lt, int e_max, int D, const int* row_off, const int* row_len, svpc_stream_t stream);
int svpc_ptr_mix_loss_fwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                          const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w, float* P,
                          float* loss_rows, int R, int V, int c_max, int e_max, float smoothing, svpc_stream_t stream);
int svpc_ptr_mix_loss_bwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                          const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w,
                          const float* P, const float* dP_ext, const float* dloss, float* dlogits, float* dg, float* dpi, int R, int V,
                          int c_max, int e_max, float smoothing, svpc_stream_t stream);
/* label_smoothing == 0: the reference then applies nn.CrossEntropyLoss(ignore_index=-1) to the PROBABILITIES (src/rtransformer/model.py:869-870
 * used at :960,982,1002,1014) — per video the mean over its non-ignored rows of logsumexp_{v<C}(P_v) − P_y.  row_w[r] = 1 / (valid rows of
 * row r's video), from svpc_ce_row_weights; loss_rows[r] = row_w[r]·(…) so that their sum is Σ_videos mean.  row_w == NULL: the
 * label-smoothed KL above (smoothing > 0 required). */
int svpc_ce_row_weights(const int* labels, const int* row_vid, int R, int n_vid, float* row_w, svpc_stream_t stream);
int svpc_ptr_mix_ce_fwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                        const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w, float* P,
                        float* loss_rows, int R, int V, int c_max, int e_max, float smoothing, const float* row_w, svpc_stream_t stream);
int svpc_ptr_mix_ce_bwd(const float* logits, const float* g, const float* pi, const int* labels, const int* row_c,
                        const int* row_vid, const int* csr_off, const int* csr_ent, const int* csr_id, const float* csr_w,
                        const float* P, const float* dP_ext, const float* dloss, float* dlogits, float* dg, float* dpi, int R, int V,
                        int c_max, int e_max, float smoothing, const float* row_w, svpc_stream_t stream);

/* ---- Gumbel-softmax (hard) re-sampling of the caption: reconstruct model.py:1018 */
int svpc_gumbel_noise(float* out, size_t n, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
int svpc_gumbel_fwd(const float* P, const float* noise, const int* row_c, const float* emb, float* bow, int* idx, float* stats, int R,
                    int c_max, int V, int W, float tau, svpc_stream_t stream);
int svpc_gumbel_bwd(const float* P, const float* noise, const int* row_c, const float* stats, const float* dy, float* dP, int R,
                    int c_max, int V, float tau, svpc_stream_t stream);
int svpc_gumbel_emb_grad(const float* dbow, const int* idx, const float* stats, float* demb, int R, int V, int W,
                         svpc_stream_t stream);

/* ---- row / span / loss kernels: Eq.(1) a/Σa model.py:798; softmax(W3 ĥ) :804; ingredient pooling :125-139; [CLS]+PE :1064;
 *      masked bag-of-words mean :1019-1021; nn.BCELoss(sum) :871; AsymmetricLoss libs/ASL/src/loss_functions/losses.py:15-50;
 *      nn.LSTM cell :865; nn.Embedding(padding_idx=0) gradient :492,:519 */
int svpc_rownorm_fwd(const float* a, float* y, int R, int C, int mode, svpc_stream_t stream);  /* mode 0: a/Σa, 1: softmax */
int svpc_rownorm_bwd(const float* dy, const float* y, const float* a, float* da, int R, int C, int mode, svpc_stream_t stream);
int svpc_span_mean_fwd(const float* x, const int* starts, const int* lens, const float* w, const float* add, const int* add_idx,
                       float* out, int G, int D, svpc_stream_t stream);
int svpc_span_mean_bwd(const float* dout, const int* starts, const int* lens, const float* w, float* dx, int G, int D,
                       svpc_stream_t stream);
int svpc_scatter_add_rows(const float* dx, const int* idx, float* dtable, int R, int D, int pad_row, svpc_stream_t stream);
int svpc_bce_rows_fwd(const float* p, const float* y, const int* widths, float* out, int R, int C, svpc_stream_t stream);
int svpc_bce_rows_bwd(const float* dout, const float* p, const float* y, const int* widths, float* dp, int R, int C,
                      svpc_stream_t stream);
int svpc_asl_rows_fwd(const float* p, const float* y, const float* active, float* out, int R, int C, float gneg, float gpos,
                      float clip, float eps, svpc_stream_t stream);
int svpc_asl_rows_bwd(const float* dout, const float* p, const float* y, const float* active, float* dp, int R, int C, float gneg,
                      float gpos, float clip, float eps, svpc_stream_t stream);
/* The whole loss sum in one launch (and its backward in one): total = sum(cap_rows) + [sum_r BCE(e_p[r], align[r]; first widths[r]
 * columns) + sum_{r: any(act[r] == 1)} ASL(a_p[r], act[r])] + lambda * [the same for the re-simulator's r_e / r_a] — reference
 * model.py:1110-1115 (per-video BCE-sum / ASL), :1168-1188 (re-simulation terms weighted by lambda_, total).  Any of e_p / a_p / r_e /
 * r_a may be NULL (modes without a simulator).  out5 = {total, caption, entity, action, re-simulation}.  Deterministic (row
 * partials in `workspace`, svpc_loss_tail_ws_floats() floats, summed in index order by the last workgroup; `counter`: one int zeroed
 * once by the caller, left at zero by every launch). */
int svpc_loss_tail_ws_floats(int n_cap, int R);
int svpc_loss_tail_fwd(const float* cap_rows, int n_cap, const float* e_p, const float* align, const int* widths, int R, int Ce,
                       const float* a_p, const float* act, int Ca, const float* r_e, const float* r_a, float lambda, float gneg,
                       float gpos, float clip, float eps, float* out5, float* workspace, int* counter, svpc_stream_t stream);
int svpc_loss_tail_bwd(const float* dout, int n_cap, const float* e_p, const float* align, const int* widths, int R, int Ce,
                       const float* a_p, const float* act, int Ca, const float* r_e, const float* r_a, float lambda, float gneg,
                       float gpos, float clip, float eps, float* d_cap, float* de_p, float* da_p, float* dr_e, float* dr_a,
                       svpc_stream_t stream);
int svpc_row_any_eq1(const float* x, float* out, int R, int C, svpc_stream_t stream);
/* token staging: up to 8 segments dst[i] = cast(src[idx ? idx[i] : i]) in one launch.  `segments`: array of
 *   struct { const void* src; const int* idx; void* dst; int src_dt, dst_dt, n; }   (dtype codes 0 fp32, 1 int64, 2 int32; dst fp32 / int32)
 * — the clip rows' and sentence rows' ids / masks / labels out of the loader's (step, video) tensors, reference train.py:91-112 +
 * model.py:1038-1042, 925-1015 (there: slicing inside a Python loop over steps and videos). */
int svpc_gather_cast_multi(const void* segments, int n, svpc_stream_t stream);
/* the same, advancing the step's dropout / Gumbel seed (the update of svpc_bump_seed) in the same launch when bump_seed != NULL */
int svpc_gather_cast_multi_seed(const void* segments, int n, svpc_u64* bump_seed, svpc_stream_t stream);
int svpc_clamp_labels(const int* in, int* out, int n, int vocab, int unk, svpc_stream_t stream); /* model.py:1013 */
int svpc_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, const float* h_prev, const float* active, float* h,
                       float* c, float* gates_act, int N, int D, svpc_stream_t stream);
int svpc_lstm_cell_bwd(const float* dh, const float* dc, const float* gates_act, const float* c_prev, const float* active,
                       float* dgates, float* dc_prev, float* dh_prev, int N, int D, svpc_stream_t stream);
/* on-device logging counters (doubles in HBM; one read-back per logging interval instead of ≈10 .item() syncs per step):
 * src/train.py:32-38 cal_performance → counters[0] += #labelled rows, counters[1] += #rows whose first-index arg-max == label;
 * src/train.py:40-49 calculate_f1   → counters[0] += Σ gold[prob>0.5], [1] += Σ gold, [2] += #(prob>0.5) */
int svpc_metric_argmax(const float* scores, int ld, int R, int C, const long long* labels, int ignore, double* counters,
                       svpc_stream_t stream);
int svpc_metric_f1(const float* prob, const float* gold, size_t n, double* counters, svpc_stream_t stream);
/* input staging: gather the frame windows of a batch out of an HBM-resident feature bank (idx < 0 → zero row) —
 * recursive_caption_dataset.py:187-189 (resnet‖bn concat), :389-416 (window / down-sample / [CLS]…[SEP] layout), train.py:91 (H2D);
 * and the video half of input_ids / input_mask built on the device from the valid-frame counts (:409-415) */
int svpc_gather_rows_f32(const float* src, const int* idx, float* dst, int rows, int width, svpc_stream_t stream);
int svpc_video_tokens(const int* n_valid, long long* ids, float* mask, int clips, int Lv, int L, int cls_id, int vid_id, int sep_id,
                      int pad_id, svpc_stream_t stream);
/* sequence forms for the fused recurrence (nn.LSTM model.py:859-860, used at :1022-1024): gx rows gathered in place; the
 * backward adds the gradient from the layer above (dh_out) to the recurrent one (dh_rec) */
int svpc_lstm_cell_fwd_idx(const float* gx_all, const int* rows, const float* gh, const float* c_prev, const float* h_prev,
                           const float* active, float* h, float* c, float* gates_act, int N, int D, svpc_stream_t stream);
int svpc_lstm_cell_bwd_seq(const float* dh_out, const float* dh_rec, const float* dc, const float* gates_act, const float* c_prev,
                           const float* active, float* dgates, float* dc_prev, float* dh_prev, int N, int D, svpc_stream_t stream);
/* both directions of the BiLSTM at one time step in one launch: every argument is a HOST array of 2 device pointers
 * (direction 0 = forward in time, 1 = reverse); `active` is shared */
int svpc_lstm_pair_fwd(const float* const* gx, const int* const* rows, const float* const* gh, const float* const* c_prev,
                       const float* const* h_prev, const float* active, float* const* h, float* const* c, float* const* gates, int N, int D,
                       svpc_stream_t stream);
/* one time step of both directions with the recurrent projection inside: gates = gx[rows] + h_prev·W_hhᵀ (bf16 MFMA operands, fp32
 * accumulate) and the cell in ONE launch — a workgroup owns 8 hidden units (their i|f|g|o rows of W_hh), so no (N, 4D) buffer goes
 * through HBM between a GEMM and a cell launch.  D % 16 == 0.  Arguments as svpc_lstm_pair_fwd (w_hh[z]: (4D, D) row-major). */
int svpc_lstm_pair_step_fwd(const float* const* h_prev, const float* const* c_prev, const float* const* w_hh, const float* const* gx,
                            const int* const* rows, const float* active, float* const* h, float* const* c, float* const* gates, int N,
                            int D, svpc_stream_t stream);
int svpc_lstm_pair_step_fwd_x3(const float* const* h_prev, const float* const* c_prev, const float* const* w_hh, const float* const* gx,
                               const int* const* rows, const float* active, float* const* h, float* const* c, float* const* gates, int N,
                               int D, svpc_stream_t stream);   /* the same with bf16x3 products */
int svpc_lstm_pair_bwd(const float* const* dh_out, const float* const* dh_rec, const float* const* dc, const float* const* gates,
                       const float* const* c_prev, const float* active, float* const* dgates, float* const* dc_prev,
                       float* const* dh_prev, int N, int D, svpc_stream_t stream);
/* the same when the recurrent dgrad dgates·W_hh of the previous time step was computed in n_parts k-parts (more workgroups pulling
 * the weight): dh_parts[z] = n_parts slabs of N·D floats, added to dh_rec in part order */
int svpc_lstm_pair_bwd_parts(const float* const* dh_out, const float* const* dh_rec, const float* const* dh_parts, int n_parts,
                             const float* const* dc, const float* const* gates, const float* const* c_prev, const float* active,
                             float* const* dgates, float* const* dc_prev, float* const* dh_prev, int N, int D, svpc_stream_t stream);
/* greedy decoding step: argmax with the UNK column suppressed + OOV→UNK remap, src/translator.py:104-112 */
int svpc_greedy_pick(const float* scores, int ld, const int* row_c, const int* row_x, int n_sent, int lt, int pos, int unk,
                     int* next_ext, int* next_model, svpc_stream_t stream);
/* … and the picked ids also stored as column `col` of the sentence-major (n_sent, ld_out) id matrices the decoding loop keeps
 * (text_out: model-side ids, ext_out: extended ids; src/translator.py:96-99 writes them at the top of the next iteration) */
int svpc_greedy_pick_append(const float* scores, int ld, const int* row_c, const int* row_x, int n_sent, int lt, int pos, int unk,
                            int* next_ext, int* next_model, int* text_out, int* ext_out, int ld_out, int col, svpc_stream_t stream);
/* rows between storage kinds in one launch (data movement): dst[r] = convert(src[idx ? idx[r] : r]); kinds 0 fp32, 1 bf16, 2 split (two bf16
 * planes, the lo plane lo_* columns behind the hi plane).  Where rows join or leave an activation stream: the decoder's memory rows
 * (src/rtransformer/model.py:939-947) entering the split stream, its output leaving it (:1086), the [CLS] rows of the clip stream (:1062-1064). */
int svpc_rows_move(const void* src, int src_kind, int ld_src, int lo_src, const int* idx, void* dst, int dst_kind, int ld_dst, int lo_dst,
                   int R, int W, svpc_stream_t stream);
/* table[idx[r]] += rows[r], fp32 rows into a dense bf16 table (distinct idx): the gradient of gathered stream rows joins the stream's
 * gradient in place */
int svpc_scatter_add_rows_bf16(const float* rows, const int* idx, void* table, int ld_table, int R, int W, svpc_stream_t stream);
/* rows of two (R', W) fp32 tables through two index lists in one launch — the two directions of the BiLSTM (model.py:1022-1024):
 * mode 0: out[r] = a[ia[r]] + b[ib[r]];  mode 1: a[ia[r]] = b[ib[r]] = out[r];  mode 2: out[r] = a[ia[r]], out2[r] = b[ib[r]] */
int svpc_pair_rows(float* a, const int* ia, float* b, const int* ib, float* out, float* out2, int R, int W, int mode, svpc_stream_t stream);
/* out = bf16(a + b), a bf16 (NULL: a plain cast), b fp32: the two gradients of a stream tensor that is also consumed as fp32 rows (the
 * decoder's last rows: the LM head reads the stream, the pointer attention its fp32 copy — model.py:896-923) in one launch */
int svpc_add_cast_bf16(const void* a, const float* b, void* out, size_t n, svpc_stream_t stream);
/* out[r] = inv[r] >= 0 ? src[inv[r]] : 0 for r < n_rows — the rows of a packed (valid tokens only) tensor back in the padded layout the
 * reference's return values have (model.py:1105-1115), zeros at the pad positions; fp32 rows of W floats */
int svpc_rows_expand(const float* src, const int* inv, float* out, int n_rows, int W, svpc_stream_t stream);
int svpc_add(const float* a, const float* b, float* c, size_t n, svpc_stream_t stream);
int svpc_sum_all(const float* x, size_t n, float* out, float scale, svpc_stream_t stream);
int svpc_fill_from(float* x, size_t n, const float* v, svpc_stream_t stream);
int svpc_dropout_mask(float* out, size_t n, float p, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
/* keep mask (1 / 0) of the dropout on the attention probabilities (model.py:213), (n_rows, max_k) with rows (sequence·H + head)·max_q +
 * query: the draw every attention kernel of the library makes for that element — one full hash per ROW, one add + xor-shift + 24-bit
 * multiply per element (the kernels are bound by vector-instruction issue).  For tests: the fp32 reference applies the kernels' own mask. */
int svpc_attn_dropout_mask(float* out, size_t n_rows, int max_k, float p, unsigned site, const svpc_u64* seed, svpc_stream_t stream);
int svpc_bump_seed(svpc_u64* seed, svpc_stream_t stream);

/* ---- fused training-step tail: clip_grad_norm_ train.py:141-142, BertAdam optimization.py:284-331, EMA :196-203.
 *      meta = device array of {float* p,g,m,v,ema; long long n; float wd; int pad; bf16* shadow (or NULL); bf16* shadow_lo (or NULL)};
 *      chunk tables built by the host.  shadow[i] = bf16(p[i]) is refreshed by the Adam kernel (operand storage of svpc_gemm_glds),
 *      shadow_lo[i] = bf16(p[i] - shadow[i]) likewise (second plane of svpc_gemm_p8x3's B operand). */
int svpc_opt_chunk(void);
int svpc_opt_meta_bytes(void);
int svpc_opt_step(const void* meta, const int* chunk_tid, const long long* chunk_start, const int* tensor_chunk_off, int n_tensors,
                  int n_chunks, float* partial, float* norms_sq, const float* hyper, svpc_stream_t stream);
int svpc_opt_zero_grad(const void* meta, const int* chunk_tid, const long long* chunk_start, int n_chunks, svpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SVPC_HIP_H */
