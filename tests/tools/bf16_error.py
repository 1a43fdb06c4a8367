"""Where does the bf16 throughput mode's error come from?  Headline shape, 'drawn' weights, eval mode, injected Gumbel noise; the
fp32 mode on the GPU (≡ the oracle to 1e-7, tests/test_headline_parity.py) is the reference.  Ablations toggle one thing at a time."""
import os, sys, copy, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from svpc_amd import ops, synthetic as syn

DEV = "cuda:0"
mt = sys.argv[1] if len(sys.argv) > 1 else "vivt"
args = bench.parse_args([])
cfg, model = bench.build(args, "cpu", model_type=mt)
drawn = syn.draw_parameters(list(model.named_parameters()), seed=7)
with torch.no_grad():
    for n, p in model.named_parameters():
        p.copy_(drawn[n])
model.eval().to(DEV)
batch = bench.device_batch(cfg, args, torch.device(DEV), seed=2019)
g = torch.Generator().manual_seed(99)
model.gumbel_noise = [(-torch.empty(12, cfg.max_t_len, cfg.vocab_size).exponential_(generator=g).log()).to(DEV) for _ in range(16)]
fargs = syn.forward_args(batch)
NAMES = ["video_embeddings.video_embeddings.2.weight", "encoder.layer.0.attention.self.query.weight", "encoder.layer.5.attention.self.value.weight",
         "decoder.layer.0.self_attention.key.weight", "decoder.layer.3.output.dense.weight", "decoder_classifier.decoder.weight",
         "Wing.weight", "Went.0.weight", "recipe_encoder.weight_hh_l0", "text_embeddings.word_fc.2.weight", "reasoner.W2.weight"]


def run():
    model.zero_grad(set_to_none=True)
    loss, probs, _, _ = model(*fargs)
    loss.backward()
    torch.cuda.synchronize()
    P = torch.cat([p.reshape(-1, p.shape[-1]) for p in probs]).detach().clone()
    return float(loss), P, {n: dict(model.named_parameters())[n].grad.detach().clone() for n in NAMES}


ops.set_precision("fp32")
ref = run()
orig_ok, orig_gemm = ops.bf16_stream_ok, ops._gemm


def report(tag, res):
    loss, P, gr = res
    out = {"loss_rel": abs(loss - ref[0]) / abs(ref[0]), "prob_abs_max": float((P - ref[1]).abs().max()),
           "prob_abs_mean": float((P - ref[1]).abs().mean()), "argmax_agree": float((P.argmax(-1) == ref[1].argmax(-1)).float().mean())}
    for n in NAMES:
        a, b = gr[n].double().reshape(-1), ref[2][n].double().reshape(-1)
        out[n] = "norm %.2e cos %.5f elem %.3f" % (abs(float(a.norm() - b.norm())) / float(b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm())),
                                                  float((a - b).abs().max() / b.abs().max()))
    print("==", tag)
    for k, v in out.items():
        print("   %-48s %s" % (k, v if isinstance(v, str) else "%.3e" % v))
    sys.stdout.flush()


def precise_small(limit_rows):
    """GEMMs with at most ``limit_rows`` rows of fp32 operands run on the f32 MFMA path"""
    def g(A, lda, a_kc, B, ldb, b_kc, C, M, N, K, **kw):
        small = A.dtype == torch.float32 and B.dtype == torch.float32 and C.dtype == torch.float32 and max(M, N if a_kc == 0 else 0) <= limit_rows
        if small:
            ops._PRECISION = "fp32"
            try:
                return orig_gemm(A, lda, a_kc, B, ldb, b_kc, C, M, N, K, **kw)
            finally:
                ops._PRECISION = "bf16"
        return orig_gemm(A, lda, a_kc, B, ldb, b_kc, C, M, N, K, **kw)
    return g


try:
    ops.set_precision("bf16")
    report("A  bf16 mode as benchmarked (bf16 MFMA everywhere, bf16 encoder + decoder streams)", run())
    ops.BF16_STREAM = False
    report("B  bf16 MFMA operands everywhere, fp32 activation storage (no bf16 streams)", run())
    ops.BF16_STREAM = True
    ops.bf16_stream_ok = lambda rows, *d: orig_ok(rows, *d) and rows > 5000
    report("C  bf16 encoder stream only (decoder activations fp32 in HBM, bf16 MFMA operands)", run())
    ops._gemm = precise_small(300)
    report("D  C + every GEMM of <= 300 rows (step level: simulators, LSTM, step encoder, memory K/V, Wing) on f32 MFMA", run())
    ops._gemm = precise_small(5000)
    report("E  C + every text-side GEMM (<= 5000 rows: decoder, head, pointer, step level) on f32 MFMA: only the clip encoder is bf16", run())
    ops.bf16_stream_ok = orig_ok
    ops._gemm = precise_small(300)
    report("F  A + step-level GEMMs (<= 300 rows) on f32 MFMA", run())
finally:
    ops._gemm, ops.bf16_stream_ok = orig_gemm, orig_ok
    ops.set_precision("fp32")
