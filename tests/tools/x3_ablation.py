"""The price of parity, family by family: at the headline shape (vivt, N=16 x S=12 x Lv=100 x F=3072, D=768, L=6),
eval mode, injected Gumbel noise, ONE family of contractions of the bf16x3 mode at a time is degraded — both operands' lo planes
dropped ("1-term": the plain bf16 product), only the weights' ("2-term, weights bf16"), only the activations' ("2-term, activations
bf16") — or one family of STORED tensors loses its lo plane (plain bf16 storage), and the loss error against the CPU oracle, the
worst / mean probability error, the arg-max agreement, the Gumbel arg-max flips and (test-sensitive weights) the agreement of the
greedy ids at config 5's size are recorded.  The degraded operands go through the UNCHANGED three-term kernels with a zero lo plane
(svpc_amd.ops.ABLATE), forward only.  → profiles/r04_x3_ablation.json (written under gpurun_out/ on the GPU box).
usage: python tests/tools/x3_ablation.py [--no-decode] [--init drawn|bench|both]"""
import argparse, copy, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_headline_parity as H
from svpc_amd import ops, synthetic as syn
from svpc_amd.optim import WeightStore

ap = argparse.ArgumentParser()
ap.add_argument("--no-decode", action="store_true")
ap.add_argument("--init", default="both")
args = ap.parse_args()
DEV = "cuda:0"

ENC = r"^encoder\.layer\.\d+\."
FAMILIES = [      # (label, regex on the parameter name of the weight (linear) / gain (LayerNorm))
    ("video embedding projection (K = 3072)", r"^video_embeddings\.video_embeddings\.2\.weight"),
    ("clip encoder: packed Q|K|V projection", ENC + r"attention\.self\.query\.weight"),
    ("clip encoder: attention-out projection", ENC + r"attention\.output\.dense\.weight"),
    ("clip encoder: FFN in (GELU)", ENC + r"hidden_intermediate\.dense\.weight"),
    ("clip encoder: FFN out", ENC + r"output\.dense\.weight"),
    ("decoder: every projection (self / cross Q, stacked memory K|V, output)", r"^decoder\.layer\.\d+\..*weight"),
    ("step-wise encoder: every projection", r"^step_wise_encoder\.layer\.\d+\..*(query|dense)\.weight"),
    ("head (transform + vocabulary projection)", r"^decoder_classifier\..*weight"),
    ("pointer (Wing, pgen), memory slots (Went, Wac)", r"^(Wing|pgen_linear\.0|Went\.0|Wac\.0)\.weight"),
    ("simulators (action selector, W1-W4), both", r"^(reasoner|recipe_reasoner)\.(action_selector\.\d|W1\.0|W2|W3|W4)\.weight"),
    ("text / ingredient embeddings (word_fc)", r"^(text|ingredient)_embeddings\.word_fc\.2\.weight"),
]
STORAGE = [
    ("clip stream: LayerNorm outputs stored as plain bf16 (attention-out LN, FFN LN, embedding LN)",
     r"^(encoder\.layer\.\d+\.(attention\.output|output)\.LayerNorm|video_embeddings\.video_embeddings\.4)\.weight"),
    ("clip stream: projection outputs stored as plain bf16 (Q|K|V, attention-out, FFN in / out)",
     ENC + r"(attention\.self\.query|attention\.output\.dense|hidden_intermediate\.dense|output\.dense)\.weight"),
    ("decoder stream: every stored tensor plain bf16", r"^decoder\.layer\.\d+\..*weight"),
]


def names_of(model):
    store = WeightStore.for_model(model)
    names = {p.data_ptr(): n for n, p in model.named_parameters()}
    return store, names


def flips(probs, ref_probs, noise):
    f = t = 0
    for p, r, n in zip(probs, ref_probs, noise):
        V = n.shape[-1]
        a = (torch.log(p.detach().cpu()[..., :V] + 1e-12) + n).argmax(-1); b = (torch.log(r[..., :V] + 1e-12) + n).argmax(-1)
        f += int((a != b).sum()); t += a.numel()
    return f, t


def measure(model, batch_dev, noise, ref):
    with torch.no_grad():
        loss, probs, _, _ = model(*syn.forward_args(batch_dev))
    torch.cuda.synchronize()
    perr = max(float((p.cpu() - r).abs().max()) for p, r in zip(probs, ref["probs"]))
    pmean = float(np.mean([float((p.cpu() - r).abs().mean()) for p, r in zip(probs, ref["probs"])]))
    agree = float(np.mean([float((p.cpu().argmax(-1) == r.argmax(-1)).float().mean()) for p, r in zip(probs, ref["probs"])]))
    f, t = flips(probs, ref["probs"], noise)
    return dict(loss_rel=abs(float(loss) - ref["loss"]) / abs(ref["loss"]), prob_abs_max=perr, prob_abs_mean=pmean, argmax_agreement=agree,
                gumbel_flips=f, gumbel_positions=t)


def decode_agreement(model, cfg, c5):
    from svpc_amd.translator import Translator
    cfg5, _, batch5, ref5 = c5
    tr = Translator(type("O", (), {"cuda": True})(), {"model_cfg": cfg5, "model": model.state_dict()}, model=model, graph=False)
    dec, _ = tr.translate_batch(syn.translate_inputs(H._to_dev(batch5)))
    same = total = 0
    for d, r in zip(dec, ref5):
        same += int((d.cpu() == r).sum()); total += r.numel()
    return same / total


out = {"shape": "vivt, 16 videos x 12 clips x 100 frames x 3072, D=768, H=12, L=6 (the headline workload), eval mode, injected Gumbel noise",
       "how": "one family at a time through the unchanged three-term kernels with a zero lo plane on the degraded operand (ops.ABLATE)", "cases": {}}
inits = ["drawn", "bench"] if args.init == "both" else [args.init]
for init in inits:
    t0 = time.time()
    cfg, model_cpu, batch, noise, ref = H._case("vivt", init)
    print("[%s] oracle %.0f s" % (init, time.time() - t0), flush=True)
    c5 = None
    if not args.no_decode and init == "drawn":
        import test_config5_gpu as C5
        t0 = time.time()
        c5 = C5._config5_case("drawn")
        print("[drawn] config-5 oracle decode %.0f s" % (time.time() - t0), flush=True)
    ops.set_precision("bf16x3")
    model = copy.deepcopy(model_cpu).to(DEV); model.eval()
    model.gumbel_noise = [n.to(DEV) for n in noise]
    store, names = names_of(model)
    bdev = H._to_dev(batch)
    model5 = None
    if c5 is not None:
        model5 = copy.deepcopy(c5[1]).to(DEV); model5.eval()
        _, names5 = names_of(model5)
    rows = []

    def run(label, match=None, attn=None, kind="", flags=""):
        ops.ABLATE = None if (match is None and attn is None) else {"names": names, "match": (match or (lambda n: "")), "attn": attn}
        r = measure(model, bdev, noise, ref)
        if model5 is not None:
            if ops.ABLATE is not None:
                ops.ABLATE = dict(ops.ABLATE, names=names5)
            r["config5_token_agreement"] = decode_agreement(model5, cfg, c5)
        ops.ABLATE = None
        r.update(family=label, degraded=kind, flags=flags)
        rows.append(r)
        print("%-6s %-92s %-28s loss %.2e  p_max %.1e  argmax %.5f  flips %d%s" % (init, label[:92], kind, r["loss_rel"], r["prob_abs_max"],
              r["argmax_agreement"], r["gumbel_flips"], ("  c5 %.4f" % r["config5_token_agreement"]) if "config5_token_agreement" in r else ""), flush=True)
    run("(none: the bf16x3 mode as it ships)")
    anyw = lambda n: True
    run("EVERY projection", match=lambda n: "ab", kind="1-term products", flags="ab")
    run("EVERY projection", match=lambda n: "b", kind="weights bf16 (2-term)", flags="b")
    run("EVERY projection", match=lambda n: "a", kind="activations bf16 (2-term)", flags="a")
    for label, rx in FAMILIES:
        pat = re.compile(rx)
        for fl, kind in (("ab", "1-term products"), ("b", "weights bf16 (2-term)"), ("a", "activations bf16 (2-term)")):
            run(label, match=(lambda n, pat=pat, fl=fl: fl if pat.search(n) else ""), kind=kind, flags=fl)
    for fl, kind in (("qk", "Q, K operands bf16"), ("v", "V operand bf16"), ("qkv", "Q, K, V operands bf16")):
        run("clip encoder: attention core (S = QK', O = PV)", attn=fl, kind=kind, flags=fl)
    for label, rx in STORAGE:
        pat = re.compile(rx)
        run(label, match=(lambda n, pat=pat: "o" if pat.search(n) else ""), kind="plain bf16 storage", flags="o")
    out["cases"][init] = rows
    ops.set_precision("fp32")
    del model, model5
    torch.cuda.empty_cache()
GATE = dict(prob_abs_max=8e-4, argmax_agreement=1.0, gumbel_flips=0, config5_token_agreement=1.0)      # tests/test_headline_parity.py, test_config5_gpu.py
out["gate"] = GATE
for init, rows in out["cases"].items():
    for r in rows:
        r["passes_gate"] = bool(r["prob_abs_max"] <= GATE["prob_abs_max"] and r["argmax_agreement"] >= 1.0 and r["gumbel_flips"] == 0 and
                                r.get("config5_token_agreement", 1.0) >= 1.0)
drop = {}
for init, rows in out["cases"].items():
    for r in rows[1:]:
        drop.setdefault((r["family"], r["degraded"]), []).append(r["passes_gate"])
out["droppable_under_the_gate_on_every_weight_set"] = [" / ".join(k) for k, v in drop.items() if all(v)]
if "bench" in out["cases"]:
    out["droppable_on_the_bench_weights_only"] = [" / ".join((r["family"], r["degraded"])) for r in out["cases"]["bench"][1:] if r["passes_gate"]]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r04_x3_ablation.json"), "w") as f:
    json.dump(out, f, indent=1)
print("written gpurun_out/r04_x3_ablation.json")
