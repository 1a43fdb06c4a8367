"""Isolate the fp32 mode's 2.5e-3 gradient residue on recipe_reasoner.* (DESIGN §10.4): the simulator module alone, drawn weights,
inputs of two kinds (LayerNorm-like step vectors as the visual simulator sees, LSTM-like small ones as the re-simulator sees), a random
linear functional of (e, a) as the loss; GPU fp32 mode vs oracle.simulator in float32 and float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_headline_parity as H
from oracle import svpc_oracle as orc
from svpc_amd import ops
from svpc_amd.model import _Ctx
from svpc_amd.ops_common import Idx

cfg, model, batch, noise = H._build("vivt", "drawn")
P = {k: v.detach().clone() for k, v in model.state_dict().items()}
dev = "cuda:0"
ops.set_precision("fp32")
gm = model.to(dev); gm.eval()
S, E, D, A = 12, 10, cfg.hidden_size, cfg.action_vocab_size
for pre in ("reasoner", "recipe_reasoner"):
    for kind, sscale in (("layernorm-like", 1.0), ("lstm-like", 0.15), ("tiny", 0.02)):
        g_ = torch.Generator().manual_seed(5)
        seq = torch.randn(S, D, generator=g_) * sscale
        ent = torch.randn(E, D, generator=g_) * 0.5
        r1 = torch.randn(S, E, generator=g_); r2 = torch.randn(S, A, generator=g_)
        names = [n for n in P if n.startswith(pre + ".") and P[n].dtype.is_floating_point]
        res = {}
        for dt in (torch.float32, torch.float64):
            Pk = {n: (v.to(dt).clone().requires_grad_(True) if n in names else v.to(dt) if v.dtype.is_floating_point else v) for n, v in P.items()}
            s_ = seq.detach().clone().to(dt).requires_grad_(True); e_ = ent.detach().clone().to(dt).requires_grad_(True)
            eo, ao, _, _, _ = orc.simulator(Pk, pre, s_, e_)
            ((eo * r1.to(dt)).sum() + (ao * r2.to(dt)).sum()).backward()
            res[dt] = ({n: Pk[n].grad.double() for n in names if Pk[n].grad is not None}, s_.grad.double(), e_.grad.double(), eo.detach().double())
        mod = getattr(gm, pre)
        for p in mod.parameters():
            p.grad = None
        sg = seq.detach().clone().to(dev).requires_grad_(True); eg = ent.detach().clone().to(dev).requires_grad_(True)
        cx = _Ctx(cfg, False, gm.rng(dev))
        eo, ao, _, _, _ = mod.run(sg, eg, (Idx([0]), Idx([S]), Idx([0]), Idx([E]), E), cx)
        ((eo[:, :E] * r1.to(dev)).sum() + (ao * r2.to(dev)).sum()).backward()
        ops.join_side(); torch.cuda.synchronize()
        gg = {pre + "." + n: p.grad.detach().cpu().double() for n, p in mod.named_parameters() if p.grad is not None}
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        print("== %s, %s inputs: e in [%.3g, %.3g]; forward e: gpu vs f32 %.2e" % (pre, kind, float(res[torch.float64][3].min()), float(res[torch.float64][3].max()),
              rel(eo.detach().cpu().double()[:, :E], res[torch.float32][3])))
        print("   d(step vectors): gpu-vs-f32 %.2e  f32-vs-f64 %.2e | d(entities): %.2e  %.2e" % (rel(sg.grad.cpu().double(), res[torch.float32][1]),
              rel(res[torch.float32][1], res[torch.float64][1]), rel(eg.grad.cpu().double(), res[torch.float32][2]), rel(res[torch.float32][2], res[torch.float64][2])))
        for n in sorted(gg):
            if n in res[torch.float32][0]:
                print("   %-44s gpu-vs-f32 %.2e  gpu-vs-f64 %.2e  f32-vs-f64 %.2e" % (n, rel(gg[n], res[torch.float32][0][n]), rel(gg[n], res[torch.float64][0][n]),
                      rel(res[torch.float32][0][n], res[torch.float64][0][n])))
