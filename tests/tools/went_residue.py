"""Went.0.{weight,bias}: 5e-4 / 1.8e-3 from the fp32 oracle where the oracle is conditioned to 1.6e-6 (fp64 leg audit): compare the visual
simulator's ebar, the ReLU mask of went = relu(Went ebar) and the gradient reaching went, GPU fp32 mode vs oracle."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_headline_parity as H
from oracle import svpc_oracle as orc
from svpc_amd import ops, synthetic as syn

cfg, model, batch, noise = H._build("vivt", "drawn")
stash = {}
orig = orc.simulator
def sim_spy(P, pre, step_vecs, ent, training=False):
    out = orig(P, pre, step_vecs, ent, training)
    if pre == "reasoner":
        out[2].retain_grad()
        stash.setdefault("ebar", []).append(out[2])
    return out
orc.simulator = sim_spy
P = {k: v.detach().clone() for k, v in model.state_dict().items()}
for n, _ in model.named_parameters():
    P[n].requires_grad_(True)
torch.set_num_threads(16)
loss, probs, ents, acts = orc.forward(P, cfg, *syn.forward_args(batch), gumbel_noise=noise)
loss.backward()
o_ebar = torch.cat(stash["ebar"], 0).detach(); o_debar = torch.cat([t.grad for t in stash["ebar"]], 0)
ops.set_precision("fp32")
gm = copy.deepcopy(model).to("cuda:0"); gm.eval()
gm.gumbel_noise = [n.to("cuda:0") for n in noise]
g = {}
run0 = gm.reasoner.run
def run_spy(seq, ents_, plan_sim, cx):
    out = run0(seq, ents_, plan_sim, cx)
    out[2].retain_grad(); g["ebar"] = out[2]
    return out
gm.reasoner.run = run_spy
lin0 = ops.linear
def lin_spy(x, w, b=None, **kw):
    y = lin0(x, w, b, **kw)
    if w is gm.Went[0].weight:
        y.retain_grad(); g["went"] = y
    return y
ops.linear = lin_spy
import svpc_amd.model as M
tot = gm(*syn.forward_args(H._to_dev(batch)))[0]
tot.backward(); ops.join_side(); torch.cuda.synchronize()
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))
W, b = P["Went.0.weight"].detach().double(), P["Went.0.bias"].detach().double()
z_o = o_ebar.double() @ W.t() + b
z_g = g["ebar"].detach().cpu().double() @ W.t() + b
print("ebar value gpu-vs-oracle %.2e ; d(ebar) %.2e" % (rel(g["ebar"], o_ebar), rel(g["ebar"].grad, o_debar)))
print("pre-activation z: max |z_gpu - z_oracle| %.3e ; smallest |z_oracle| %.3e ; relu-mask mismatches %d of %d" % (
    float((z_g - z_o).abs().max()), float(z_o.abs().min()), int(((z_g > 0) != (z_o > 0)).sum()), z_o.numel()))
went_k = g["went"].detach().cpu().double()
print("kernel went vs relu(z_gpu): %.2e ; kernel mask vs oracle mask mismatches %d" % (float((went_k - z_g.clamp(min=0)).abs().max()), int(((went_k > 0) != (z_o > 0)).sum())))
dz_g = (g["went"].grad.cpu().double() * (went_k > 0))
print("bias grad: gpu %s" % "")
gb = gm.Went[0].bias.grad.cpu().double(); ob = P["Went.0.bias"].grad.double()
print("Went.bias grad rel(norm) %.3e ; from captured dz: %.3e" % (float((gb - ob).norm() / ob.norm()), float((dz_g.sum(0) - ob).norm() / ob.norm())))
d = (gb - ob).abs(); i = int(d.argmax()); print("worst bias element %d gpu %.6g oracle %.6g" % (i, float(gb[i]), float(ob[i])))
