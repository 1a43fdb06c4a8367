"""Where the fp32 mode's 2.5e-3 residue on recipe_reasoner.* enters: full headline-shape step, test-sensitive weights; the
re-simulator's inputs, outputs and their gradients on the GPU (fp32 mode) vs the CPU oracle (float32)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_headline_parity as H
from oracle import svpc_oracle as orc
from svpc_amd import ops, synthetic as syn

cfg, model, batch, noise = H._build("vivt", "drawn")
# ---- oracle, stashing the re-simulator's tensors
stash = {}
orig = orc.simulator
def sim_spy(P, pre, step_vecs, ent, training=False):
    out = orig(P, pre, step_vecs, ent, training)
    if pre == "recipe_reasoner":
        step_vecs.retain_grad(); out[0].retain_grad(); out[1].retain_grad()
        stash.setdefault("seq", []).append(step_vecs); stash.setdefault("e", []).append(out[0]); stash.setdefault("a", []).append(out[1])
    return out
orc.simulator = sim_spy
P = {k: v.detach().clone() for k, v in model.state_dict().items()}
for n, _ in model.named_parameters():
    P[n].requires_grad_(True)
torch.set_num_threads(16)
loss, probs, ents, acts = orc.forward(P, cfg, *syn.forward_args(batch), gumbel_noise=noise)
loss.backward()
o_seq = torch.cat(stash["seq"], 0); o_e = torch.cat([e for e in stash["e"]], 0); o_a = torch.cat(stash["a"], 0)
o_dseq = torch.cat([t.grad for t in stash["seq"]], 0); o_de = torch.cat([t.grad for t in stash["e"]], 0); o_da = torch.cat([t.grad for t in stash["a"]], 0)
# ---- GPU fp32 mode
ops.set_precision("fp32")
gm = copy.deepcopy(model).to("cuda:0"); gm.eval()
gm.gumbel_noise = [n.to("cuda:0") for n in noise]
g = {}
run0 = gm.recipe_reasoner.run
def run_spy(seq, ents_, plan_sim, cx):
    seq.retain_grad()
    out = run0(seq, ents_, plan_sim, cx)
    out[0].retain_grad(); out[1].retain_grad()
    g.update(seq=seq, e=out[0], a=out[1])
    return out
gm.recipe_reasoner.run = run_spy
tot = gm(*syn.forward_args(H._to_dev(batch)))[0]
tot.backward(); ops.join_side(); torch.cuda.synchronize()
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))
E = o_e.shape[1]
print("loss gpu %.6f oracle %.6f" % (float(tot), float(loss)))
print("seq_vec (LSTM out)   value %.2e   grad %.2e   |grad| max %.3g" % (rel(g["seq"], o_seq), rel(g["seq"].grad, o_dseq), float(o_dseq.abs().max())))
print("r_e                  value %.2e   grad %.2e   |grad| max %.3g" % (rel(g["e"][:, :E], o_e), rel(g["e"].grad[:, :E], o_de), float(o_de.abs().max())))
print("r_a                  value %.2e   grad %.2e   |grad| max %.3g" % (rel(g["a"], o_a), rel(g["a"].grad, o_da), float(o_da.abs().max())))
d = (g["a"].grad.cpu().double() - o_da.double()).abs()
i = int(d.argmax()); r, c = divmod(i, o_da.shape[1])
print("worst r_a grad element: row %d col %d gpu %.6g oracle %.6g  p=%.8g" % (r, c, float(g["a"].grad[r, c]), float(o_da[r, c]), float(o_a[r, c])))
d = (g["e"].grad[:, :E].cpu().double() - o_de.double()).abs()
i = int(d.argmax()); r, c = divmod(i, E)
print("worst r_e grad element: row %d col %d gpu %.6g oracle %.6g  p=%.8g" % (r, c, float(g["e"].grad[r, c]), float(o_de[r, c]), float(o_e[r, c])))
for n, p in gm.named_parameters():
    if n.startswith("recipe_reasoner.") and p.grad is not None and P[n].grad is not None:
        print("   %-44s %.2e" % (n, rel(p.grad, P[n].grad)))
