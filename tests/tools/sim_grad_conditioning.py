"""Why the fp32 mode leaves a 1e-3 residue on the simulators' W1 / W2 / action-selector gradients with the test-sensitive weights
(tests/test_headline_parity.py: recipe_reasoner.W2.weight norm_rel 1.5e-3 at a loss error of 1e-7, 0 Gumbel flips): the CPU oracle ITSELF, run in
fp32 and in fp64 on the same inputs, differs by that much on these tensors — nn.BCELoss(sum) on saturated sigmoid outputs
(model.py:871, :810: 23 % of the entity probabilities are > 0.999 or < 0.001 with n/sqrt(fan_in) weights) differentiates through
1/(1-e)·e(1-e) evaluated from the ROUNDED probability, whose last ulp (different exp implementations on CPU and GPU) moves a
saturated entry's gradient by up to ulp/(1-e).  With the bench weights (N(0, 0.02)) nothing saturates and the residue is 1e-6.
usage: python tests/tools/sim_grad_conditioning.py   (CPU only)"""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from oracle import svpc_oracle as orc
from svpc_amd import synthetic as syn
args = bench.parse_args([])
cfg, model = bench.build(args, "cpu", model_type="vivt")
drawn = syn.draw_parameters(list(model.named_parameters()), seed=7)
g = torch.Generator().manual_seed(5)
S, E, D = 12, 10, cfg.hidden_size
step = torch.randn(S, D, generator=g)          # LayerNorm-scale step vectors
ent = torch.randn(E, D, generator=g)
we, wa = torch.randn(S, E, generator=g), torch.randn(S, cfg.action_vocab_size, generator=g)
def run(dt):
    torch.set_default_dtype(dt)
    P = {k: v.detach().clone().to(dt) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    for n in drawn: P[n] = drawn[n].clone().to(dt)
    names = [n for n in P if n.startswith("recipe_reasoner.") and not n.endswith("action_embeddings")]
    for n in names: P[n].requires_grad_(True)
    e, a, bar_e, all_e, bar_f = orc.simulator(P, "recipe_reasoner", step.to(dt), ent.to(dt))
    # the model's own use of e and a: BCE(sum) against an alignment and ASL on the action probabilities
    y = (we > 1.0).to(dt)
    loss = torch.nn.functional.binary_cross_entropy(e, y, reduction="sum") + (a * wa.to(dt)).sum()      # the model's entity loss: nn.BCELoss(sum), model.py:871
    loss.backward()
    sat = float(((e > 0.999) | (e < 0.001)).double().mean())
    return float(loss), {n: P[n].grad.double() for n in names}, sat, e.detach().double()
l32, g32, sat, e32 = run(torch.float32)
l64, g64, _, e64 = run(torch.float64)
print("saturated entity probabilities (>0.999 or <0.001): %.2f" % sat, " max |e32-e64| %.2e" % float((e32 - e64).abs().max()))
for n in g64:
    a, b = g32[n].reshape(-1), g64[n].reshape(-1)
    print("%-40s norm_rel %.2e  diff_rel %.2e" % (n, abs(float(a.norm()) - float(b.norm())) / float(b.norm()), float((a - b).norm() / b.norm())))
