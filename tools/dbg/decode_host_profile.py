"""host-side profile of Translator.translate_batch (graph replay) at config 5"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from svpc_amd import ops, make_batch, synthetic as syn
from svpc_amd.translator import Translator
args = bench.parse_args([])
dev = torch.device("cuda:0")
ops.set_precision("bf16x3")
cfg, model = bench.build(args, dev, model_type="vivt")
model.eval()
b = make_batch(cfg, n_videos=64, max_steps=args.clips, n_ingr=10, n_oov=0, seed=2019, full_clips=True)
for k, v in list(b.items()):
    if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
        b[k] = [t.to(dev) for t in v]
    elif isinstance(v, torch.Tensor):
        b[k] = v.to(dev)
tr = Translator(type("O", (), {"cuda": True})(), {"model_cfg": cfg, "model": model.state_dict()}, model=model, graph=True)
for _ in range(3):
    tr.translate_batch(syn.translate_inputs(b))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    tr.translate_batch(syn.translate_inputs(b))
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print("host %.2f ms/call, wall %.2f ms/call" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    tr.translate_batch(syn.translate_inputs(b))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
