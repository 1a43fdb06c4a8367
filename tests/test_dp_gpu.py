"""Data-parallel gradient exchange through the REAL kernels' readiness path: two gloo ranks that share the one MI355X of a
test box, each with one video of the config-1 fixture, bf16 mode (packed Q/K/V views, the decoder's stacked K/V projection, deferred
grouped weight gradients, residual-gradient hand-over — everything that reports arena writes through ``ops._ready``), eager overlap
mode (buckets released by the post-accumulate hooks and by pointer reports while backward is still running).  The SUM of the two
ranks' gradients must equal the full-batch gradient one process computes on both videos (the reference's loss is a sum over videos:
src/rtransformer/model.py:1110-1115, :1188; exchange before the clip: src/train.py:140-143).
(The reducer's refusal of a view written again after its bucket's launch is a CPU test: tests/test_dp_gloo.py.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(batch, rank):
    steps = batch["batch_step_num"][rank]
    sl = lambda lst: [t[rank:rank + 1] for t in lst[:steps]]
    shard = dict(batch)
    for k in ("input_ids_list", "video_features_list", "input_masks_list", "token_type_ids_list", "input_labels_list"):
        shard[k] = sl(batch[k])
    for k in ("ingr_input_ids", "ingr_masks", "ingr_sep_masks"):
        shard[k] = batch[k][rank:rank + 1]
    for k in ("batch_step_num", "ingr_id_dict", "extra_zeros", "alignments", "actions"):
        shard[k] = batch[k][rank:rank + 1]
    return shard


def _grads_after_two_backwards(model, fargs, reducer_factory):
    """first backward builds arena + weight store (direct writes from then on); the measured one is the second"""
    from svpc_amd import ops
    from svpc_amd.optim import FusedBertAdam
    opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, grad_clip=-1.0, max_grad_norm=-1.0)
    model(*fargs)[0].backward()
    arena = opt.ensure_built()
    red = reducer_factory(arena)
    opt.zero_grad()
    loss = model(*fargs)[0]
    loss.backward()
    if red is not None:
        red.finish()
    ops.join_side()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().float().cpu().numpy().copy() for n, p in zip(arena.names, arena.params)}, red


def _worker(rank, world, port, golden_dir, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import build_model
    from svpc_amd import ops, synthetic as syn
    from svpc_amd.optim import GradReducer
    ops.set_precision("bf16")
    z, cfg, batch, model = build_model("c1", "vivt", golden_dir, "cuda:0")
    shard = _shard(batch, rank)
    model.gumbel_noise = [model.gumbel_noise[rank]]
    loss, grads, red = _grads_after_two_backwards(model, syn.forward_args(shard),
                                                  lambda arena: GradReducer(arena, bucket_bytes=256 << 10, overlap=True))
    tl = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(tl)
    if rank == 0:
        out_q.put((tl.item(), grads, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(280)
def test_two_ranks_overlapped_exchange_through_the_real_ready_path(golden_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_model
    from svpc_amd import ops, synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, grads, n_buckets = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n_buckets > 3
    # the full batch in ONE process, same arithmetic mode, same direct-write path
    ops.set_precision("bf16")
    try:
        z, cfg, batch, model = build_model("c1", "vivt", golden_dir, "cuda:0")
        loss, full, _ = _grads_after_two_backwards(model, syn.forward_args(batch), lambda arena: None)
    finally:
        ops.set_precision("fp32")
    assert abs(total - loss) <= 1e-5 * abs(loss), (total, loss)
    n = 0
    for name, g in full.items():
        scale = max(1e-6, float(np.abs(g).max()))
        # per-video activations are row-for-row identical in both runs; only the fp32 summation order of the weight gradients differs
        assert float(np.abs(grads[name] - g).max()) <= 2e-3 * scale + 1e-6, (name, float(np.abs(grads[name] - g).max()), scale)
        n += 1
    assert n > 100
    packed = [k for k in full if ".query.weight" in k or "dec_enc_attention.key.weight" in k]
    assert packed and all(float(np.abs(full[k]).max()) > 0 for k in packed if "encoder.layer.0" in k)
