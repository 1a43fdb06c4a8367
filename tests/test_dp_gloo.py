"""Data-parallel path on CPU (gloo, world_size 2): the bucketed SUM all-reduce over the gradient arena reproduces the
full-batch gradient — valid because the reference's loss is a SUM over videos (model.py:1110-1115, :1188), so two ranks
with one video each ≡ one process with both videos (the tiny golden fixture).  HIP kernels are emulated (tests/emul_ops.py);
the arena / bucket / hook logic under test is exactly what runs over RCCL on the MI355X node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, golden_dir, overlap, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import emul_ops
    from helpers import build_model
    from svpc_amd import model as M
    from svpc_amd import synthetic as syn
    from svpc_amd.optim import GradArena, GradReducer
    M.ops = emul_ops
    z, cfg, batch, model = build_model("tiny", "vivt", golden_dir)
    # shard: rank r keeps video r of the 2-video fixture (video 1 has only 2 steps → drop its padding step)
    steps = batch["batch_step_num"][rank]
    sl = lambda lst: [t[rank:rank + 1] for t in lst[:steps]]
    shard = dict(batch)
    for k in ("input_ids_list", "video_features_list", "input_masks_list", "token_type_ids_list", "input_labels_list"):
        shard[k] = sl(batch[k])
    for k in ("ingr_input_ids", "ingr_masks", "ingr_sep_masks"):
        shard[k] = batch[k][rank:rank + 1]
    for k in ("batch_step_num", "ingr_id_dict", "extra_zeros", "alignments", "actions"):
        shard[k] = batch[k][rank:rank + 1]
    model.gumbel_noise = [model.gumbel_noise[rank]]
    if overlap == "pack":        # the sentence side over the valid tokens only (svpc_amd.model.TextPack) under the reducer's hooks
        from svpc_amd import keep_host_copy
        for m in shard["input_masks_list"]:
            keep_host_copy(m, m)
        model.pack_text_rows = True
    named = [(n, p) for n, p in model.named_parameters() if "memory_intermediate" not in n]
    if overlap == "split":
        # two-phase backward (svpc_amd/graph.py): cut at the [CLS] rows, exchange the text-side buckets before the clip encoder's
        # backward has run, the rest afterwards
        from svpc_amd.graph import backward_all
        for _, p in named:
            p.grad = torch.zeros_like(p)
        arena = GradArena(named)
        red = GradReducer(arena, bucket_bytes=16 << 10, overlap=False)
        model.split_backward = True
        loss = model(*syn.forward_args(shard))[0]
        loss.backward()
        out, cut = model.split_boundary
        assert all(p.grad.abs().max() == 0 for n, p in named if n.startswith(GradReducer.CLIP_SIDE))      # phase 1 left them untouched
        red.start_early()
        n_early = sum(red.launched)
        assert 0 < n_early < len(red.buckets)
        out.backward(cut.grad)
        red.finish()
    elif overlap is True or overlap == "pack":      # ("bf16" takes the branch below: the optional wire format)
        for _, p in named:
            p.grad = torch.zeros_like(p)
        arena = GradArena(named)
        red = GradReducer(arena, bucket_bytes=16 << 10, overlap=True)
        loss = model(*syn.forward_args(shard))[0]
        loss.backward()
        red.finish()
    else:
        loss = model(*syn.forward_args(shard))[0]
        loss.backward()
        named = [(n, p) for n, p in named if p.grad is not None]
        arena = GradArena(named)
        # "bf16": the optional bf16 wire format (cast → all-reduce → back into the fp32 arena) with the per-bucket timeline on
        red = GradReducer(arena, bucket_bytes=16 << 10, overlap=False, wire_dtype="bf16" if overlap == "bf16" else "fp32",
                          timeline=overlap == "bf16")
        red.mark_step_start()
        red.finish()
        if overlap == "bf16":
            tl_ = red.timeline_ms()
            assert len(tl_) == len(red.buckets) and all(t["complete_ms"] >= t["issued_ms"] >= 0.0 for t in tl_)
            assert sum(t["bytes"] for t in tl_) == red.bytes_per_step() == sum((e - s_) * 2 for s_, e, _ in red.buckets)
    tl = torch.tensor([loss.item()], dtype=torch.float64)
    dist.all_reduce(tl)
    if rank == 0:
        out_q.put((tl.item(), {n: p.grad.clone().numpy() for n, p in arena_named(arena)}, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def arena_named(arena):
    return list(zip(arena.names, arena.params))


@pytest.mark.parametrize("overlap", [False, True, "split", "bf16", "pack"])
def test_two_rank_sum_allreduce_equals_full_batch_gradient(golden_dir, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, grads, n_buckets = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    z = np.load(os.path.join(golden_dir, "tiny_vivt.npz"))
    assert n_buckets > 3
    assert abs(total - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    n = 0
    for name, g in grads.items():
        k = "grad/" + name
        if k not in z.files:
            assert np.abs(g).max() == 0.0, name
            continue
        ref = z[k]
        # fp32 exchange: the full-batch gradient to summation order.  bf16 wire format: each rank's contribution rounded once to 8
        # significant bits (2⁻⁹ relative) plus one bf16 add — bounded at 1 % of the tensor's largest element
        tol = 1e-2 if overlap == "bf16" else 3e-4
        assert np.abs(g - ref).max() <= tol * max(1e-6, np.abs(ref).max()) + 1e-6, name
        n += 1
    assert n > 20


def test_post_accumulate_hook_fires_without_a_gradient_tensor():
    """GradReducer's readiness signal: a leaf's post-accumulate hook runs once per backward, after all of its consumers, even when
    their backward returned no gradient tensor for it (the in-place arena write of the HIP ops) — torch behaviour the overlap
    logic relies on (svpc_amd/optim.py)."""
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            w.grad += x.t() @ g          # "kernel writes in place"
            return g @ w.t(), None
    w = torch.nn.Parameter(torch.randn(3, 3))
    w.grad = torch.zeros(3, 3)
    x = torch.randn(2, 3, requires_grad=True)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(float(p.grad.abs().sum())))
    (F.apply(x, w).sum() + F.apply(2 * x, w).sum()).backward()
    assert len(fired) == 1 and fired[0] > 0          # once, after BOTH uses have written


def test_reducer_refuses_a_view_written_again_after_launch():
    from svpc_amd.optim import GradArena, GradReducer
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        params = [("w%d" % i, torch.nn.Parameter(torch.zeros(64, 64))) for i in range(4)]
        for _, p in params:
            p.grad = torch.zeros_like(p)
        arena = GradArena(params)
        red = GradReducer(arena, bucket_bytes=8 << 10, overlap=True, force=True)
        bi_last = len(red.buckets) - 1
        for bi, (_, _, members) in enumerate(red.buckets):
            for i in members:
                red._done(i, bi)
        assert all(red.launched)
        with pytest.raises(RuntimeError, match="reported ready again"):
            red._done(red.buckets[bi_last][2][0], bi_last)
        red.finish()
        red.close()
    finally:
        dist.destroy_process_group()
