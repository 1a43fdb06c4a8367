"""Clip-encoder hipGraphs per clip count (svpc_amd/clip_graphs.py) under batches whose structure changes every step (reference loop:
src/train.py:91-132 over recursive_caption_dataset.py:528-576): a step whose clip encoder is replayed ≡ the eager step."""
import pytest
import torch

from helpers import build_model
from svpc_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_replayed_clip_encoder_step_equals_eager_step(golden_dir, precision):
    from svpc_amd import keep_host_copy, ops
    from svpc_amd import clip_graphs
    from svpc_amd.graph import backward_all, ops_stream
    from svpc_amd.optim import FusedBertAdam
    z, cfg, _, model = build_model("c1", "vivt", golden_dir, DEV)          # D=128, L=2, F=3072; Lv=32; eval mode: no dropout
    structs = [dict(S=[16, 3, 7, 12, 5], E=[31, 1, 10, 17, 4], X=[2, 0, 1, 2, 0], seed=31),
               dict(S=[4, 9, 3], E=[2, 25, 8], X=[1, 2, 0], seed=32)]
    batches = []
    for st in structs:
        b_cpu = syn.make_batch(cfg, n_videos=len(st["S"]), max_steps=max(st["S"]), step_nums=st["S"], n_ingr=st["E"], n_oov=st["X"],
                               seed=st["seed"], full_clips=False)
        gn = torch.Generator().manual_seed(100 + st["seed"])
        noise = [-torch.empty(s_, cfg.max_t_len, cfg.vocab_size + x).exponential_(generator=gn).log().to(DEV) for s_, x in zip(st["S"], st["X"])]
        b = {kk: ([t.to(DEV) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else
                  (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for kk, v in b_cpu.items()}
        keep_host_copy(b["ingr_sep_masks"], b_cpu["ingr_sep_masks"])
        batches.append((syn.forward_args(b), noise))
    ops.set_precision(precision)
    try:
        opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, warmup=0.1, t_total=1000, weight_decay=0.0, grad_clip=1.0)

        def run(k):
            args, noise = batches[k % 2]
            model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
            model.gumbel_noise = noise
            opt.zero_grad()
            tot, probs, _, _ = model(*args)
            backward_all(model, tot)
            ops.join_side()
            torch.cuda.synchronize()
            opt.ensure_built()                       # (first call: the gradient arena; lr = 0, no optimizer step is taken)
            return float(tot.detach()), [p.detach().clone() for p in probs], opt.arena.flat.clone()
        with torch.cuda.stream(ops_stream()):
            run(0)
            clip_graphs.enable(model, False)
            eager = [run(k) for k in range(4)]
            cg, dg = clip_graphs.enable(model)
            graphed = [run(k) for k in range(4)]
        assert cg.stats["captures"] == 2 and cg.stats["hits"] == 2, cg.stats
        assert dg.stats["captures"] == 2 and dg.stats["hits"] == 2, dg.stats
        for k, (e, g) in enumerate(zip(eager, graphed)):
            # the replayed forward runs the same kernels on the same values (the frame rows gathered beforehand instead of inside the
            # first LayerNorm): loss and probabilities bit for bit; gradient tails that sum in arrival order to 1e-5
            assert e[0] == g[0], (k, e[0], g[0])
            assert all(torch.equal(a, c) for a, c in zip(e[1], g[1])), k
            assert float((e[2] - g[2]).abs().max()) <= 1e-5 * float(e[2].abs().max()), k
            assert float(g[2].abs().max()) > 0
    finally:
        clip_graphs.enable(model, False)
        ops.set_precision("fp32")


def test_eviction_recapture_and_training_mode_dropout(golden_dir):
    """more clip counts than cache entries (LRU eviction, recapture into the shared pool), in TRAINING mode with dropout on: every step
    finite, the captured encoder / decoder draw fresh masks per replay (two replays of one structure give different losses), and with the
    graphs switched off again the eager path still reproduces itself from the same seed state"""
    from svpc_amd import keep_host_copy, ops, clip_graphs
    from svpc_amd.graph import backward_all, ops_stream
    from svpc_amd.optim import FusedBertAdam
    z, cfg, _, model = build_model("c1", "vivt", golden_dir, DEV)
    model.train()
    # (clip counts 15, 25, 19, 40: four different buckets of 8 — the graphs are captured per clip count rounded up to a multiple of 8)
    structs = [dict(S=[5, 3, 7], E=[3, 1, 10], X=[0, 0, 1], seed=41), dict(S=[4, 9, 12], E=[2, 25, 8], X=[1, 2, 0], seed=42),
               dict(S=[16, 3], E=[31, 4], X=[2, 0], seed=43), dict(S=[6, 6, 6, 6, 9, 7], E=[5, 5, 5, 5, 3, 8], X=[0, 0, 0, 0, 1, 0], seed=44)]
    batches = []
    for st in structs:
        b_cpu = syn.make_batch(cfg, n_videos=len(st["S"]), max_steps=max(st["S"]), step_nums=st["S"], n_ingr=st["E"], n_oov=st["X"],
                               seed=st["seed"], full_clips=False)
        b = {kk: ([t.to(DEV) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else
                  (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for kk, v in b_cpu.items()}
        keep_host_copy(b["ingr_sep_masks"], b_cpu["ingr_sep_masks"])
        batches.append(syn.forward_args(b))
    ops.set_precision("bf16x3")
    try:
        opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, warmup=0.1, t_total=1000, weight_decay=0.0, grad_clip=1.0)   # (weights stay put)
        model.gumbel_noise = None

        def run(k):
            model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
            opt.zero_grad()
            tot = model(*batches[k % len(batches)])[0]
            backward_all(model, tot)
            opt.step()
            return float(tot.detach())
        with torch.cuda.stream(ops_stream()):
            run(0)
            cg, dg = clip_graphs.enable(model)
            cg.max_entries = dg.max_entries = 2
            losses = [run(k) for k in range(12)]          # 4 clip counts through 2 entries: every step after the first pass recaptures
            torch.cuda.synchronize()
        assert all(l == l and abs(l) < 1e9 for l in losses), "losses: %r" % (losses,)
        assert len(cg.entries) == 2 and len(dg.entries) == 2
        assert cg.stats["captures"] >= 8 and dg.stats["captures"] >= 8, (cg.stats, dg.stats)
        cg.max_entries = dg.max_entries = 8
        with torch.cuda.stream(ops_stream()):
            warm = [run(k) for k in range(4)]
            a = [run(0) for _ in range(3)]                  # same structure and weights, replayed: only the dropout / Gumbel seeds move on
            torch.cuda.synchronize()
        assert cg.stats["hits"] >= 3
        assert len({round(v, 3) for v in a}) == 3, a      # three different dropout draws
    finally:
        clip_graphs.enable(model, False)
        ops.set_precision("fp32")


def test_guards_plain_backward_second_forward_and_default_stream(golden_dir):
    """the three unguarded ways to misuse the replayed clip encoder fail loudly — (1) ``loss.backward()`` instead of
    ``graph.backward_all`` (the clip encoder's gradients would silently be missing) raises at the optimizer step, (2) a second forward
    with gradients before the owed backward replay raises, (3) on the legacy default stream (where a capture cannot run) the model
    takes the eager path instead of raising mid-step."""
    from svpc_amd import keep_host_copy, ops, clip_graphs
    from svpc_amd.graph import backward_all, ops_stream
    from svpc_amd.optim import FusedBertAdam
    z, cfg, _, model = build_model("c1", "vivt", golden_dir, DEV)
    b_cpu = syn.make_batch(cfg, n_videos=3, max_steps=7, step_nums=[5, 3, 7], n_ingr=[3, 1, 10], n_oov=[0, 0, 1], seed=51, full_clips=False)
    b = {kk: ([t.to(DEV) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else
              (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for kk, v in b_cpu.items()}
    keep_host_copy(b["ingr_sep_masks"], b_cpu["ingr_sep_masks"])
    args = syn.forward_args(b)
    model.gumbel_noise = None
    try:
        opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, weight_decay=0.0, grad_clip=1.0)
        with torch.cuda.stream(ops_stream()):
            opt.zero_grad()
            backward_all(model, model(*args)[0])
            opt.step()
            cg, dg = clip_graphs.enable(model)
            # (1) plain backward: the optimizer refuses to step on gradients that lack the clip encoder's part
            opt.zero_grad()
            model(*args)[0].backward()
            with pytest.raises(RuntimeError, match="backward_all"):
                opt.step()
            assert cg.pending is None
            # (2) two forwards, no backward in between
            opt.zero_grad()
            model(*args)
            with pytest.raises(RuntimeError, match="second forward"):
                model(*args)
            # … and the loop recovers: a correct step afterwards works
            opt.zero_grad()
            backward_all(model, model(*args)[0])
            opt.step()
            hits = cg.stats["hits"]
            torch.cuda.synchronize()
        # (3) default stream: eager path, no capture attempted, gradients complete
        torch.cuda.synchronize()
        opt.zero_grad()
        backward_all(model, model(*args)[0])
        ops.join_side()
        torch.cuda.synchronize()
        assert cg.stats["hits"] == hits and model.split_boundary is None
        assert float(model.video_embeddings.video_embeddings[2].weight.grad.abs().max()) > 0
    finally:
        clip_graphs.enable(model, False)


def test_bucketed_clip_counts_and_the_dp_reducer(golden_dir):
    """(a) two batches whose clip counts fall into the SAME bucket of 8 share one pair of graphs (the padding clips
    contribute exact zeros); (b) the graphs stay in use with the data-parallel reducer attached — one rank over gloo, SUM over one rank is
    the identity — through ``backward_all(model, loss, exchange=reducer)``: text-side buckets are started between the eager backward and
    the clip encoder's replay, the rest by ``finish()``; the exchanged gradients equal the plain eager step's."""
    import os
    import socket
    import torch.distributed as dist
    from svpc_amd import keep_host_copy, ops, clip_graphs
    from svpc_amd.graph import backward_all, ops_stream
    from svpc_amd.optim import FusedBertAdam, GradReducer
    z, cfg, _, model = build_model("c1", "vivt", golden_dir, DEV)
    structs = [dict(S=[5, 3, 7], E=[3, 1, 10], X=[0, 0, 1], seed=61), dict(S=[4, 6, 3], E=[2, 25, 8], X=[1, 2, 0], seed=62)]      # T = 15, 13
    batches = []
    for st in structs:
        b_cpu = syn.make_batch(cfg, n_videos=len(st["S"]), max_steps=max(st["S"]), step_nums=st["S"], n_ingr=st["E"], n_oov=st["X"],
                               seed=st["seed"], full_clips=False)
        gn = torch.Generator().manual_seed(100 + st["seed"])
        noise = [-torch.empty(s_, cfg.max_t_len, cfg.vocab_size + x).exponential_(generator=gn).log().to(DEV) for s_, x in zip(st["S"], st["X"])]
        b = {kk: ([t.to(DEV) if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, list) else
                  (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for kk, v in b_cpu.items()}
        keep_host_copy(b["ingr_sep_masks"], b_cpu["ingr_sep_masks"])
        batches.append((syn.forward_args(b), noise))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    ops.set_precision("bf16x3")
    red = None
    try:
        opt = FusedBertAdam(list(model.named_parameters()), lr=0.0, warmup=0.1, t_total=1000, weight_decay=0.0, grad_clip=1.0)

        def run(k, exchange=None):
            args, noise = batches[k % 2]
            model._plans.clear(); model._ptr_plans.clear(); model._span_cache.clear()
            model.gumbel_noise = noise
            opt.zero_grad()
            tot = model(*args)[0]
            backward_all(model, tot, exchange=exchange)
            if exchange is not None:
                exchange.finish()
            ops.join_side()
            torch.cuda.synchronize()
            arena = opt.ensure_built()
            return float(tot.detach()), arena.flat.clone()
        with torch.cuda.stream(ops_stream()):
            run(0)
            eager = [run(k) for k in range(2)]
            cg, dg = clip_graphs.enable(model)
            graphed = [run(k) for k in range(4)]
            assert cg.stats["captures"] == 1 and cg.stats["hits"] == 3, cg.stats          # 15 and 13 clips: one bucket (16)
            assert dg.stats["captures"] == 1 and dg.stats["hits"] == 3, dg.stats
            red = GradReducer(opt.arena, bucket_bytes=256 << 10, overlap=True, force=True)
            assert red.active and red.overlap
            with_dp = [run(k, exchange=red) for k in range(4)]
            assert cg.stats["hits"] == 7 and cg.stats["bypassed"] == 0, cg.stats          # the graphs stayed in use under the reducer
        for k in range(4):
            e = eager[k % 2]
            for tag, g in (("graphs", graphed[k]), ("graphs + reducer", with_dp[k])):
                assert abs(e[0] - g[0]) <= 1e-6 * abs(e[0]), (tag, k, e[0], g[0])
                assert float((e[1] - g[1]).abs().max()) <= 2e-5 * float(e[1].abs().max()), (tag, k)
    finally:
        clip_graphs.enable(model, False)
        ops.set_precision("fp32")
        if red is not None:
            red.close()
        if created:
            dist.destroy_process_group()
