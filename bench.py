#!/usr/bin/env python
"""Headline benchmark: train steps/sec of the state-aware recurrent transformer (MODEL_TYPE=vivt, batch 16 clip sequences
× 12 clips × 100 frames × 3072 features, D=768, 6+6+6 layers) on N MI355X, one process per GPU, RCCL gradient all-reduce.

A step = {zero_grad, forward, backward, gradient all-reduce (N>1), global clip 1.0, BertAdam} on one synthetic batch that
is already resident in HBM (reference: src/train.py:125-147 without EMA/logging).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (metric/value/… + "roofline" for the dominant kernel + "cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from svpc_amd import StateAwareRecursiveTransformer, make_batch, make_config  # noqa: E402
from svpc_amd import ops, synthetic as syn  # noqa: E402
from svpc_amd.graph import backward_all
from svpc_amd.optim import FusedBertAdam, GradReducer  # noqa: E402

PMC_TRAFFIC_BYTES = 119381333   # (2·FETCH_SIZE + WRITE_SIZE)·1024, means per launch of the dominant kernel: profiles/r01_i_pmc_bench_dominant_gemm.csv
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md dense peaks (f32-in MFMA; bf16 MFMA)


def build(args, device):
    cfg = make_config(model_type=args.model_type, hidden_size=args.hidden, num_hidden_layers=args.layers,
                      num_attention_heads=args.heads)
    torch.manual_seed(2019)
    model = StateAwareRecursiveTransformer(cfg)
    g = torch.Generator().manual_seed(2019)
    glove = 0.4 * torch.randn(cfg.vocab_size, cfg.word_vec_size, generator=g)
    verb = 0.4 * torch.randn(cfg.action_vocab_size, cfg.word_vec_size, generator=g)
    model.ingredient_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
    model.text_embeddings.set_pretrained_embedding(glove.clone(), freeze=False)
    if args.model_type in ("vivt", "viv"):
        model.reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
    if args.model_type == "vivt":
        model.recipe_reasoner.set_pretrained_embedding(verb.clone(), freeze=False)
    return cfg, model.to(device)


def device_batch(cfg, args, device, seed):
    b = make_batch(cfg, n_videos=args.batch, max_steps=args.clips, n_ingr=10, n_oov=0, seed=seed, full_clips=True)
    feats = torch.stack(b["video_features_list"]).to(device)           # one (S, N, L, F) buffer: consumed in place
    b["video_features_list"] = [feats[s] for s in range(feats.shape[0])]
    for k, v in list(b.items()):
        if k == "video_features_list":
            continue
        if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            b[k] = [t.to(device) for t in v]
        elif isinstance(v, torch.Tensor):
            b[k] = v.to(device)
    return b


def cpu_baseline(cfg, model, args):
    """The CPU oracle (a port of the reference path, pinned to it by tests/golden) timed on this host's cores on a
    bounded sample of the same workload; reported next to the GPU number, never the target."""
    from oracle import svpc_oracle as orc
    n_vid = args.cpu_videos
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    b = make_batch(cfg, n_videos=n_vid, max_steps=args.clips, n_ingr=10, n_oov=0, seed=7, full_clips=True)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, v in P.items() if v.dtype.is_floating_point and not k.endswith(".pe")]
    state = {}
    wd = {n: (0.0 if any(t in n for t in ("bias", "LayerNorm.bias", "LayerNorm.weight")) else 0.01) for n in names}

    def one_step():
        for n in names:
            P[n].requires_grad_(True)
            P[n].grad = None
        total, _, _, _ = orc.forward(P, cfg, *syn.forward_args(b), training=True)
        total.backward()
        grads = {n: P[n].grad for n in names if P[n].grad is not None}
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
        coef = float(min(1.0, 1.0 / (gn + 1e-6)))
        with torch.no_grad():
            for n in names:
                P[n].requires_grad_(False)
            orc.bert_adam_step({n: P[n] for n in grads}, {n: g * coef for n, g in grads.items()}, state, 1e-4, wd=wd)
    one_step()
    t0 = time.time()
    k = 0
    while k < args.cpu_steps:
        one_step()
        k += 1
    dt = (time.time() - t0) / k
    steps_per_s = (n_vid / float(args.batch)) / dt
    return {"value": steps_per_s, "unit": "steps/s (16-video steps)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of %d videos per step (S=%d, L=%d, same model), %d timed steps of fwd+bwd+clip+BertAdam on torch-CPU fp32, "
                      "%.2f s/step, scaled by %d/%d" % (n_vid, args.batch, args.clips, args.layers, k, dt, n_vid, args.batch)}


def decode_bench(cfg, model, args, device, world, rank, dist):
    """captions/s of Translator.translate_batch (greedy) on synthetic clips, videos sharded over ranks (replicas only)."""
    from svpc_amd.translator import Translator
    n_vid = args.decode_videos
    b = make_batch(cfg, n_videos=n_vid, max_steps=args.clips, n_ingr=10, n_oov=0, seed=2019 + rank, full_clips=True)
    for k, v in list(b.items()):
        if isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            b[k] = [t.to(device) for t in v]
        elif isinstance(v, torch.Tensor):
            b[k] = v.to(device)
    tr = Translator(type("O", (), {"cuda": True})(), {"model_cfg": cfg, "model": model.state_dict()}, model=model,
                    incremental=not args.decode_full, graph=not args.decode_eager)
    for _ in range(max(1, args.warmup)):
        tr.translate_batch(syn.translate_inputs(b))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, _ = tr.translate_batch(syn.translate_inputs(b))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        caps = world * args.steps * n_vid * args.clips
        print(json.dumps({"metric": "greedy-decode captions/sec (vivt, 64 videos)", "value": caps / elapsed, "unit": "captions/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                          "config": {"workload": "MODEL_TYPE=%s translate_batch greedy: %d videos/GPU x %d clips, Lt=%d, L=%d; %s, batched "
                                                 "over videos, on-device pick"
                                                 % (args.model_type, n_vid, args.clips, cfg.max_t_len, cfg.num_hidden_layers,
                                                    "full decoder re-run per position (reference loop shape)" if args.decode_full
                                                    else "KV-cached incremental decoder (one new token per sentence and iteration)"),
                                     "launch": "eager" if args.decode_eager else "hipGraph replay per batch structure (inputs copied into the captured buffers)"}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model-type", default="vivt")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--clips", type=int, default=12)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="arithmetic type of the GEMM operands (accumulation and storage are fp32 either way)")
    ap.add_argument("--decode", action="store_true", help="secondary metric: greedy-decode captions/s (BASELINE config 5: 64 videos)")
    ap.add_argument("--decode-videos", type=int, default=64)
    ap.add_argument("--decode-full", action="store_true", help="decode with the reference-shaped loop (all Lt positions every iteration)")
    ap.add_argument("--decode-eager", action="store_true", help="launch the decode kernels eagerly instead of replaying the batch structure's hipGraph")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-videos", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=8)      # ≈ 12 s of CPU work (1 warm-up + 8 timed 2-video steps)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        dev_index = local_rank % max(1, n_dev)
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    device = torch.device("cuda", dev_index)

    ops.set_precision(args.precision)
    cfg, model = build(args, device)
    if args.decode:
        return decode_bench(cfg, model, args, device, world, rank, dist)
    model.train()
    batch = device_batch(cfg, args, device, seed=2019 + rank)
    fargs = syn.forward_args(batch)
    opt = FusedBertAdam(list(model.named_parameters()), lr=1e-4, warmup=0.1, t_total=100000, grad_clip=1.0, ema_decay=-1.0)
    reducer = None

    def step():
        nonlocal reducer
        opt.zero_grad()
        loss = model(*fargs)[0]
        backward_all(model, loss)
        arena = opt.ensure_built()
        if world > 1:
            if reducer is None:
                # graph mode: the exchange runs between two captured graphs (no hooks); eager mode: overlapped with backward
                reducer = GradReducer(arena, overlap=args.no_graph)
                reducer.reset()
                for bi in range(len(reducer.buckets)):   # first step: hooks were not installed during this backward
                    reducer.launched[bi] = False
            reducer.finish()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 2)):
        loss = step()
    torch.cuda.synchronize()

    # ---- capture the step in hipGraphs: ≈1,100 launches per step would otherwise make it host-bound.  One GPU: one graph
    # (zero_grad → forward → backward → clip+BertAdam).  N GPUs: {zero_grad, forward, backward} and {clip+BertAdam} with the
    # bucketed RCCL all-reduce issued eagerly in between.  Falls back to eager (overlapped exchange) on failure.
    graph = None
    if not args.no_graph:
        try:
            from svpc_amd.graph import GraphedTrainStep
            loss = None          # drop the last eager autograd graph before capture
            graph = GraphedTrainStep(model, opt, fargs, warmup=2, exchange=(reducer if world > 1 else None))
            eager_step = step
            step = graph
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print("[bench] hipGraph capture failed (%s: %s) - running eagerly" % (type(e).__name__, e), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    # host-side enqueue cost of one step (GPU idle at start, no sync at the end): tells whether the step is launch-bound
    th = time.perf_counter()
    loss = step()
    host_enqueue_ms = 1000.0 * (time.perf_counter() - th)
    torch.cuda.synchronize()
    # roofline leg: HIP events around every launch of the dominant kernel symbol.  Eager mode: inside the timed region.  Graph
    # mode: events cannot be recorded inside a replay, so the same kernels are bracketed in instrumented eager steps right after it.
    # Dominant kernel = the forward GEMM of the clip-encoder activation stream: ONE kernel symbol (gemm_glds_pp_kernel<true,true,__bf16>:
    # bf16 activations × bf16 weight shadow → bf16, both operands direct-to-LDS, 256×256 ping-pong tiles) covering the Q/K/V,
    # attention-output, FFN-in/out and video-embedding projections of the clip encoder (M = 19,200 rows; 22 launches per step) —
    # every launch of that symbol is bracketed, so the average can be checked against rocprofv3's per-kernel average.
    rows_enc = args.batch * args.clips * cfg.max_v_len
    bf16_stream = args.precision == "bf16" and ops.bf16_stream_ok(rows_enc, cfg.hidden_size, cfg.video_feature_size)
    glds = bf16_stream and ops.USE_GLDS      # weights come from the optimizer's bf16 shadow → direct-to-LDS kernel
    want_dt = ((1, 1, 1) if glds else (1, 0, 1)) if bf16_stream else (0, 0, 0)

    def dom_select(d):
        M_, N_, K_, akc, bkc, adt, bdt, cdt = d
        return akc == 1 and bkc == 1 and (adt, bdt, cdt) == want_dt and M_ == rows_enc
    if graph is None:
        ops.GEMM_TIMER = ops.KernelTimer(select=dom_select)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if graph is not None:
        ops.GEMM_TIMER = ops.KernelTimer(select=dom_select)
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
    timer, ops.GEMM_TIMER = ops.GEMM_TIMER, None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gsum = timer.summary()
    final_loss = float(loss.item())

    if rank == 0:
        ms = 1000.0 * elapsed / args.steps
        achieved = gsum["work"] / (gsum["ms"] * 1e-3) / 1e12 if gsum["ms"] > 0 else 0.0
        D_, F_, L_ = cfg.hidden_size, cfg.video_feature_size, cfg.num_hidden_layers
        n_l = max(1, gsum["launches"])
        alg_bytes, alg_flop = gsum["bytes"] / n_l, gsum["work"] / n_l
        # mean HBM traffic per launch of this symbol from rocprofv3 PMC passes over this very command (profiles/r01_i_pmc_*.csv):
        # FETCH_SIZE × 2 (gfx950 reports half of a 16-B/lane stream) + WRITE_SIZE, in KB.  Only valid for the default workload.
        default_cfg = (args.batch, args.clips, L_, D_, F_, args.model_type) == (16, 12, 6, 768, 3072, "vivt") and bf16_stream
        traffic = PMC_TRAFFIC_BYTES if (default_cfg and glds) else None
        out = {
            "metric": "train steps/sec (vivt, batch=16, clip_seq=12)", "value": world * args.steps / elapsed,
            "unit": "steps/s (one step = 16 clip-sequences per GPU; whole-job aggregate)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "MODEL_TYPE=%s train step: N=%d videos/GPU x S=%d clips x Lv=%d frames x F=%d, Lt=%d, D=%d, H=%d, "
                                   "L=%d (enc+step-enc+dec), V=%d, A=%d, E=10; dropout .1/.1/.4; fwd+bwd+allreduce+clip+BertAdam"
                                   % (args.model_type, args.batch, args.clips, cfg.max_v_len, cfg.video_feature_size, cfg.max_t_len,
                                      cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers, cfg.vocab_size, cfg.action_vocab_size),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "host_enqueue_ms_per_step": host_enqueue_ms,
                       "launch": ("hipGraph replay" if world == 1 else "hipGraph replay (fwd + text-side bwd | clip-encoder bwd beside the text-side all-reduce | remaining all-reduce | optimizer)")
                                 if graph is not None else "eager"},
            "roofline": {"bound": "mfma", "kernel": "%s — every forward projection of the clip-encoder activation stream (M=%d rows: "
                                   "Q/K/V, attention-out, FFN, video embedding)"
                                   % ("gemm_glds_pp_kernel<true,true,__bf16> (bf16·bf16→bf16, direct-to-LDS, 256x256 ping-pong tiles)" if glds else
                                      "gemm_bf16_kernel<128,128,NT,interior,8 waves,%s>" % ("bf16·f32→bf16" if bf16_stream else "f32"), rows_enc),
                         "achieved": achieved, "peak": MFMA_PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS[args.precision],
                         "traffic": traffic, "algorithmic_flop_per_launch": alg_flop, "algorithmic_bytes_per_launch": alg_bytes,
                         "launches": gsum["launches"], "avg_launch_ms": gsum["ms"] / max(1, gsum["launches"]),
                         "event_pair_overhead_ms": gsum["event_overhead_ms"],
                         "measured": "HIP events on the launch stream (net of the calibrated empty event-pair time), " + ("3 instrumented eager steps after the timed graph replays"
                                                                            if graph is not None else "inside the timed region")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, args)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
