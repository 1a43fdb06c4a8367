"""hipGraphs of the clip encoder for batches whose structure changes every step.

A whole-step graph (svpc_amd/graph.py) needs a recurring batch structure; the reference's loader (src/rtransformer/
recursive_caption_dataset.py:528-576, train.py:91-132) yields a new one per step — S_b clips, E_b ingredients, X_b out-of-vocabulary
words per video — and an eager step is bound by the host (≈14 ms of Python / launch work for ≈12 ms of kernels).  But the clip encoder
(video embedding + L encoder layers over all valid clips at once, src/rtransformer/model.py:1038-1042 — a quarter of the step's launches
and most of its kernel time) depends on the structure only through ONE number, the clip count T = Σ S_b: every clip is Lv frame rows, the
attention segmentation is uniform, which (step, video) slot a clip came from is data (an index table).  So its forward and its backward
are captured once per T — at most 16·N − 3·N + 1 values for N videos of 3..16 clips, a few dozen in practice — and replayed; the rest of
the step (step encoder, simulators, decoder, pointer, losses: ragged in S_b, E_b, X_b) stays eager.

Mechanics (as vLLM-style per-batch-size graphs): all graphs share one memory pool (only one of them runs per step, so the activations of
T = 150 and T = 170 occupy the same memory) and shared capacity buffers for the inputs (the valid clips' feature rows gathered compactly,
their token-type ids and key masks).  The cut between the graphs and the eager autograd graph is the (T, D) tensor of [CLS] rows — the
only path from the loss into the clip encoder, the same cut ``model.split_backward`` uses for the data-parallel overlap: ``run`` returns a
fresh leaf holding the replayed [CLS] rows and leaves ``model.split_boundary = (entry, leaf)``; after ``loss.backward()`` the caller runs
``entry.backward(leaf.grad)`` — ``svpc_amd.graph.backward_all`` does — which replays the captured backward (weight / bias / gain gradients
accumulate straight into the optimizer's arena, as in eager mode).  Dropout seeds live in HBM and are bumped by a kernel, so a replay draws
fresh masks; the per-op site numbers are those of the capture, and the Python site counter is advanced past them on replay.
"""
from __future__ import annotations

import collections

import torch

from . import ops
from .graph import capturing


_OWNERS = []        # every ClipEncoderGraphs alive in this process (weak): the optimizer asks them whether a backward is still owed


def check_consumed(who="optimizer step"):
    """Raise if a clip-encoder forward replay is still waiting for its backward replay: the training loop called ``loss.backward()``
    instead of ``svpc_amd.graph.backward_all`` (the clip encoder's gradients of this step would silently be missing)."""
    for ref in list(_OWNERS):
        cg = ref()
        if cg is None:
            _OWNERS.remove(ref)
        elif cg.pending is not None:
            cg.pending = None
            raise RuntimeError("clip_graphs: %s with the clip encoder's backward replay still pending — the step's backward must be "
                               "svpc_amd.graph.backward_all(model, loss), not loss.backward() (the clip-encoder gradients of this step "
                               "were never computed)" % who)


class _Entry:
    __slots__ = ("T", "g_fwd", "g_bwd", "feats", "ids", "mask", "seq", "cls_rows", "seq_cls", "cls", "gout", "n_sites", "replays", "owner", "t_valid")

    def backward(self, grad):
        """the second phase of the step's backward (see ``svpc_amd.graph.backward_all``)"""
        self.owner.pending = None
        if grad is None:
            return
        n = self.t_valid
        self.gout[:n].copy_(grad)
        if n < self.T:
            self.gout[n:].zero_()          # the padding clips of the bucket: zero output gradient → exact zeros in every parameter gradient
        self.g_bwd.replay()


T_BUCKET = 8       # clip counts are rounded up to a multiple of this: ≤ 27 graphs per family cover every batch of 16 videos × 3..16 clips


def _bucket(T):
    return -(-T // T_BUCKET) * T_BUCKET


def _rows(t, n):
    """the first n rows of a (possibly split) 2-D buffer, still tagged with its lo-plane offset"""
    v = t[:n]
    lo = ops.lo_off(t)
    if lo is not None:
        v._svpc_lo = lo
    return v


class ClipEncoderGraphs:
    def __init__(self, model, max_entries=48, capacity_clips=256):
        self.model = model
        self.max_entries = max_entries
        self.capacity = capacity_clips
        self.entries = collections.OrderedDict()
        self.pool = None
        self.feats_cap = self.ids_cap = self.mask_cap = None
        self.stats = {"hits": 0, "captures": 0, "bypassed": 0}
        self.pending = None                 # the entry whose forward has been replayed and whose backward has not (see check_consumed)
        self._warmed = set()
        import weakref
        _OWNERS.append(weakref.ref(self))

    # -------------------------------------------------------------------------------------------------------------------------
    def usable(self, feats):
        """eager fallback: no gradients wanted, inside a whole-step capture, parameters not yet in the optimizer's arena.  (A data-parallel
        reducer may be attached: its hooks are paused while a graph is captured — ``ops.hooks_paused`` — and the parameters of a replayed
        part, which never fire a hook, are exchanged by ``graph.backward_all(..., exchange=reducer)``: text-side buckets after the eager
        backward, the clip encoder's after its replay — the cut of the three-graph step.)"""
        if not (torch.is_grad_enabled() and feats.is_cuda):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        if torch.cuda.current_stream() == torch.cuda.default_stream():      # (a capture cannot run on the legacy default stream:
            return False                                                   #  run the loop under ``graph.ops_stream()``)
        w = self.model.video_embeddings.video_embeddings[2].weight
        return getattr(w, "_svpc_direct", False)

    def _buffers(self, rows, F, dev):
        if self.feats_cap is None or self.feats_cap.shape[0] < rows or self.feats_cap.shape[1] != F:
            Lv = self.model.config.max_v_len
            cap = max(rows, self.capacity * Lv)
            self.entries.clear()                       # (their graphs read the old buffers)
            self.feats_cap = torch.zeros(cap, F, dtype=torch.float32, device=dev)
            self.ids_cap = torch.zeros(cap, dtype=torch.int32, device=dev)
            self.mask_cap = torch.zeros(cap, dtype=torch.float32, device=dev)

    def run(self, feats, video_rows, ids_v, mask_v, T, cx):
        """[CLS] rows (T, D) of the T valid clips as a fresh autograd leaf; ``model.split_boundary`` is set for the backward"""
        model = self.model
        cfg = model.config
        Lv, F = cfg.max_v_len, feats.shape[1]
        rows = T * Lv
        dev = feats.device
        if self.pending is not None:
            self.pending = None
            raise RuntimeError("clip_graphs: a second forward before the previous step's clip-encoder backward was replayed (its saved "
                               "activations live in the shared graph pool and would be overwritten) — run svpc_amd.graph.backward_all "
                               "after every forward that wants gradients, or call the model under torch.no_grad()")
        # the graph is captured for the clip count rounded UP to a multiple of T_BUCKET: the padding clips are whatever finite rows the
        # capacity buffers hold, their [CLS] rows are dropped and their output gradient is zero, so they contribute exact zeros to every
        # parameter gradient (masked work is exact under the summed loss) — ≤ 27 graphs instead of ≈100 clip counts
        Tb = _bucket(T)
        self._buffers(Tb * Lv, F, dev)
        key = (Tb, Lv, F, ops._PRECISION, cx.training, cx.p_h, cx.p_a, torch.cuda.current_stream().cuda_stream)
        e = self.entries.get(key)
        if e is None:
            e = self._capture(key, Tb, cx)
        else:
            self.entries.move_to_end(key)
            self.stats["hits"] += 1
        torch.index_select(feats, 0, video_rows.long(), out=e.feats[:rows])
        e.ids[:rows].copy_(ids_v)
        e.mask[:rows].copy_(mask_v)
        e.g_fwd.replay()
        e.replays += 1
        e.t_valid = T
        cx.rng._site += e.n_sites
        cut = e.cls[:T].detach().requires_grad_(True)
        model.split_boundary = (e, cut)
        self.pending = e
        return cut

    def _capture(self, key, T, cx):
        model = self.model
        cfg = model.config
        Lv, D = cfg.max_v_len, cfg.hidden_size
        dev = self.feats_cap.device
        e = _Entry()
        e.T, e.replays, e.owner, e.t_valid = T, 0, self, T
        rows = T * Lv
        e.feats, e.ids, e.mask = self.feats_cap[:rows], self.ids_cap[:rows], self.mask_cap[:rows]
        e.seq = ops.SeqInfo.uniform(T, Lv, Lv, dev)
        e.cls_rows = torch.tensor([c * Lv for c in range(T)], dtype=torch.int32, device=dev)
        e.seq_cls = ops.SeqInfo(list(range(T)), [1] * T, [c * Lv for c in range(T)], [Lv] * T, dev)
        e.gout = torch.zeros(T, D, dtype=torch.float32, device=dev)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.current_stream()
        site0 = cx.rng._site
        # one EAGER pass at this clip count first: anything a first call does besides launching kernels (the 256 MB kernel workspace of
        # this stream, lazily uploaded index tables) must not be baked into the graph.  Its backward runs on a zero output gradient:
        # every parameter gradient it accumulates into the optimizer's arena is an exact zero.
        with ops.hooks_paused():                    # (a data-parallel reducer must not see the warm-up's / the capture's parameter writes)
            wkey = (stream.cuda_stream, ops._PRECISION, cx.training)
            if wkey not in self._warmed:            # (once per stream and mode: later clip counts meet the workspace and the kernels' LDS limits set up)
                warm = model._encode_clips(e.feats, None, e.ids, e.mask, e.seq, cx, cls_only=(e.cls_rows, e.seq_cls))
                warm.backward(e.gout)
                ops.join_side()
                del warm
                cx.rng._site = site0
                self._warmed.add(wkey)
            e.g_fwd, e.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with capturing(e.g_fwd, pool=self.pool, stream=stream, capture_error_mode="thread_local", light=True):
                cls = model._encode_clips(e.feats, None, e.ids, e.mask, e.seq, cx, cls_only=(e.cls_rows, e.seq_cls))
            e.cls = cls
            e.n_sites = cx.rng._site - site0
            cx.rng._site = site0                        # (the replay that follows advances the counter)
            with capturing(e.g_bwd, pool=self.pool, stream=stream, capture_error_mode="thread_local", light=True):
                cls.backward(e.gout)
                ops.join_side()
        e.cls = cls.detach()
        self.entries[key] = e
        self.stats["captures"] += 1
        while len(self.entries) > self.max_entries:
            self.entries.popitem(last=False)
        return e


# ------------------------------------------------------------------------------------------------------------------------------------
# The caption decoder (src/rtransformer/model.py:666-694 over all T sentences at once) is the second part of the step whose launch sequence
# depends on the batch structure only through T: T·Lt sentence rows, uniform self-attention segments of Lt, cross-attention over n_mem
# memory rows per sentence; masks and memory rows are data.  It sits in the MIDDLE of the autograd graph (inputs: the text embeddings and
# the memory rows, both with gradients), so the replay is an autograd node: forward copies the inputs into the captured buffers and
# replays, backward copies the incoming gradient, replays the captured backward and hands the two input gradients on.  Its graphs share a
# pool of their own — the clip encoder's saved activations must survive from its forward replay to its backward replay at the end of the
# step, with the decoder's replays in between.
class _DecEntry:
    __slots__ = ("g_fwd", "g_bwd", "xt", "xt_leaf", "mem", "mem_leaf", "mask", "out", "gout", "seq_self", "seq_cross", "n_sites", "T", "lt", "nm")


def _copy_rows(dst, src):
    """dst ← src for plain tensors and for split ones (both planes)"""
    dst.copy_(src)
    if ops.lo_off(src) is not None:
        ops._lo_view(dst).copy_(ops._lo_view(src))


class _DecoderReplay(ops.Function):
    @staticmethod
    def forward(ctx, xt, mem, text_mask, e):
        nx, nmr = xt.shape[0], mem.shape[0]
        _copy_rows(_rows(e.xt, nx), xt)
        _copy_rows(_rows(e.mem, nmr), mem)
        e.mask[:nx].copy_(text_mask)
        if nx < e.mask.shape[0]:
            e.mask[nx:].zero_()                    # the padding sentences of the bucket: all positions masked
        e.g_fwd.replay()
        ctx.e, ctx.n = e, (nx, nmr)
        return e.out[:nx].detach()

    @staticmethod
    def backward(ctx, g):
        e = ctx.e
        nx, nmr = ctx.n
        e.gout[:nx].copy_(g)
        if nx < e.gout.shape[0]:
            e.gout[nx:].zero_()                    # zero output gradient for the padding sentences: exact zeros downstream
        e.g_bwd.replay()
        return e.xt_leaf.grad[:nx], e.mem_leaf.grad[:nmr], None, None


class DecoderGraphs:
    def __init__(self, model, max_entries=48):
        self.model = model
        self.max_entries = max_entries
        self.entries = collections.OrderedDict()
        self.pool = None
        self.stats = {"hits": 0, "captures": 0}
        self._warmed = set()

    def usable(self, xt, mem):
        if not (torch.is_grad_enabled() and xt.is_cuda and xt.requires_grad and mem.requires_grad):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        if torch.cuda.current_stream() == torch.cuda.default_stream():
            return False
        w = self.model.decoder.layer[0].output.dense.weight
        return getattr(w, "_svpc_direct", False)

    def run(self, xt, text_mask, mem, T, cx):
        cfg = self.model.config
        Tb = _bucket(T)          # (captured for the sentence count rounded up to a multiple of T_BUCKET: see ClipEncoderGraphs.run)
        lt, nm = xt.shape[0] // T, mem.shape[0] // T
        key = (Tb, lt, xt.shape[1], str(xt.dtype), ops.lo_off(xt), nm, mem.shape[1], str(mem.dtype), ops.lo_off(mem), ops._PRECISION, cx.training,
               cx.p_h, cx.p_a, torch.cuda.current_stream().cuda_stream)
        e = self.entries.get(key)
        if e is None:
            e = self._capture(key, xt, text_mask, mem, Tb, lt, nm, cx)
        else:
            self.entries.move_to_end(key)
            self.stats["hits"] += 1
        out = _DecoderReplay.apply(xt, mem, text_mask, e)
        cx.rng._site += e.n_sites
        return out

    @staticmethod
    def _like(t, rows):
        """a zero-filled buffer of ``rows`` rows in t's layout (split tensors keep their lo plane) and a leaf over it for the captured
        autograd graph (zero-filled: the eager warm-up pass before the capture reads it, and 0 × garbage is NaN)"""
        lo = ops.lo_off(t)
        if lo is not None:
            buf = torch.zeros(rows, 2 * t.shape[1], dtype=torch.bfloat16, device=t.device)[:, :t.shape[1]]
            buf._svpc_lo = t.shape[1]
        else:
            buf = torch.zeros(rows, t.shape[1], dtype=t.dtype, device=t.device)
        leaf = buf.detach().requires_grad_(True)
        if lo is not None:
            leaf._svpc_lo = buf._svpc_lo
        return buf, leaf

    def _capture(self, key, xt, text_mask, mem, T, Lt, n_mem, cx):
        model = self.model
        cfg = model.config
        dev = xt.device
        e = _DecEntry()
        e.T, e.lt, e.nm = T, Lt, n_mem
        e.seq_self = ops.SeqInfo.uniform(T, Lt, Lt, dev)
        e.seq_cross = ops.SeqInfo.uniform(T, Lt, n_mem, dev)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.current_stream()
        site0 = cx.rng._site
        e.g_fwd, e.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # (the graph's input / output-gradient buffers are made BEFORE the capture: an allocation-with-fill inside it would be replayed)
        e.xt, e.xt_leaf = self._like(xt, T * Lt)
        e.mem, e.mem_leaf = self._like(mem, T * n_mem)
        e.mask = torch.zeros(T * Lt, dtype=text_mask.dtype, device=dev)
        e.gout = torch.zeros(T * Lt, cfg.hidden_size, dtype=torch.float32, device=dev)
        with ops.hooks_paused():
            # (one eager pass first — per stream and mode —, backward on a zero gradient: see ClipEncoderGraphs._capture)
            wkey = (stream.cuda_stream, ops._PRECISION, cx.training, str(xt.dtype), ops.lo_off(xt) is not None)
            if wkey not in self._warmed:
                warm = model.decoder.run(e.xt_leaf, e.mask, e.mem_leaf, e.seq_self, e.seq_cross, None, cx)
                warm.backward(e.gout)
                ops.join_side()
                del warm
                e.xt_leaf.grad = e.mem_leaf.grad = None
                cx.rng._site = site0
                self._warmed.add(wkey)
            with capturing(e.g_fwd, pool=self.pool, stream=stream, capture_error_mode="thread_local", light=True):
                out = model.decoder.run(e.xt_leaf, e.mask, e.mem_leaf, e.seq_self, e.seq_cross, None, cx)
            assert out.shape == e.gout.shape and out.dtype == e.gout.dtype
            e.n_sites = cx.rng._site - site0
            cx.rng._site = site0
            with capturing(e.g_bwd, pool=self.pool, stream=stream, capture_error_mode="thread_local", light=True):
                out.backward(e.gout)
                ops.join_side()
        e.out = out.detach()
        assert e.xt_leaf.grad is not None and e.mem_leaf.grad is not None
        self.entries[key] = e
        self.stats["captures"] += 1
        while len(self.entries) > self.max_entries:
            self.entries.popitem(last=False)
        return e


def enable(model, on=True):
    """replay the T-only parts of the step (clip encoder, caption decoder) from per-clip-count hipGraphs; the training loop's backward must
    be ``svpc_amd.graph.backward_all``"""
    model.clip_graphs = ClipEncoderGraphs(model) if on else None
    model.decoder_graphs = DecoderGraphs(model) if on else None
    return model.clip_graphs, model.decoder_graphs
