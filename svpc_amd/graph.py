"""Whole-step hipGraph capture: {zero_grad, forward, backward, clip + BertAdam (+EMA)} recorded once, replayed per step.

The training step issues ≈2,000 kernel launches; eagerly that is ≈20 ms of host work per step, more than the kernels take.
Everything on the step is capture-safe by construction: kernels launch on PyTorch's current stream, nothing allocates or
synchronises inside the C-ABI, index maps are cached in HBM by ``BatchPlan``, dropout/Gumbel seeds live in HBM and are bumped
by a kernel, and the optimizer's hyper-parameters are read from a device buffer refreshed before each replay.
Inputs are static: refill the tensors handed to the constructor in place (``tensor.copy_``) to train on a new batch of the
same shape (step counts / ingredient counts / copy tables are part of the captured plan — and, with ``model.pack_text_rows``, the
sentence LENGTHS: the packed row maps are captured, so a refilled batch must have the same lengths or the step must be captured again).

Data parallel (``exchange=`` a GradReducer): an eager step is host-bound (≈25 ms of Python/launch work for ≈17 ms of kernels), so
the step is captured as THREE graphs with the RCCL exchange issued eagerly between their replays:
  G1 {zero_grad, forward, backward of everything downstream of the [CLS] rows}   → start all-reduce of the text-side buckets (≈74 %
  of the gradient bytes: decoder, step encoder, simulators, LSTM, head, embeddings) asynchronously,
  G2 {backward of the clip encoder}, which runs while those buckets travel over xGMI           → all-reduce of the remaining buckets,
  G3 {clip + BertAdam (+EMA)} after every bucket has arrived.
The cut uses ``model.split_backward`` (svpc_amd/model.py): the [CLS] rows are the only path from the loss into the clip encoder.
"""
from __future__ import annotations

import gc

import torch


_STREAMS = {}


class capturing:
    """``with capturing(graph, **kw):`` = ``torch.cuda.graph(graph, **kw)`` with the Python garbage collector switched OFF for the
    duration of the capture.  The collector runs whenever allocation counts say so — also in the middle of a capture — and a finaliser
    that makes a HIP call which is not capturable (destroying another hipGraph whose last reference sat in a cycle: an evicted entry of
    the per-clip-count caches; freeing an event) aborts the process ("Fatal Python error: Aborted … Garbage-collecting" inside
    DecoderGraphs._capture, hit by this round's own GPU suite).  No collection of our own here: a full ``gc.collect()`` per capture cost
    ≈ 50 ms each on the bench's heap — 44 of them took the cold ragged leg (22 captures of two graphs in 100 steps) from 74 to 28 steps/s;
    whatever garbage exists is collected by the next automatic run after the capture."""

    def __init__(self, graph, light=False, **kw):
        """light=True (the per-clip-count captures of svpc_amd/clip_graphs.py, tens per epoch): begin / end the capture directly —
        ``torch.cuda.graph.__enter__`` synchronises the device and EMPTIES the caching allocator first (to give a one-off whole-step
        capture as much memory as possible), which makes every allocation after it a fresh hipMalloc."""
        self.light = bool(light)
        if self.light:
            self.graph = graph
            self.pool = kw.get("pool")
            self.stream = kw.get("stream") or torch.cuda.current_stream()
            self.mode = kw.get("capture_error_mode", "global")
            self.stream_ctx = torch.cuda.stream(self.stream)
        else:
            self.cm = torch.cuda.graph(graph, **kw)

    def __enter__(self):
        self.was = gc.isenabled()
        gc.disable()
        try:
            if not self.light:
                return self.cm.__enter__()
            self.stream_ctx.__enter__()
            if self.pool is not None:
                self.graph.capture_begin(self.pool, capture_error_mode=self.mode)
            else:
                self.graph.capture_begin(capture_error_mode=self.mode)
            return None
        except BaseException:
            if self.was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        try:
            if not self.light:
                return self.cm.__exit__(*exc)
            self.graph.capture_end()
            self.stream_ctx.__exit__(*exc)
            return None
        finally:
            if self.was:
                gc.enable()


def ops_stream(device=None):
    """The one warm-up / capture stream of this process per device (a fresh stream per captured object would pin a 256 MB
    kernel workspace each: svpc_amd.ops._ws is per (device, stream))."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _STREAMS.get(dev)
    if st is None:
        st = _STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


_STREAMS_B = {}


def ops_stream_b(device=None):
    """a second capture / replay stream per device: the other half of a batch decoded beside the first (svpc_amd.translator)"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _STREAMS_B.get(dev)
    if st is None:
        st = _STREAMS_B[dev] = torch.cuda.Stream(device=dev)
    return st


_ONES = {}


def unit_grad(loss):
    """a cached tensor of ones shaped like ``loss`` — ``loss.backward(unit_grad(loss))`` instead of ``loss.backward()``, whose implicit
    ``ones_like`` is a fill launch per step"""
    key = (str(loss.device), loss.dtype, tuple(loss.shape))
    g = _ONES.get(key)
    if g is None:
        g = _ONES[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return g


def backward_all(model, loss, exchange=None):
    """``loss.backward()`` plus, when the model cut its autograd graph at the [CLS] rows (``model.split_backward``, or the clip encoder
    replayed from a per-clip-count hipGraph: svpc_amd/clip_graphs.py), the second phase through the clip encoder.  ``exchange``: the
    data-parallel ``GradReducer`` — between the two phases every bucket without a clip-encoder member is started (the text side's
    gradients, ≈74 % of the bytes, are final: those of eager parts were reported by their hooks during the backward, those of replayed
    parts — which fire no hook — are complete because their replay ran inside it), so they travel while the clip encoder's backward
    runs: the cut of the three-graph step.  The caller then calls ``exchange.finish()`` and the optimizer."""
    loss.backward(unit_grad(loss))
    cut = getattr(model, "split_boundary", None)
    if cut is not None:
        if exchange is not None:
            exchange.start_early()
        cut[0].backward(cut[1].grad)
        model.split_boundary = None


class GraphedTrainStep:
    def __init__(self, model, optimizer, forward_args, warmup=2, exchange=None):
        self.model, self.opt, self.args = model, optimizer, forward_args
        self.exchange = exchange
        assert optimizer.arena is not None, "run at least one eager step first (the gradient arena is built lazily)"
        gc.collect()     # stale autograd graphs keep AccumulateGrad nodes bound to the stream of an earlier backward
        # ONE stream for warm-up, capture and any later eager step of this model: autograd binds a leaf's AccumulateGrad node to the
        # stream of the backward that created it, so eager steps on another stream (the legacy default stream above all) make
        # the engine insert cross-stream syncs ("AccumulateGrad node's stream does not match …") that a capture cannot contain.
        # A caller already running on a non-default stream (bench.py does) keeps it; otherwise a dedicated one is made.
        cur = torch.cuda.current_stream()
        self.stream = cur if cur != torch.cuda.default_stream() else ops_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._eager()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        gc.collect()
        unit_grad(torch.empty((), dtype=torch.float32, device=next(model.parameters()).device))     # (exists before the capture begins)
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if exchange is None:
            with capturing(self.graph, stream=self.stream):
                optimizer.zero_grad()
                self.loss = model(*forward_args)[0]
                self.loss.backward(unit_grad(self.loss))
                optimizer.launch()
        else:
            from . import ops
            model.split_backward = True
            # other threads (the collective library's watchdog) keep making driver calls: only this thread's are policed
            with capturing(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                optimizer.zero_grad()
                self.loss = model(*forward_args)[0]
                self.loss.backward(unit_grad(self.loss))
                ops.join_side()
            self.graph_clip = torch.cuda.CUDAGraph()
            with capturing(self.graph_clip, pool=self.graph.pool(), stream=self.stream, capture_error_mode="thread_local"):
                out, cut = model.split_boundary
                out.backward(cut.grad)
                ops.join_side()
            self.graph_opt = torch.cuda.CUDAGraph()
            with capturing(self.graph_opt, pool=self.graph.pool(), stream=self.stream, capture_error_mode="thread_local"):
                optimizer.launch()
        self._versions = optimizer.weights.versions()

    def _eager(self):
        self.opt.zero_grad()
        loss = self.model(*self.args)[0]
        backward_all(self.model, loss)
        if self.exchange is not None:
            self.exchange.finish()
        self.opt.step()
        return loss

    def __call__(self):
        v = self.opt.weights.versions()
        if v != self._versions:          # parameters were edited from Python (load_state_dict …): re-cast the bf16 shadow
            self.opt.weights.refresh()
            self._versions = self.opt.weights.versions()
        self.opt.set_hyper()
        if self.exchange is not None:
            self.exchange.mark_step_start()
        self.graph.replay()
        if self.graph_opt is not None:
            self.exchange.start_early()          # text-side buckets travel while the clip encoder's backward runs
            self.graph_clip.replay()
            self.exchange.finish()
            self.graph_opt.replay()
        self.opt.step_count += 1
        from .optim import WEIGHTS_EPOCH
        WEIGHTS_EPOCH[0] += 1         # (the replay ran the optimizer kernels: see optim.WEIGHTS_EPOCH)
        return self.loss
