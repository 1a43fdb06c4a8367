"""HIP-backed primitives of the hot path: thin ``torch.autograd.Function`` wrappers over the C-ABI
(include/svpc_hip.h).  PyTorch supplies device memory, the current stream and autograd wiring; all arithmetic
runs in the gfx950 kernels of svpc_amd/csrc.  No CPU path exists: tensors must live on the MI355X.

The pure-torch statement of every function here (same signatures) lives in tests/emul_ops.py and is what the
GPU tests compare against.
"""
from __future__ import annotations

import ctypes
import math
import os

import torch
from torch.autograd import Function as _TorchFunction

from . import _lib
from .ops_common import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, FIdx, Idx, SeqInfo, as_idx  # noqa: F401

_WS = {}
_WS_BYTES = 256 << 20


_DEV_IDX = [None]


def _stream():
    """raw handle of PyTorch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object and
    resolves the device through three Python layers: 9 µs a call, ≈240 calls per eager step — the raw getter is ≈0.3 µs.)"""
    i = _DEV_IDX[0]
    if i is None or not _FAST_STREAM:
        return torch.cuda.current_stream().cuda_stream
    return torch._C._cuda_getCurrentRawStream(i)


_FAST_STREAM = os.environ.get("SVPC_SLOW_STREAM", "") == "" and hasattr(torch._C, "_cuda_getCurrentRawStream")


class Function(_TorchFunction):
    """autograd.Function whose ``apply`` goes straight to the C++ binding: the stock classmethod first walks every argument looking for
    dead functorch wrappers (≈12 Python calls per argument — 0.6 ms of an eager step's 1,500 ops); no functorch transform is ever active
    around these ops and none of them defines ``setup_context``."""

    @classmethod
    def apply(cls, *args):
        return super(_TorchFunction, cls).apply(*args)


def _p(t):
    return None if t is None else t.data_ptr()


def _ws(device):
    """One scratch arena per (device, stream) (split-K slabs, reduction partials); ops on a stream use it one at a time."""
    key = (device.index if isinstance(device, torch.device) else str(device), _stream())
    w = _WS.get(key)
    if w is None:
        w = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
        _WS[key] = w
    return w


# Side streams for parameter gradients (OFF by default, SVPC_SIDE=1 / SVPC_BRANCH=1 enable them): a weight / bias gradient written
# straight into the optimizer's arena has no consumer until the optimizer (or the gradient all-reduce) runs, so its kernels can be
# forked onto a side stream and run beside the dgrad chain.  Measured: this won 6 % while the side kernels were slow and numerous;
# since the finalizers / column sums were rewritten and the weight gradients grouped, the captured graph replays 3 % FASTER as one
# linear chain — the graph executor spreads a forked graph over four hardware queues and a main-chain kernel then waits behind
# unrelated side work that happens to share its queue (gaps of 20–80 µs in the kernel trace).  A gradient always uses the same
# side stream (accumulating launches stay ordered); the streams are joined at the end of backward (autograd callback), before a
# gradient bucket is all-reduced, and before the optimizer kernels.  Inside a hipGraph capture the forks/joins become edges.
SIDE_WGRAD = os.environ.get("SVPC_SIDE", "") != ""
_N_SIDE = int(os.environ.get("SVPC_N_SIDE", "2"))
_SIDE = {}
_SIDE_DIRTY = []
_JOIN_QUEUED = [False]


_BRANCH = {}
BRANCH_STREAMS = os.environ.get("SVPC_BRANCH", "") != ""


def branch_stream(device):
    """Stream for an independent forward branch (its autograd nodes run their backward there too); None when disabled."""
    if not BRANCH_STREAMS:
        return None
    st = _BRANCH.get(device)
    if st is None:
        st = _BRANCH[device] = torch.cuda.Stream(device=device)
    return st


# Grouped weight gradients: the wgrad (+ bias gradient) of an fp32-storage linear whose gradients go straight to the arena is not
# launched in its backward but queued; the queue is flushed as ONE grouped launch (svpc_gemm_group_wgrad) on a side stream when it
# is full, when a queued target would be written twice, and at every join point (end of backward, before a gradient bucket is
# all-reduced, before the optimizer).  ≈50 text-side / step-level linears per step × 3 launches (wgrad, column sum, finalize) of
# ≈5–15 µs each become 2–3 launches.
USE_GROUPED_WGRAD = os.environ.get("SVPC_NO_GROUPED_WGRAD", "") == ""
GROUP_BF16 = os.environ.get("SVPC_NO_GROUP_BF16", "") == ""      # also the bf16-stream wgrads (one launch, no split-K)
GROUP_FLUSH_AT = int(os.environ.get("SVPC_GROUP_FLUSH_AT", "16"))
# packed Q/K/V (432 tiles of 64²) and LSTM (576) weight gradients over K = 192 rows are pure latency as launches of their own
GROUP_MAX_TILES = int(os.environ.get("SVPC_GROUP_MAX_TILES", "1200"))
_WQ = []            # (dz, x, wgrad, bgrad)


class _WgradProblem(ctypes.Structure):
    _fields_ = [("dz", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("n_out", ctypes.c_int), ("n_in", ctypes.c_int), ("rows", ctypes.c_int), ("ld_dz", ctypes.c_int),
                ("ld_x", ctypes.c_int), ("ld_dw", ctypes.c_int)]


def _queue_end_of_backward_join():
    if not _JOIN_QUEUED[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side)
            _JOIN_QUEUED[0] = True
        except RuntimeError:      # not inside a backward pass: the caller joins (optimizer / reducer do)
            pass


_WQ16 = []          # bf16-stream problems (dz, x, wgrad, None): one launch at the join points, whole k-loop per tile


P8W_BIAS = os.environ.get("SVPC_NO_P8W_BIAS", "") == ""      # bias gradients of the bf16-stream linears inside the grouped wgrad launch


def _defer_wgrad16(dz, x, wgrad, bgrad=None):
    rows, n_out = dz.shape
    n_in = x.shape[1]
    if n_out % 8 or n_in % 8 or dz.stride(0) % 8 or x.stride(0) % 8 or wgrad.stride(0) % 4 or rows < 1:
        return False
    if (dz.data_ptr() | x.data_ptr()) % 16 or dz.stride(1) != 1 or x.stride(1) != 1 or not wgrad.is_contiguous():
        return False
    wp = wgrad.data_ptr()
    bp = bgrad.data_ptr() if bgrad is not None else -1
    if any(q[2].data_ptr() == wp or (q[3] is not None and q[3].data_ptr() == bp) for q in _WQ16) or \
            len(_WQ16) >= _lib.load().svpc_gemm_group_wgrad_max():
        flush_wgrads()
    _WQ16.append((dz, x, wgrad, bgrad))
    _queue_end_of_backward_join()
    return True


def defer_wgrad(dz, x, wgrad, bgrad):
    """Queue dW += dzᵀ·x (and db += Σ dz) for the grouped launch; False if this problem must be launched on its own."""
    if USE_GROUPED_WGRAD and GROUP_BF16 and _fast() and wgrad is not None and not SIDE_WGRAD and \
            dz.dtype == torch.bfloat16 and x.dtype == torch.bfloat16:
        return _defer_wgrad16(dz, x, wgrad, bgrad)
    if not (USE_GROUPED_WGRAD and _fast() and wgrad is not None and dz.dtype == torch.float32 and x.dtype == torch.float32) or BWD_EXACT:
        return False
    rows, n_out = dz.shape
    n_in = x.shape[1]
    if rows < 1 or n_out % 4 or n_in % 4 or dz.stride(0) % 4 or x.stride(0) % 4 or wgrad.stride(0) % 4:      # any row count: k tail zero-sourced
        return False
    if (dz.data_ptr() | x.data_ptr()) % 16 or dz.stride(1) != 1 or x.stride(1) != 1 or not wgrad.is_contiguous():
        return False
    tiles = -(-n_out // 64) * -(-n_in // 64)
    if tiles > GROUP_MAX_TILES:   # a grid of its own fills the chip for long enough: nothing to gain from grouping
        return False
    wp = wgrad.data_ptr()
    if any(q[2].data_ptr() == wp for q in _WQ):
        flush_wgrads(bf16=False)  # two accumulations into one gradient stay ordered
    _WQ.append((dz, x, wgrad, bgrad))
    if len(_WQ) >= min(GROUP_FLUSH_AT, _lib.load().svpc_gemm_group_wgrad_max()):
        flush_wgrads(bf16=False)
    _queue_end_of_backward_join()
    return True


def flush_wgrads(bf16=True):
    if _WQ16 and bf16:
        probs = (_WgradProblem * len(_WQ16))()
        for i, (dz, x, wg, bg) in enumerate(_WQ16):
            probs[i] = _WgradProblem(dz.data_ptr(), x.data_ptr(), wg.data_ptr(), bg.data_ptr() if bg is not None else None, dz.shape[1],
                                     x.shape[1], dz.shape[0], dz.stride(0), x.stride(0), wg.stride(0))
        ws = _ws(_WQ16[0][0].device)
        # the 8-phase template with transposed fragment reads when every problem is at least 256 wide (gemm_p8w.hip: it also takes the
        # bias gradients from the dz tiles it stages), else the round-1 form with the column sums as a pass of their own
        p8w = USE_P8W and _lib.load().svpc_gemm_group_wgrad_bf16_p8_ok(ctypes.addressof(probs), len(_WQ16)) == 1
        done = list(_WQ16)
        del _WQ16[:]
        if not p8w:
            for i, (dz, x, wg, bg) in enumerate(done):
                if bg is not None:
                    probs[i].db = None
                    defer_colsum(dz, bg)
        _lib.call("gemm_group_wgrad_bf16_p8" if p8w else "gemm_group_wgrad_bf16_ws", ctypes.addressof(probs), len(done), _p(ws),
                  ws.numel() * 4, _stream())
        for _, _, wg, bg in done:
            _ready(wg, "w")
            if bg is not None and p8w:
                _ready(bg, "b")
    if not _WQ:
        return
    dev = _WQ[0][0].device
    pool = _SIDE.get(dev)
    if pool is None:
        pool = _SIDE[dev] = [torch.cuda.Stream(device=dev) for _ in range(_N_SIDE)]
    side = pool[0] if SIDE_WGRAD else torch.cuda.current_stream()
    probs = (_WgradProblem * len(_WQ))()
    for i, (dz, x, wg, bg) in enumerate(_WQ):
        probs[i] = _WgradProblem(dz.data_ptr(), x.data_ptr(), wg.data_ptr(), bg.data_ptr() if bg is not None else None,
                                 dz.shape[1], x.shape[1], dz.shape[0], dz.stride(0), x.stride(0), wg.stride(0))
    if SIDE_WGRAD:
        side.wait_stream(torch.cuda.current_stream())
        for dz, x, _, _ in _WQ:
            dz.record_stream(side); x.record_stream(side)
        if side not in _SIDE_DIRTY:
            _SIDE_DIRTY.append(side)
    _lib.call("gemm_group_wgrad", ctypes.addressof(probs), len(_WQ), side.cuda_stream)
    done = list(_WQ)
    del _WQ[:]
    for _, _, wg, bg in done:
        _ready(wg, "w")
        if bg is not None:
            _ready(bg, "b")


# Deferred reduction tails: the second stage of every bias-gradient column sum and of every LayerNorm gain/shift gradient (≈90 per
# step, 5–6 µs each, a few KB of work) is queued and run as ONE table-driven launch (svpc_multi_finalize) at the join points.
USE_MULTI_FINALIZE = os.environ.get("SVPC_NO_MULTI_FINALIZE", "") == ""
GROUP_COLSUM = os.environ.get("SVPC_NO_GROUP_COLSUM", "") == ""
_FQ = []            # (partial, out0, out1, groups, ncols, split)


class _FinalizeEntry(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p), ("groups", ctypes.c_int),
                ("ncols", ctypes.c_int), ("split", ctypes.c_int)]


def defer_finalize(partial, groups, ncols, out0, out1=None, split=None):
    """out0/out1 (+)= column sums of ``partial`` (groups × ncols), later, together with every other pending tail"""
    tgt = (out0.data_ptr(), out1.data_ptr() if out1 is not None else 0)
    if any(q[1].data_ptr() in tgt or (q[2] is not None and q[2].data_ptr() in tgt) for q in _FQ):
        flush_finalizes()
    _FQ.append((partial, out0, out1, int(groups), int(ncols), int(ncols if split is None else split)))
    if len(_FQ) >= _lib.load().svpc_multi_finalize_max():
        flush_finalizes()
    _queue_end_of_backward_join()


_CQ = []            # pending first stages of bias-gradient column sums: (x, partial)


class _ColsumEntry(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("partial", ctypes.c_void_p), ("dt", ctypes.c_int), ("ldx", ctypes.c_int), ("R", ctypes.c_int),
                ("C", ctypes.c_int)]


def defer_colsum(x, out):
    """out += Σ_rows x, both stages deferred: the column sums of all pending tensors run as one launch, then the finalizes"""
    R, C = x.shape
    partial = torch.empty(_lib.load().svpc_colsum_chunks(R) * C, dtype=torch.float32, device=x.device)
    if len(_CQ) >= 48:
        flush_finalizes()
    _CQ.append((x, partial))
    defer_finalize(partial, _lib.load().svpc_colsum_chunks(R), C, out)


def _flush_colsums():
    if not _CQ:
        return
    ents = (_ColsumEntry * len(_CQ))()
    for i, (x, partial) in enumerate(_CQ):
        ents[i] = _ColsumEntry(x.data_ptr(), partial.data_ptr(), _dt(x), x.stride(0), x.shape[0], x.shape[1])
    _lib.call("multi_colsum", ctypes.addressof(ents), len(_CQ), _stream())
    del _CQ[:]


def flush_finalizes():
    _flush_colsums()
    if not _FQ:
        return
    ents = (_FinalizeEntry * len(_FQ))()
    for i, (partial, o0, o1, g, nc, sp) in enumerate(_FQ):
        ents[i] = _FinalizeEntry(partial.data_ptr(), o0.data_ptr(), (o1 if o1 is not None else o0).data_ptr(), g, nc, sp)
    _lib.call("multi_finalize", ctypes.addressof(ents), len(_FQ), _stream())
    done = list(_FQ)
    del _FQ[:]
    for _, o0, o1, _, _, _ in done:
        _ready(o0, "b" if o1 is None else None)
        if o1 is not None:
            _ready(o1)


def flush_pending():
    """Launch every queued gradient tail (grouped wgrads, column sums, finalizers) and make the current stream wait for the side
    streams that carry gradient work.  Safe at any point of a backward pass: parked residual gradients (``_RES_SINK``) are left
    alone — LayerNorm backwards legitimately keep one parked until the consuming projection's dgrad runs."""
    flush_wgrads()
    flush_finalizes()
    if _SIDE_DIRTY:
        cur = torch.cuda.current_stream()
        for st in _SIDE_DIRTY:
            cur.wait_stream(st)
        del _SIDE_DIRTY[:]


def join_side():
    """End of a backward pass (autograd callback) / before the optimizer kernels: ``flush_pending`` + the leftover check of the
    residual-gradient hand-over.  NOT for use in the middle of backward (a gradient bucket released by a hook calls
    ``flush_pending``): a parked gradient is normal there."""
    flush_pending()
    _JOIN_QUEUED[0] = False
    if _RES_SINK:
        n_left = len(_RES_SINK)
        _RES_SINK.clear()
        raise _lib.SvpcKernelError("residual-gradient hand-over: %d parked gradient(s) were never absorbed by a projection's dgrad "
                                   "(layernorm(..., sink=True) without a consuming ops.linear)" % n_left)


class _side_of:
    """``with _side_of(grad, dz, x):`` — launches inside run on the side stream that owns arena gradient ``grad``; the listed
    tensors are produced on the current stream and read there."""

    def __init__(self, grad, *tensors):
        self.on = SIDE_WGRAD and grad is not None and grad.is_cuda
        if not self.on:
            return
        dev = grad.device
        pool = _SIDE.get(dev)
        if pool is None:
            pool = _SIDE[dev] = [torch.cuda.Stream(device=dev) for _ in range(_N_SIDE)]
        self.stream = pool[(grad.data_ptr() >> 9) % len(pool)]
        self.tensors = tensors

    def __enter__(self):
        if not self.on:
            return self
        self.stream.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            if t is not None:
                t.record_stream(self.stream)
        if self.stream not in _SIDE_DIRTY:
            _SIDE_DIRTY.append(self.stream)
        if not _JOIN_QUEUED[0]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(join_side)
                _JOIN_QUEUED[0] = True
            except RuntimeError:      # not inside a backward pass: the caller joins (optimizer / reducer do)
                pass
        self.cm = torch.cuda.stream(self.stream)
        self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.cm.__exit__(*exc)
        return False


def _need_gpu(t):
    if not t.is_cuda:
        raise _lib.SvpcKernelError("svpc_amd.ops: tensors must be on the GPU (no CPU fallback exists)")
    _DEV_IDX[0] = t.device.index          # (the device the following launches go to: see _stream)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """2-D, unit inner stride (row stride may exceed the width: packed projection outputs are read in place)."""
    assert t.dim() == 2
    if t.stride(1) != 1 and t.shape[1] != 1:
        t = t.contiguous()
    if t.shape[1] == 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t


# ------------------------------------------------------------------------------------------------ RNG
class Rng:
    """Counter-based dropout / Gumbel generator: one 64-bit seed word in HBM (bumped on device every training
    step, so a captured graph draws fresh masks on every replay) + a per-call site id."""

    def __init__(self, device, seed=2019):
        self.device = torch.device(device)
        self.seed = torch.tensor([seed], dtype=torch.int64, device=device)
        self._site = 0

    def begin_step(self, defer=False):
        """a new seed for the step's dropout / Gumbel draws; defer=True: the caller's first launch is ``gather_cast_multi(…, rng=self)``,
        which advances the seed itself (one launch fewer)"""
        self._site = 0
        if not defer:
            _lib.call("bump_seed", _p(self.seed), _stream())

    def site(self):
        self._site += 1
        return self._site

    def mask(self, site, n, p, device):
        out = torch.empty(n, dtype=torch.float32, device=device)
        _lib.call("dropout_mask", _p(out), n, float(p), int(site), _p(self.seed), _stream())
        return out

    def attn_mask(self, site, n_rows, max_k, p, device):
        """keep mask of the attention-probability dropout, (n_rows, max_k), rows (sequence·H + head)·max_q + query"""
        out = torch.empty(n_rows, max_k, dtype=torch.float32, device=device)
        _lib.call("attn_dropout_mask", _p(out), n_rows, max_k, float(p), int(site), _p(self.seed), _stream())
        return out


_DEFAULT_RNG = {}


def make_rng(device, seed=2019):
    return Rng(device, seed)


def default_rng(device):
    key = str(device)
    if key not in _DEFAULT_RNG:
        _DEFAULT_RNG[key] = Rng(device)
    return _DEFAULT_RNG[key]


def _drop_args(drop):
    if drop is None or drop[0] <= 0.0:
        return 0.0, 0, None
    p, rng, site = drop
    return float(p), int(site), rng.seed


# ------------------------------------------------------------------------------------------------ GEMM
class KernelTimer:
    """Optional HIP-event bracket around launches of one kernel class (bench.py's roofline leg).  Events are
    recorded on the stream the kernel is launched on (PyTorch's current stream)."""

    def __init__(self, min_flops=0.0, select=None):
        self.min_flops = min_flops
        self.select = select  # optional predicate on (M, N, K, a_kc, b_kc, a_dt, b_dt, c_dt)
        self.records = []     # (work, start_event, end_event)

    def bracket(self, work, desc=None):
        if work < self.min_flops or (self.select is not None and not self.select(desc)):
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((work, e0, e1, desc))
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for _, e0, e1, _ in self.records)
        work = sum(w for w, _, _, _ in self.records)
        # algorithmic HBM bytes: every operand and the result once, in their storage types
        el = lambda d: 2 if d == 1 else 4      # (split = two bf16 planes = 4 bytes per element)
        nbytes = sum(M * K * el(a) + N * K * el(b) + M * N * el(c) for _, _, _, (M, N, K, _, _, a, b, c) in self.records)
        # an event pair around NOTHING still measures a few µs (record-to-record latency on the stream): calibrate and remove it,
        # so the per-launch average is comparable with the profiler's kernel durations
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for e0, e1 in pairs:
            e0.record(); e1.record()
        torch.cuda.synchronize()
        empty = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[len(pairs) // 2]
        net = max(ms - empty * len(self.records), 0.0)
        return dict(launches=len(self.records), work=work, ms=net, ms_raw=ms, event_overhead_ms=empty, bytes=nbytes)


GEMM_TIMER = None   # set by bench.py
ATTN_TIMER = None   # set by bench.py: {"select": predicate on (n_seq·H, max_q, max_k), "fwd": [(e0, e1, bytes)], "bwd": [...]} — HIP-event
                    # brackets around the attention launches of the clip encoder (the roofline north_star names)


def _attn_bracket(which, n_pairs, max_q, max_k, nbytes):
    t = ATTN_TIMER
    if t is None or not t["select"](n_pairs, max_q, max_k):
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t[which].append((e0, e1, nbytes))
    e0.record()
    return e1

# Arithmetic type of the dense contractions:
#   "fp32"   — v_mfma_f32_32x32x2_f32, fp32 storage: bit-level parity mode (1/16 of the bf16 matrix rate);
#   "bf16"   — bf16 MFMA operands with fp32 accumulate, bf16 activation streams: the throughput mode the baseline names;
#   "bf16x3" — the ≤1e-4-parity throughput mode: every FORWARD contraction is a three-term split-bf16 product
#              (a_lo·b_hi + a_hi·b_lo + a_hi·b_hi on the bf16 matrix cores, fp32 accumulate: ≈2⁻¹⁷ per operand instead of 2⁻⁹), the
#              clip-encoder stream is stored as two bf16 planes per row (hi = bf16(v), lo = bf16(v - hi): the bytes of fp32), all
#              other activations as fp32; the BACKWARD runs the bf16 mode's kernels on the hi planes (gradients have bf16-mode
#              accuracy, the loss / probabilities / token ids have ≈fp32 accuracy).
_PRECISION = "fp32"


def set_precision(p):
    global _PRECISION
    assert p in ("fp32", "bf16", "bf16x3")
    _PRECISION = p


def get_precision():
    return _PRECISION


def _fast():
    """the bf16 matrix cores carry the contractions (either throughput mode)"""
    return _PRECISION != "fp32"


def is_x3():
    return _PRECISION == "bf16x3"


# ---- split tensors (bf16x3 mode).  A split activation is handed around as its HI plane: a bf16 view (R, W) with row stride ≥ 2W of
# a (R, 2W) buffer, tagged with ``_svpc_lo`` = the column offset of the lo plane.  To autograd it is an ordinary bf16 tensor (its
# gradient is a dense (R, W) bf16 tensor); the x3 forward kernels read hi + lo, the bf16 backward kernels read the view in place.
# The tag lives on the Python object: only tensors handed straight from one svpc op to the next carry it (every producer sets it);
# anything else (a torch view / cast / cat) silently drops to the hi plane — use ``to_f32`` to leave the split domain.
def lo_off(t):
    return getattr(t, "_svpc_lo", None)


def require_split_tag(t, who):
    """bf16x3 mode: a bf16 ACTIVATION reaching a forward op must carry its lo-plane tag.  A torch view / cast / detach / cat of a split
    tensor drops the tag silently, and the op would then compute a one-term bf16 product on the hi plane alone — outside the mode's
    ≤ 1e-4 contract with no error anywhere.  Fail instead (use ``to_f32`` to leave the split domain on purpose)."""
    if _PRECISION == "bf16x3" and t is not None and t.dtype == torch.bfloat16 and lo_off(t) is None:
        raise _lib.SvpcKernelError("%s: bf16 tensor without its lo-plane tag in bf16x3 mode (a torch view / cast / cat of a split "
                                   "tensor drops `_svpc_lo`; hand split tensors from op to op, or convert with ops.to_f32)" % who)


def new_split(rows, width, device):
    buf = torch.empty(rows, 2 * width, dtype=torch.bfloat16, device=device)
    hi = buf[:, :width]
    hi._svpc_lo = width
    return hi


def _lo_view(t):
    return torch.as_strided(t.detach(), t.shape, t.stride(), t.storage_offset() + t._svpc_lo)


def _rows_move(src, src_lo, idx, dst, dst_lo, R, W):
    """dst[r] = convert(src[idx ? idx[r] : r]) between storage kinds in one launch (svpc_rows_move; data movement)"""
    sk = 2 if src_lo is not None else _dt(src)
    dk = 2 if dst_lo is not None else _dt(dst)
    _lib.call("rows_move", _p(src), sk, src.stride(0), src_lo or 0, _p(idx), _p(dst), dk, dst.stride(0), dst_lo or 0, R, W, _stream())


class _ToF32(Function):
    @staticmethod
    def forward(ctx, t, lo):
        ctx.dt = t.dtype
        if t.dim() == 2 and t.is_cuda and t.stride(1) == 1:
            out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
            _rows_move(t, lo, None, out, None, t.shape[0], t.shape[1])
            return out
        out = t.float()
        if lo is not None:
            out += torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + lo).float()
        return out

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt), None


class _StreamAndF32(Function):
    """(alias of the stream tensor, its fp32 copy) as ONE node: the backward adds the two gradients and casts in one launch (two separate
    consumers cost a cast of the fp32 gradient, then autograd's add)."""

    @staticmethod
    def forward(ctx, t, lo):
        out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        _rows_move(t, lo, None, out, None, t.shape[0], t.shape[1])
        ctx.set_materialize_grads(False)
        return t.view_as(t), out

    @staticmethod
    def backward(ctx, g_stream, g32):
        if g32 is None:
            return g_stream, None
        g32 = _c(g32)
        a = _c(g_stream) if g_stream is not None else None
        out = torch.empty(g32.shape, dtype=torch.bfloat16, device=g32.device)
        _lib.call("add_cast_bf16", _p(a), _p(g32), _p(out), g32.numel(), _stream())
        return out, None


def stream_and_f32(t):
    """(t, fp32 copy of t) for a bf16 / split 2-D stream tensor with two consumers (one per form); fp32 input: (t, t)"""
    if t.dtype == torch.float32:
        return t, t
    lo = lo_off(t)
    if not (t.dim() == 2 and t.is_cuda and t.stride(1) == 1):
        return t, to_f32(t)
    a, f = _StreamAndF32.apply(t, lo)
    if lo is not None:
        a._svpc_lo = lo
    return a, f


def to_f32(t):
    """fp32 copy of a bf16 / split tensor (hi + lo for a split one; data movement and one exact add)."""
    if t.dtype == torch.float32:
        return t
    return _ToF32.apply(t, lo_off(t))


class _ScatterRows(Function):
    """out (n_rows, W) = zeros with out[idx[r]] = src[r] (idx int64, distinct rows); backward gathers the rows back"""

    @staticmethod
    def forward(ctx, src, idx, n_rows, inv=None):
        ctx.save_for_backward(idx)
        if inv is not None and src.is_cuda and src.dtype == torch.float32:       # inv: padded row → packed row or -1 (one launch)
            src = _c(src)
            out = torch.empty(n_rows, src.shape[1], dtype=torch.float32, device=src.device)
            _lib.call("rows_expand", _p(src), _p(inv), _p(out), n_rows, src.shape[1], _stream())
            return out
        out = torch.zeros(n_rows, src.shape[1], dtype=src.dtype, device=src.device)
        out.index_copy_(0, idx, src)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return torch.index_select(g, 0, idx), None, None, None


def scatter_rows(src, idx, n_rows, inv=None):
    """rows of a packed tensor back into a padded layout (zeros elsewhere); data movement.  inv (int32, n_rows): the packed row of every
    padded row or -1 — with it the expansion is one launch"""
    return _ScatterRows.apply(src, idx, int(n_rows), inv)


def take_rows_f32(t, idx):
    """fp32 rows ``idx`` of a (possibly split / bf16) 2-D tensor."""
    lo = lo_off(t)
    if lo is None:
        return torch.index_select(t, 0, idx).float()
    return _TakeSplit.apply(t, idx, lo)[0]


def take_rows_f32_alias(t, idx):
    """(fp32 rows ``idx`` of ``t``, an alias of ``t``) for a stream tensor with exactly ONE other consumer: that consumer takes the alias,
    and the gradient of the gathered rows is added into the alias's gradient IN PLACE (one small scatter launch) instead of a zero-filled
    stream-sized tensor, an index_add and autograd's stream-sized sum.  Plain tensors: (rows, t)."""
    lo = lo_off(t)
    if (lo is None and t.dtype != torch.bfloat16) or not t.is_cuda or idx.dtype != torch.int32 or t.dim() != 2:
        return take_rows_f32(t, idx), t
    rows, alias = _TakeSplit.apply(t, idx, lo)
    if lo is not None:
        alias._svpc_lo = lo
    return rows, alias


class _TakeSplit(Function):
    @staticmethod
    def forward(ctx, t, idx, lo):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.dt = t.shape, t.dtype
        ctx.set_materialize_grads(False)         # (an unused alias must reach backward as None, not as a stream-sized tensor of zeros)
        if t.is_cuda and idx.dtype == torch.int32 and t.stride(1) == 1:
            out = torch.empty(idx.numel(), t.shape[1], dtype=torch.float32, device=t.device)
            _rows_move(t, lo, idx, out, None, idx.numel(), t.shape[1])
        elif lo is not None:
            low = torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + lo)
            out = torch.index_select(t, 0, idx).float() + torch.index_select(low, 0, idx).float()
        else:
            out = torch.index_select(t, 0, idx).float()
        return out, t.view_as(t)

    @staticmethod
    def backward(ctx, g, g_alias):
        idx, = ctx.saved_tensors
        if g is None:
            return g_alias, None, None
        if (g_alias is not None and g_alias.is_cuda and g_alias.dtype == torch.bfloat16 and g_alias.is_contiguous() and idx.dtype == torch.int32
                and tuple(g_alias.shape) == tuple(ctx.shape)):
            g = _c(g.float())
            _lib.call("scatter_add_rows_bf16", _p(g), _p(idx), _p(g_alias), g_alias.stride(0), idx.numel(), g.shape[1], _stream())
            return g_alias, None, None
        out = torch.zeros(ctx.shape, dtype=ctx.dt, device=g.device)
        out.index_add_(0, idx.long(), g.to(ctx.dt))
        if g_alias is not None:
            out = out + g_alias
        return out, None, None


# bf16 activation stream: in "bf16" precision the clip encoder and the decoder keep their activations (and their gradients) in
# HBM as bf16 — half the bytes for every LayerNorm / attention / GEMM operand; statistics, softmax, accumulation and all parameter
# gradients stay fp32.  Any row count qualifies (the direct-to-LDS GEMM clamps its M/N edges and zero-fills the K tail of the
# wgrad); feature dims must be multiples of 32.  Without the direct-to-LDS kernel only interior-only shapes do (multiples of 128).
BF16_STREAM = True
USE_GLDS = True        # direct-to-LDS GEMM for bf16 × bf16 interior shapes
USE_L32 = True         # direct-to-LDS GEMM for fp32 × fp32 operands (latency-bound text / step-level side)


def _dt(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def bf16_stream_ok(rows, *dims):
    if not (_fast() and BF16_STREAM and rows > 0):
        return False
    if is_x3():          # the split stream's GEMM (gemm_p8x3.hip) walks 64-deep k-tiles
        return USE_GLDS and all(d % 64 == 0 for d in dims)
    if USE_GLDS:
        return all(d % 32 == 0 for d in dims)
    return rows % 128 == 0 and all(d % 128 == 0 for d in dims)


# ---- backward ablation switches (tests/tools/bwd_ablation.py; all False in the product): which part of the bf16 backward of the bf16x3
# mode moves a training trajectory away from the fp32 one?  BWD_EXACT: every contraction on fp32 storage that is NOT a three-term forward
# product (i.e. the dgrads / wgrads of the text / step-level side) runs on the exact f32 MFMA, ungrouped; ATTN_BWD_EXACT: attention on
# fp32 storage takes the exact fp32 backward kernel instead of the matrix-core one.  (The third switch is ``BF16_STREAM`` below: False
# keeps every activation — and so every gradient — in fp32 storage.)
BWD_EXACT = False
ATTN_BWD_EXACT = False


def _gemm(A, lda, a_kc, B, ldb, b_kc, C, M, N, K, Z=None, bias=None, act=ACT_NONE, p=0.0, site=0, seed=None, accumulate=0, R=None, G=None,
          x3=False):
    """C = epi(A·B) (+ R: an addend of C's type and layout, only on the bf16 direct-to-LDS path; other paths add it afterwards).
    G = (aux, act): C = (A·B) ⊙ act'(aux) — absorbed only by the bf16 direct-to-LDS path; returns whether it was applied.
    x3 (fp32 operands, bf16x3 mode, forward products): three-term split-bf16 products (svpc_gemm_l32_x3); shapes that kernel does
    not take (a k-strided operand with K % 32 != 0, K % 4 != 0) run on the exact f32 MFMA instead — never on one-term bf16."""
    ws = _ws(C.device)
    g_done = False
    ev = (GEMM_TIMER.bracket(2.0 * M * N * K, (M, N, K, a_kc, b_kc, _dt(A), _dt(B), _dt(C)))
          if GEMM_TIMER is not None else None)
    if ev:
        ev[0].record()
    if BWD_EXACT and not x3 and A.dtype == B.dtype == C.dtype == torch.float32:
        _lib.call("gemm_f32", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Z), M, N, K, _p(bias), act, p, site,
                  _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
    elif x3 and A.dtype == B.dtype == C.dtype == torch.float32:
        if (A.data_ptr() | B.data_ptr()) % 16 == 0 and _lib.load().svpc_gemm_l32_supported(a_kc, b_kc, lda, ldb, M, N, K) == 1:
            _lib.call("gemm_l32_x3", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Z), None, M, N, K, _p(bias), act, p, site,
                      _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
        else:
            _lib.call("gemm_f32", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Z), M, N, K, _p(bias), act, p, site,
                      _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
    elif (_fast() and USE_P8T and a_kc == 1 and b_kc == 0 and A.dtype == B.dtype == C.dtype == torch.bfloat16 and Z is None and bias is None
          and act == ACT_NONE and p <= 0.0 and not accumulate and (-(-M // 256)) * (-(-N // 256)) >= P8T_MIN_TILES
          and (A.data_ptr() | B.data_ptr() | C.data_ptr()) % 16 == 0
          and _lib.load().svpc_gemm_p8t_supported(lda, ldb, C.stride(0), M, N, K) == 1
          and (R is None or (R.dtype == torch.bfloat16 and R.shape == C.shape and R.stride() == C.stride() and R.data_ptr() % 16 == 0))
          and (G is None or (G[1] in (ACT_RELU, ACT_GELU) and G[0].dtype == torch.bfloat16 and G[0].shape == C.shape
                             and G[0].stride() == C.stride() and G[0].data_ptr() % 16 == 0))):
        # stream dgrad on the 8-phase template with the weight matrix read k-strided in place (gemm_p8t.hip)
        _lib.call("gemm_p8t", _p(A), lda, _p(B), ldb, _p(C), C.stride(0), _p(G[0]) if G is not None else None, int(G[1]) if G is not None else 0,
                  _p(R), M, N, K, _stream())
        R = None
        g_done = G is not None
    elif (_fast() and USE_S4T and a_kc == 1 and b_kc == 0 and A.dtype == B.dtype == C.dtype == torch.bfloat16 and Z is None and bias is None
          and act == ACT_NONE and p <= 0.0 and not accumulate and G is None and S4T_MIN_TILES <= (-(-M // 128)) * (-(-N // 128)) <= S4T_MAX_TILES
          and (A.data_ptr() | B.data_ptr() | C.data_ptr()) % 16 == 0
          and _lib.load().svpc_gemm_s4t_supported(lda, ldb, C.stride(0), M, N, K) == 1
          and (R is None or (R.dtype == torch.bfloat16 and R.shape == C.shape and R.stride() == C.stride() and R.data_ptr() % 16 == 0))):
        # small-M stream dgrad (the decoder's 4,224 / 576 rows) on 128² tiles with the weight matrix read k-strided in place (gemm_s4t.hip)
        _lib.call("gemm_s4t", _p(A), lda, _p(B), ldb, _p(C), C.stride(0), _p(R), M, N, K, _stream())
        R = None
    elif _fast() and USE_GLDS and A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and \
            (A.data_ptr() | B.data_ptr()) % 16 == 0 and _lib.load().svpc_gemm_glds_supported(a_kc, b_kc, lda, ldb, M, N, K) == 1:
        g_ok = G is not None and G[0].dtype == C.dtype and G[0].shape == C.shape and G[0].stride() == C.stride() and not accumulate
        _lib.call("gemm_glds_rg", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), _dt(C), C.stride(0), _p(Z), _p(R), _p(G[0]) if g_ok else None,
                  int(G[1]) if g_ok else 0, M, N, K, _p(bias), act, p, site, _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
        R = None
        g_done = g_ok
    elif _fast() and USE_L32 and A.dtype == B.dtype == C.dtype == torch.float32 and \
            (A.data_ptr() | B.data_ptr()) % 16 == 0 and _lib.load().svpc_gemm_l32_preferred(a_kc, b_kc, lda, ldb, M, N, K) == 1:
        if R is not None and not (R.dtype == torch.float32 and R.is_contiguous() and R.shape == C.shape and C.is_contiguous()):
            Rk = None
        else:
            Rk, R = R, None       # absorbed by the epilogue
        g_ok = (G is not None and G[0].dtype == torch.float32 and G[0].shape == C.shape and G[0].is_contiguous() and C.is_contiguous()
                and Z is None and bias is None and act == ACT_NONE and p <= 0.0)
        if g_ok:                  # C = (A·B) ⊙ act'(aux) + R: the activation backward of the tensor this dgrad differentiates
            _lib.call("gemm_l32_rg", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Rk), _p(G[0]), int(G[1]), M, N, K,
                      accumulate, _p(ws), ws.numel() * 4, _stream())
            g_done = True
        else:
            _lib.call("gemm_l32_r", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Z), _p(Rk), M, N, K, _p(bias), act, p, site,
                      _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
    elif _fast():
        dt = lambda t: 1 if t.dtype == torch.bfloat16 else 0
        _lib.call("gemm_mx", _p(A), dt(A), lda, a_kc, _p(B), dt(B), ldb, b_kc, _p(C), dt(C), C.stride(0), _p(Z), M, N, K, _p(bias),
                  act, p, site, _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
    else:
        _lib.call("gemm_f32", _p(A), lda, a_kc, _p(B), ldb, b_kc, _p(C), C.stride(0), _p(Z), M, N, K, _p(bias), act, p, site,
                  _p(seed), accumulate, _p(ws), ws.numel() * 4, _stream())
    if R is not None:
        if G is not None and not g_done:
            raise _lib.SvpcKernelError("gemm: an activation-backward factor and an addend need the direct-to-LDS path")
        C.add_(R)
    if ev:
        ev[1].record()
    return g_done


def _colsum(x2d, idx=None, K=1, out=None, accumulate=0):
    R, Cc = x2d.shape
    if out is None:
        out = torch.empty(K, Cc, dtype=torch.float32, device=x2d.device)
        if R == 0:
            return out.zero_()
    if R == 0:
        return out
    ws = _ws(x2d.device)
    _lib.call("bucket_colsum_t", _p(x2d), _dt(x2d), x2d.stride(0), _p(idx), R, Cc, K, _p(out), accumulate, _p(ws), _stream())
    return out


# Direct-to-arena parameter gradients: once the optimizer has re-pointed ``p.grad`` into its contiguous arena (and marked
# the parameter), backward kernels accumulate weight / bias / gain gradients straight into that storage (GEMM epilogue
# ``accumulate``, reduction finalizers) instead of returning a tensor for autograd to add — no temporaries, no add kernels.
GRAD_READY_HOOK = None     # set by GradReducer: called with the data_ptr of every arena gradient that has just been written


def _direct(p):
    if p is None or not p.is_leaf:
        return None
    g = getattr(p, "grad", None)
    return g if (g is not None and getattr(p, "_svpc_direct", False)) else None


def direct_grads(*params):
    """the arena gradients of ``params`` when every one of them is written in place (else None) — for ``linear(..., bgrad=(…))`` with a
    bias that is a sum of parameters"""
    gs = tuple(_direct(p) for p in params)
    return gs if all(g is not None for g in gs) else None


HOOKS_PAUSED = [0]         # > 0: gradient-ready notifications (pointer reports here, the reducer's post-accumulate hooks) are ignored


class hooks_paused:
    """``with ops.hooks_paused():`` — while a part of the step is warmed up / captured into a hipGraph (svpc_amd/clip_graphs.py) the
    data-parallel reducer must not count the parameter writes of that pass (nor issue a collective from inside a capture)."""

    def __enter__(self):
        HOOKS_PAUSED[0] += 1
        return self

    def __exit__(self, *exc):
        HOOKS_PAUSED[0] -= 1
        return False


def _ready(t, kind=None):
    if GRAD_READY_HOOK is not None and t is not None and not HOOKS_PAUSED[0]:
        GRAD_READY_HOOK(t.data_ptr(), t.numel(), kind)


def _shadow(w):
    """bf16 shadow of parameter ``w`` (svpc_amd.optim.WeightStore), re-cast first if ``w`` changed behind the store's back.  A store
    built with the lo plane (bf16x3 mode) keeps it ``_svpc_lo`` elements behind the shadow; it is re-cast together with it."""
    s = getattr(w, "_svpc_bf16", None)
    if s is None:
        return None
    if w._version != w._svpc_bf16_ver:
        with torch.no_grad():
            s.copy_(w.detach())
            if lo_off(s) is not None:
                split_planes(w.detach(), s, _lo_view_flat(s))
        w._svpc_bf16_ver = w._version
    return s


def _lo_view_flat(s):
    """the lo plane of a weight-shadow view (same shape and strides, ``_svpc_lo`` elements further on in the store's buffer)"""
    return torch.as_strided(s, s.shape, s.stride(), s.storage_offset() + s._svpc_lo)


def split_planes(src, hi, lo):
    """hi ← bf16(src), lo ← bf16(src - hi) (weights leaving the fp32 master copy; data preparation, not on the step's hot path)"""
    with torch.no_grad():
        hi.copy_(src)
        lo.copy_(src - hi.float())


def _transient_split(w):
    """(hi, lo offset) planes of a weight that has no store (first step, tests): one 2·numel bf16 buffer"""
    buf = torch.empty(2, *w.shape, dtype=torch.bfloat16, device=w.device)
    split_planes(w.detach(), buf[0], buf[1])
    hi = buf[0]
    hi._svpc_lo = w.numel()
    return hi


# Residual-gradient hand-over.  In `y = LayerNorm(sub(h) + h)` the gradient of h has two parts: the LayerNorm's dh (residual path)
# and the dgrad of sub's first projection.  Autograd would add them with a separate kernel (≈30 per step on the bf16 streams);
# instead a LayerNorm called with sink=True parks its dh here, keyed by h's address, and returns no residual gradient, and the
# backward of the projection that consumes h adds the parked tensor in its dgrad epilogue (C = dz·W + R).  The caller promises
# that exactly one ops.linear consumes h and needs its input gradient; join_side() fails loudly if a parked gradient is left over.
USE_RES_SINK = os.environ.get("SVPC_RES_SINK", "1") != "0"
FUSE_ACT_BWD = os.environ.get("SVPC_FUSE_ACT_BWD", "1") != "0"      # activation backward inside the following projection's dgrad
FUSE_ACT_BWD_F32 = os.environ.get("SVPC_FUSE_ACT_BWD_F32", "1") != "0"      # … also for fp32-storage projections (step-wise encoder)
_RES_SINK = {}
SINK_STATS = [0, 0]        # parked by LayerNorm backwards / absorbed by dgrad epilogues (since import)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, act, trans_w, drop, wgrad, bgrad, w16, tok_out=None, tok_in=None):
        _need_gpu(x)
        x_lo = lo_off(x)
        x = _rows2d(x)
        w_lo = lo_off(w16) if w16 is not None else None
        w = _c(w if w16 is None else w16)      # bf16 shadow: both GEMM operands stream straight into LDS
        M, K = x.shape
        N = w.shape[1] if trans_w else w.shape[0]
        p, site, seed = _drop_args(drop)
        if x_lo is not None:
            # bf16x3 stream: split activations × split weight shadow → split output (gemm_p8x3.hip); the pre-activation copy z is a
            # plain bf16 matrix (all the bf16 backward needs, for ReLU too: z > 0 ⇔ y > 0)
            if w_lo is None or trans_w or p > 0.0:
                raise _lib.SvpcKernelError("linear: a split activation needs a split weight shadow, an (out, in) weight and no dropout")
            y = new_split(M, N, x.device)
            z = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if act != ACT_NONE else None
            # (bench.py's roofline bracket: dtype code 2 = split operands; work = the product's 2·M·N·K, the kernel issues 3× that)
            ev = GEMM_TIMER.bracket(2.0 * M * N * K, (M, N, K, 1, 1, 2, 2, 2)) if GEMM_TIMER is not None else None
            if ev:
                ev[0].record()
            # 256×256 tiles when they fill the chip, else 128×128 (the decoder's 4,224 / 576 rows)
            kern = "gemm_p8x3" if (-(-M // 256)) * (-(-N // 256)) >= X3_BIG_TILES else "gemm_s4x3"
            _lib.call(kern, _p(x), x.stride(0), x_lo, _p(w), w.stride(0), w_lo, _p(y), y.stride(0), y._svpc_lo, _p(z), N, M, N, K,
                      _p(b), act, _stream())
            if ev:
                ev[1].record()
            ctx.save_for_backward(x, w, z)
            ctx.cfg = (act, trans_w, p, site, seed, b is not None)
            ctx.direct = (wgrad, bgrad)
            ctx.tok_out, ctx.tok_in = tok_out, tok_in
            if tok_out is not None:
                tok_out["aux"], tok_out["act"] = z, act
            return y
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        z = torch.empty_like(y) if act == ACT_GELU else None
        _gemm(x, x.stride(0), 1, w, w.stride(0), 0 if trans_w else 1, y, M, N, K, Z=z, bias=b, act=act, p=p, site=site, seed=seed,
              x3=is_x3())
        ctx.save_for_backward(x, w, z if act == ACT_GELU else (y if act != ACT_NONE else None))
        ctx.cfg = (act, trans_w, p, site, seed, b is not None)
        ctx.direct = (wgrad, bgrad)
        # activation backward folded into the NEXT projection's dgrad (see linear(): fuse_act_bwd): this node publishes what its
        # activation's backward needs; the consumer's backward reports the gradient it has already multiplied
        ctx.tok_out, ctx.tok_in = tok_out, tok_in
        if tok_out is not None:
            tok_out["aux"], tok_out["act"] = (z if act == ACT_GELU else y), act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, aux = ctx.saved_tensors
        act, trans_w, p, site, seed, has_b = ctx.cfg
        M, K = x.shape
        N = w.shape[1] if trans_w else w.shape[0]
        dy = _c(dy)
        tok = ctx.tok_out
        premul = tok is not None and tok.get("done") is not None
        if premul:
            if tok["done"] != (dy.data_ptr(), tuple(dy.shape)):
                raise _lib.SvpcKernelError("fused activation backward: the activated tensor has a second consumer (its gradient is not "
                                           "the one the following projection wrote)")
            tok["done"] = None
        if premul:
            dz = dy                  # the following projection's dgrad has applied act'(.) already
        elif act != ACT_NONE or p > 0.0:
            dz = torch.empty_like(dy)
            _lib.call("act_bwd_t", _p(dy), _p(aux if aux is not None else dy), _p(dz), _dt(dy), dy.numel(), act, p, site, _p(seed),
                      _stream())
        else:
            dz = dy
        dx = dw = db = None
        wgrad, bgrad = ctx.direct
        if isinstance(bgrad, tuple):
            # the bias is the sum of several parameters (the LSTM's b_ih + b_hh, model.py:1022): each of them receives Σ_rows dz in place —
            # the first the usual way below, the others through one more entry of the grouped column sum
            for e in bgrad[1:]:
                V = 8 if dz.dtype == torch.bfloat16 else 4
                if USE_MULTI_FINALIZE and not SIDE_WGRAD and N % V == 0 and dz.stride(0) % V == 0 and dz.data_ptr() % 16 == 0 and M > 0:
                    defer_colsum(dz, e)
                else:
                    with _side_of(e, dz):
                        _colsum(dz, out=e.view(1, -1), accumulate=1)
                    _ready(e, "b")
            bgrad = bgrad[0]
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=x.dtype, device=dy.device)
            parked = _RES_SINK.pop(x.data_ptr(), None) if _RES_SINK else None
            if parked is not None:
                SINK_STATS[1] += 1
            if parked is not None and (parked.shape != dx.shape or parked.dtype != dx.dtype):
                raise _lib.SvpcKernelError("residual-gradient hand-over: parked gradient does not match the projection's input")
            tin = ctx.tok_in
            G = (tin["aux"], tin["act"]) if (tin is not None and tin.get("aux") is not None and parked is None) else None
            if _gemm(dz, N, 1, w, w.stride(0), 1 if trans_w else 0, dx, M, K, N, R=parked, G=G):
                tin["done"] = (dx.data_ptr(), tuple(dx.shape))
        w_done = False
        if wgrad is not None and not trans_w and (not has_b or bgrad is not None):
            if dz.dtype == torch.bfloat16:
                # bf16 stream: grouped wgrad; the bias gradient rides along (taken from the dz tiles the launch stages) when it is a
                # plain fp32 arena vector, else it goes its usual way below
                ride = (P8W_BIAS and USE_P8W and has_b and bgrad is not None and bgrad.dtype == torch.float32 and bgrad.is_contiguous()
                        and bgrad.numel() == N and min(N, K) >= 256)
                if defer_wgrad(dz, x, wgrad, bgrad if ride else None):
                    w_done = True
                    if ride:
                        return dx, None, None, None, None, None, None, None, None, None, None
            elif defer_wgrad(dz, x, wgrad, bgrad if has_b else None):
                return dx, None, None, None, None, None, None, None, None, None, None
        if not w_done and (wgrad is not None or ctx.needs_input_grad[1]):
            acc = 1 if wgrad is not None else 0
            dw = wgrad if wgrad is not None else torch.empty_like(w)
            with _side_of(wgrad, dz, x):
                if trans_w:   # w (K, N): dw = xᵀ dz
                    _gemm(x, x.stride(0), 0, dz, N, 0, dw, K, N, M, accumulate=acc)
                else:         # w (N, K): dw = dzᵀ x
                    _gemm(dz, N, 0, x, x.stride(0), 0, dw, N, K, M, accumulate=acc)
            if wgrad is not None:
                _ready(wgrad, "w")
                dw = None
        if has_b and (bgrad is not None or ctx.needs_input_grad[2]):
            if bgrad is not None:
                V = 8 if dz.dtype == torch.bfloat16 else 4
                if (USE_MULTI_FINALIZE and not SIDE_WGRAD and N % V == 0 and dz.stride(0) % V == 0 and dz.data_ptr() % 16 == 0 and M > 0):
                    if GROUP_COLSUM:
                        defer_colsum(dz, bgrad)
                    else:
                        chunks = _lib.load().svpc_colsum_chunks(M)
                        partial = torch.empty(chunks * N, dtype=torch.float32, device=dz.device)
                        _lib.call("colsum_partial_t", _p(dz), _dt(dz), dz.stride(0), M, N, _p(partial), _stream())
                        defer_finalize(partial, chunks, N, bgrad)
                else:
                    with _side_of(bgrad, dz):
                        _colsum(dz, out=bgrad.view(1, -1), accumulate=1)
                    _ready(bgrad, "b")
            else:
                db = _colsum(dz).view(-1)
        return dx, dw, db, None, None, None, None, None, None, None, None


# ---- bf16x3 ablation experiments (tools/x3_ablation.py; None in the product): which contractions need their lo planes?  One family of
# projections at a time is degraded — its activation operand's lo plane dropped ("a"), its weight operand's ("b"), its output's ("o": the
# stored value becomes plain bf16) — by handing the UNCHANGED three-term kernels operands whose lo plane is zero (fp32-storage families:
# operands rounded to bf16 first), forward only, under no_grad.  ABLATE = {"match": f(param_name) -> str of "abo" flags or "", "names":
# {data_ptr: parameter name}}.
ABLATE = None


def _ablate_flags(w):
    if ABLATE is None:
        return ""
    name = ABLATE["names"].get(w.data_ptr())
    return ABLATE["match"](name) if name is not None else ""


def _zero_lo_copy(t):
    lo = lo_off(t)
    if lo is None:
        return t.to(torch.bfloat16).float() if t.dtype == torch.float32 else t
    c = new_split(t.shape[0], t.shape[1], t.device)
    c.copy_(t.detach())
    _lo_view(c).zero_()
    return c


def linear(x, w, b=None, act=ACT_NONE, trans_w=False, drop=None, wgrad=None, bgrad=None, w16=None, fuse_act_bwd=False):
    """fuse_act_bwd=True (with an activation, no dropout): the caller promises that the returned tensor is consumed by exactly ONE
    ops.linear; on a bf16 stream that projection's dgrad then writes the gradient of this projection's pre-activation directly
    (C = dz·W ⊙ act'(z), svpc_gemm_glds_rg) and the separate activation-backward pass over the stream is skipped.  A second consumer
    is detected in backward and fails loudly."""
    tok_in = getattr(x, "_svpc_act_tok", None)
    require_split_tag(x, "linear")
    abl = _ablate_flags(w) if ABLATE is not None else ""
    if abl:
        assert not torch.is_grad_enabled(), "ablation experiments are forward-only"
        if "a" in abl:
            x = _zero_lo_copy(x)
        if "b" in abl:
            if w16 is None:
                w16 = _shadow(w)
            if w16 is not None and lo_off(w16) is not None:          # (a weight shadow keeps its lo plane numel-strided, like _transient_split)
                buf = torch.zeros(2, *w16.shape, dtype=torch.bfloat16, device=w16.device)
                buf[0].copy_(w16)
                w16 = buf[0]
                w16._svpc_lo = w16.numel()
            w = w.detach().to(torch.bfloat16).float()
    split = lo_off(x) is not None
    if x.dtype == torch.bfloat16:
        n_out = w.shape[1] if trans_w else w.shape[0]
        ok = bf16_stream_ok(x.shape[0], x.shape[1], n_out) and not trans_w and drop is None
        if ok and split:
            ok = act != ACT_SIGMOID and _lib.load().svpc_gemm_p8x3_supported(x.stride(0), x._svpc_lo, x.shape[1], 2 * n_out, n_out, n_out,
                                                                             x.shape[0], n_out, x.shape[1]) == 1
        if not ok:
            x = to_f32(x)          # shapes outside the bf16-stream GEMM variants fall back to fp32 storage
            split = False
    if wgrad is None:
        wgrad = _direct(w)
    if bgrad is None and b is not None:
        bgrad = _direct(b)
    if x.dtype == torch.bfloat16 and USE_GLDS:
        if w16 is None:
            w16 = _shadow(w)
        if split and (w16 is None or lo_off(w16) is None):
            w16 = _transient_split(w)          # no store with a lo plane (first step, a store built in another mode): split on the fly
        if w16 is None:            # no weight store yet (first step, inference without one): a transient bf16 copy
            w16 = w.detach().to(torch.bfloat16)
    else:
        w16 = None
    # (bf16 streams: the direct-to-LDS dgrads; fp32 storage in the fast modes: svpc_gemm_l32_rg)
    can_g = (x.dtype == torch.bfloat16 and USE_GLDS) or (x.dtype == torch.float32 and USE_L32 and FUSE_ACT_BWD_F32 and _fast())
    fuse = fuse_act_bwd and FUSE_ACT_BWD and act != ACT_NONE and drop is None and can_g and torch.is_grad_enabled()
    tok_out = {} if fuse else None
    if tok_in is not None and (trans_w or not can_g):
        tok_in = None
    y = _Linear.apply(x, w, b, act, trans_w, drop, wgrad, bgrad, w16, tok_out, tok_in)
    if split:
        y._svpc_lo = y.shape[1]        # (a split activation yields a split output: see _Linear.forward)
    if abl and "o" in abl:
        if split:
            _lo_view(y).zero_()
        else:
            y = y.to(torch.bfloat16).float()
    if tok_out is not None:
        y._svpc_act_tok = tok_out
    return y


# ------------------------------------------------------------------------------------------------ LayerNorm family
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, add2, eps, src_rows, pad_row, pre_drop, post_drop, add1, add1_mod, add2_idx,
                out_bf16=False, sink=False, out_split=False):
        _need_gpu(x)
        ctx.sink = bool(sink)
        ctx.direct = (_direct(gamma), _direct(beta), _direct(x) if src_rows is not None else None,
                      _direct(add2) if add2 is not None else None)
        x_lo = lo_off(x)
        r_lo = lo_off(residual) if residual is not None else None
        split = out_split or x_lo is not None or r_lo is not None
        if x_lo is None:
            x = _c(x)
        D = x.shape[1]
        R = src_rows.numel() if src_rows is not None else x.shape[0]
        p_pre, s_pre, seed1 = _drop_args(pre_drop)
        p_post, s_post, seed2 = _drop_args(post_drop)
        seed = seed1 if seed1 is not None else seed2
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        add1 = _c(add1) if add1 is not None else None
        if split:
            # bf16x3 stream: x is fp32 (the gathered feature / embedding rows) or split, the residual is split, y is split
            if (x_lo is None and x.dtype != torch.float32) or (residual is not None and r_lo is None):
                raise _lib.SvpcKernelError("layernorm: a split output needs fp32 or split inputs and a split residual")
            y = new_split(R, D, x.device)
            _lib.call("ln_fwd_s", _p(x), 2 if x_lo is not None else 0, x.stride(0), x_lo or 0, _p(src_rows), _p(residual),
                      residual.stride(0) if residual is not None else 0, r_lo or 0, _p(gamma), _p(beta), _p(y), 2, y.stride(0), y._svpc_lo,
                      _p(mean), _p(rstd), R, D, float(eps), p_pre, s_pre, p_post, s_post, _p(seed), _p(add1), int(add1_mod), _p(add2),
                      _p(add2_idx), _stream())
            ctx.save_for_backward(x, gamma, residual, mean, rstd, src_rows, add2_idx, seed)      # (the hi planes: bf16 views)
            ctx.cfg = (R, D, p_pre, s_pre, p_post, s_post, pad_row, add2.shape[0] if add2 is not None else 0)
            return y
        residual = _c(residual) if residual is not None else None
        y_dtype = torch.bfloat16 if (out_bf16 or x.dtype == torch.bfloat16) else torch.float32
        if residual is not None and residual.dtype != y_dtype:
            residual = residual.to(y_dtype)
        y = torch.empty(R, D, dtype=y_dtype, device=x.device)
        _lib.call("ln_fwd_t", _p(x), _dt(x), _p(src_rows), _p(residual), _p(gamma), _p(beta), _p(y), _dt(y), _p(mean), _p(rstd), R, D,
                  float(eps), p_pre, s_pre, p_post, s_post, _p(seed), _p(add1), int(add1_mod), _p(add2), _p(add2_idx), _stream())
        ctx.save_for_backward(x, gamma, residual, mean, rstd, src_rows, add2_idx, seed)
        ctx.cfg = (R, D, p_pre, s_pre, p_post, s_post, pad_row, add2.shape[0] if add2 is not None else 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, residual, mean, rstd, src_rows, add2_idx, seed = ctx.saved_tensors
        R, D, p_pre, s_pre, p_post, s_post, pad_row, k_add2 = ctx.cfg
        dy = _c(dy)
        dev = dy.device
        need_x, need_res = ctx.needs_input_grad[0], (residual is not None and ctx.needs_input_grad[3])
        dh = dx_rows = None
        same_t = x.dtype == dy.dtype
        if need_res or (need_x and p_pre <= 0.0 and same_t):
            dh = torch.empty(R, D, dtype=dy.dtype, device=dev)
        if need_x:
            dx_rows = dh if (p_pre <= 0.0 and same_t) else torch.empty(R, D, dtype=x.dtype, device=dev)
        g_dir, b_dir, x_dir, a2_dir = ctx.direct
        direct_gb = g_dir is not None and b_dir is not None
        dgamma = g_dir if direct_gb else torch.empty(D, dtype=torch.float32, device=dev)
        dbeta = b_dir if direct_gb else torch.empty(D, dtype=torch.float32, device=dev)
        # rows part on this stream; the dgamma/dbeta tail only feeds the optimizer, so it is forked off (own partial buffer)
        groups = (_lib.load().svpc_ln_param_only_groups(R) if dh is None and dx_rows is None else _lib.load().svpc_ln_bwd_groups(R))
        partial = torch.empty(groups * 2 * D, dtype=torch.float32, device=dev)
        # (strided form: in bf16x3 mode the saved x / residual are the hi planes of split rows, read in place)
        _lib.call("ln_bwd_rows_s", _p(dy), _p(x), _dt(x), x.stride(0), _dt(dy), _p(src_rows), _p(residual),
                  residual.stride(0) if residual is not None else 0, _p(gamma), _p(mean), _p(rstd), _p(dh), _p(dx_rows), _p(partial), R, D,
                  p_pre, s_pre, p_post, s_post, _p(seed), _stream())
        if direct_gb and USE_MULTI_FINALIZE and not SIDE_WGRAD:
            defer_finalize(partial, groups, 2 * D, dgamma, dbeta, D)
            dgamma = dbeta = None
        else:
            with _side_of(g_dir if direct_gb else None, partial):
                _lib.call("ln_param_grads_g", _p(partial), groups, D, _p(dgamma), _p(dbeta), 1 if direct_gb else 0, _stream())
            if direct_gb:
                _ready(dgamma); _ready(dbeta)
                dgamma = dbeta = None
        dx = None
        if need_x:
            if src_rows is not None:
                dx = x_dir if x_dir is not None else torch.zeros_like(x)
                _lib.call("scatter_add_rows", _p(dx_rows), _p(src_rows), _p(dx), R, D, int(pad_row), _stream())
                if x_dir is not None:
                    _ready(dx)
                    dx = None
            else:
                dx = dx_rows
        dadd2 = None
        if k_add2 and ctx.needs_input_grad[4]:
            if a2_dir is not None:
                _colsum(dy, add2_idx, k_add2, out=a2_dir, accumulate=1)
                _ready(a2_dir)
            else:
                dadd2 = _colsum(dy, add2_idx, k_add2)
        dres = dh if need_res else None
        if need_res and ctx.sink and USE_RES_SINK and (dh.dtype == torch.bfloat16 or (_fast() and USE_L32)):
            _RES_SINK[residual.data_ptr()] = dh       # joins the dgrad of the projection that consumes the residual tensor
            SINK_STATS[0] += 1
            _queue_end_of_backward_join()
            dres = None
        return dx, dgamma, dbeta, dres, dadd2, None, None, None, None, None, None, None, None, None, None, None


def layernorm(x, gamma, beta, eps, residual=None, src_rows=None, pad_row=-1, pre_drop=None, post_drop=None,
              add1=None, add1_mod=0, add2=None, add2_idx=None, out_bf16=False, sink=False):
    """sink=True: the residual tensor's only other consumer is an ops.linear whose backward will absorb this LayerNorm's
    residual-path gradient in its dgrad epilogue (see _RES_SINK).  out_bf16: the output joins an activation stream — bf16 in the
    bf16 mode, split (two bf16 planes) in the bf16x3 mode; split inputs always give a split output."""
    require_split_tag(x, "layernorm")
    require_split_tag(residual, "layernorm (residual)")
    out_split = bool(out_bf16) and is_x3()
    split = out_split or lo_off(x) is not None or (residual is not None and lo_off(residual) is not None)
    y = _LayerNorm.apply(x, gamma, beta, residual, add2, eps, src_rows, pad_row, pre_drop, post_drop, add1, add1_mod,
                         add2_idx, out_bf16, sink, out_split)
    if split:
        y._svpc_lo = y.shape[1]
    if ABLATE is not None and "o" in _ablate_flags(gamma):          # (experiments: the normalised rows stored as plain bf16)
        if split:
            _lo_view(y).zero_()
        else:
            y = y.to(torch.bfloat16).float()
    return y


# ------------------------------------------------------------------------------------------------ attention
class _Attention(Function):
    @staticmethod
    def forward(ctx, qt, kvt, cols, D, H, seq, key_mask, causal, drop):
        _need_gpu(qt)
        q_lo, kv_lo = lo_off(qt), lo_off(kvt)
        qt, kvt_c = _rows2d(qt), None
        same = kvt is qt or (kvt.data_ptr() == qt.data_ptr() and kvt.shape == qt.shape)
        kvt_c = qt if same else _rows2d(kvt)
        dh = D // H
        p, site, seed = _drop_args(drop)
        if qt.dtype == torch.float32 and kvt_c.dtype == torch.bfloat16:
            # ONE fp32 query per sequence against bf16 / split keys and values (the [CLS]-only clip-encoder layer): attention_q1s.hip
            # (the wrapper has checked the shape)
            es = 2
            out = torch.empty(seq.n_q_rows, D, dtype=torch.float32, device=qt.device)
            lse = torch.empty(seq.n, H, 1, dtype=torch.float32, device=qt.device)
            tbl = seq.table if seq.table.device == qt.device else seq.table.to(qt.device)
            ev1 = _attn_bracket("fwd", seq.n * H, seq.max_q, seq.max_k, 2 * seq.n_k_rows * D * (4 if kv_lo else 2)) if ATTN_TIMER is not None else None
            _lib.call("attn_q1s_fwd", qt.data_ptr() + cols[0] * 4, qt.stride(0), kvt_c.data_ptr() + cols[1] * es, kvt_c.stride(0), kv_lo or 0,
                      kvt_c.data_ptr() + cols[2] * es, kvt_c.stride(0), kv_lo or 0, _p(out), D, _p(lse), _p(tbl), seq.n, H, dh, seq.max_k,
                      _p(key_mask), 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
            if ev1 is not None:
                ev1.record()
            ctx.save_for_backward(qt, kvt_c, out, lse, key_mask, seed, tbl)
            ctx.cfg = (cols, D, H, seq.n, seq.max_q, seq.max_k, causal, p, site, False)
            ctx.kv_into = getattr(kvt, "_svpc_grad_into", None)
            ctx.mfma, ctx.q1s = False, (kv_lo or 0, True)
            ctx.n_k_rows = seq.n_k_rows
            return out
        ctx.q1s = None
        if kvt_c.dtype != qt.dtype:
            raise _lib.SvpcKernelError("attention: query and key/value tensors must have the same dtype")
        split = q_lo is not None
        if split != (kv_lo is not None):
            raise _lib.SvpcKernelError("attention: query and key/value tensors must both be split or both be plain")
        out = new_split(seq.n_q_rows, D, qt.device) if split else torch.empty(seq.n_q_rows, D, dtype=qt.dtype, device=qt.device)
        lse = torch.empty(seq.n, H, seq.max_q, dtype=torch.float32, device=qt.device)
        tbl = seq.table if seq.table.device == qt.device else seq.table.to(qt.device)
        es = qt.element_size()
        qp, kp, vp = qt.data_ptr() + cols[0] * es, kvt_c.data_ptr() + cols[1] * es, kvt_c.data_ptr() + cols[2] * es
        mfma = (_fast() and ((qp | kp | vp) & 15) == 0 and
                _lib.load().svpc_attn_mfma_supported(dh, seq.max_q, seq.max_k, qt.stride(0), kvt_c.stride(0), kvt_c.stride(0), _dt(qt)) == 1)
        fwd_done = False
        # incremental decoding (one query per sequence, nothing differentiates through it): a wave per (sequence, head), exact fp32.
        # (grad mode is always off INSIDE a Function.forward: what tells inference from training here is requires_grad of the inputs)
        use_q1 = (seq.max_q == 1 and p <= 0.0 and not causal and qt.dtype == torch.float32 and dh <= 64
                  and not (qt.requires_grad or kvt_c.requires_grad))
        # algorithmic bytes of this launch: Q, O rows and K, V rows of every (sequence, head), in their storage types
        ev1 = _attn_bracket("fwd", seq.n * H, seq.max_q, seq.max_k,
                            (2 * seq.n_q_rows + 2 * seq.n_k_rows) * D * (4 if split else es)) if ATTN_TIMER is not None else None
        if split:
            # bf16x3 stream (the wrapper has checked the shape): three-term products over the hi / lo planes; the backward is the
            # bf16 kernel on the hi planes
            if not mfma or (causal and (seq.max_q > 32 or seq.max_k > 32)):
                raise _lib.SvpcKernelError("attention: split tensors need the MFMA shape (head dim 32/64, ≤128 rows; causal: ≤32 rows)")
            _lib.call("attn_x3_fwd", qp, qt.stride(0), q_lo, kp, kvt_c.stride(0), kv_lo, vp, kvt_c.stride(0), kv_lo, _p(out),
                      out.stride(0), out._svpc_lo, _p(lse), _p(tbl), seq.n, H, dh, seq.max_q, seq.max_k, _p(key_mask), 1 if causal else 0,
                      1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
            fwd_done = True
        elif is_x3() and qt.dtype == torch.float32 and not use_q1:
            # bf16x3 mode, fp32 storage (step encoder, decoder, the [CLS]-only layer): exact fp32 forward; the backward still runs
            # on the matrix cores when the shape allows (same LSE definition in both kernel families)
            _lib.call("attn_fwd", qp, qt.stride(0), kp, kvt_c.stride(0), vp, kvt_c.stride(0), _p(out), D, _p(lse), _p(tbl), seq.n, H,
                      dh, seq.max_q, seq.max_k, _p(key_mask), 1 if causal else 0, 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
            fwd_done = True
        if fwd_done:
            pass
        elif use_q1:
            # incremental decoding: one query per sequence — a wave per (sequence, head), no tiles, no LDS
            _lib.call("attn_q1_fwd", qp, qt.stride(0), kp, kvt_c.stride(0), vp, kvt_c.stride(0), _p(out), D, _p(lse), _p(tbl), seq.n, H,
                      dh, seq.max_k, _p(key_mask), 1.0 / math.sqrt(dh), _stream())
            mfma = False
        elif mfma:
            _lib.call("attn_mfma_fwd_t", qp, qt.stride(0), kp, kvt_c.stride(0), vp, kvt_c.stride(0), _p(out), D, _dt(qt), _p(lse),
                      _p(tbl), seq.n, H, dh, seq.max_q, seq.max_k, _p(key_mask), 1 if causal else 0, 1.0 / math.sqrt(dh), p, site,
                      _p(seed), _stream())
        else:
            if qt.dtype != torch.float32:
                raise _lib.SvpcKernelError("attention: the bf16 stream needs the MFMA kernel (head dim 32/64, ≤128 rows)")
            _lib.call("attn_fwd", qp, qt.stride(0), kp, kvt_c.stride(0), vp, kvt_c.stride(0), _p(out), D, _p(lse), _p(tbl), seq.n, H,
                      dh, seq.max_q, seq.max_k, _p(key_mask), 1 if causal else 0, 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
        if ev1 is not None:
            ev1.record()
        ctx.save_for_backward(qt, kvt_c, out, lse, key_mask, seed, tbl)
        ctx.cfg = (cols, D, H, seq.n, seq.max_q, seq.max_k, causal, p, site, same)
        ctx.kv_into = None if same else getattr(kvt, "_svpc_grad_into", None)     # a column block of a shared gradient buffer
        ctx.mfma = mfma and not (ATTN_BWD_EXACT and qt.dtype == torch.float32)
        ctx.n_k_rows = seq.n_k_rows
        return out

    @staticmethod
    def backward(ctx, dO):
        qt, kvt, out, lse, key_mask, seed, tbl = ctx.saved_tensors
        cols, D, H, n, max_q, max_k, causal, p, site, same = ctx.cfg
        dh = D // H
        dO = _c(dO)
        dev = dO.device
        def covered(t, used_cols, used_rows):
            full = t.shape[1] == used_cols and t.shape[0] == used_rows
            mk = torch.empty_like if full else torch.zeros_like
            return mk(t, memory_format=torch.contiguous_format)
        n_q_rows, n_k_rows = out.shape[0], ctx.n_k_rows
        if ctx.q1s is not None:
            kv_lo = ctx.q1s[0]
            dq_t = covered(qt, D, n_q_rows)
            into = ctx.kv_into
            if into is not None and into.shape == kvt.shape and into.dtype == kvt.dtype and kvt.shape == (n_k_rows, 2 * D):
                dkv_t = into
            else:
                dkv_t = covered(kvt, 2 * D, n_k_rows)
            _lib.call("attn_q1s_bwd", qt.data_ptr() + cols[0] * 4, qt.stride(0), kvt.data_ptr() + cols[1] * 2, kvt.stride(0), kv_lo,
                      kvt.data_ptr() + cols[2] * 2, kvt.stride(0), kv_lo, _p(dO), D, dq_t.data_ptr() + cols[0] * 4, dq_t.stride(0),
                      dkv_t.data_ptr() + cols[1] * 2, dkv_t.stride(0), dkv_t.data_ptr() + cols[2] * 2, dkv_t.stride(0), _p(tbl), n, H, dh,
                      max_k, _p(key_mask), 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
            return dq_t, dkv_t, None, None, None, None, None, None, None
        if same:
            dq_t = covered(qt, 3 * D, n_q_rows)
            dkv_t = dq_t
        else:
            dq_t = covered(qt, D, n_q_rows)
            into = ctx.kv_into
            if into is not None and into.shape == kvt.shape and into.dtype == kvt.dtype and kvt.shape == (n_k_rows, 2 * D):
                dkv_t = into                 # every element is written by the kernel below
            else:
                dkv_t = covered(kvt, 2 * D, n_k_rows)
        es = qt.element_size()
        # backward: Q, K, V, O, dO read, dQ, dK, dV written (their storage types: the hi planes in bf16x3 mode)
        ev1 = _attn_bracket("bwd", n * H, max_q, max_k, (3 * n_q_rows + 2 * n_k_rows + n_q_rows + 2 * n_k_rows) * D * es) \
            if ATTN_TIMER is not None else None
        if ctx.mfma:
            _lib.call("attn_mfma_bwd_t", qt.data_ptr() + cols[0] * es, qt.stride(0), kvt.data_ptr() + cols[1] * es, kvt.stride(0),
                      kvt.data_ptr() + cols[2] * es, kvt.stride(0), _p(out), out.stride(0), _dt(qt), _p(lse), _p(dO), D,
                      dq_t.data_ptr() + cols[0] * es, dq_t.stride(0), dkv_t.data_ptr() + cols[1] * es, dkv_t.stride(0),
                      dkv_t.data_ptr() + cols[2] * es, dkv_t.stride(0), _p(tbl), n, H, dh, max_q, max_k, _p(key_mask),
                      1 if causal else 0, 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
            if ev1 is not None:
                ev1.record()
            return dq_t, (None if same else dkv_t), None, None, None, None, None, None, None
        delta = torch.empty(n, H, max_q, dtype=torch.float32, device=dev)
        _lib.call("attn_bwd", qt.data_ptr() + cols[0] * es, qt.stride(0), kvt.data_ptr() + cols[1] * es, kvt.stride(0),
                  kvt.data_ptr() + cols[2] * es, kvt.stride(0), _p(out), D, _p(lse), _p(dO), D,
                  dq_t.data_ptr() + cols[0] * es, dq_t.stride(0), dkv_t.data_ptr() + cols[1] * es, dkv_t.stride(0),
                  dkv_t.data_ptr() + cols[2] * es, dkv_t.stride(0), _p(delta), _p(tbl), n, H, dh, max_q, max_k, _p(key_mask),
                  1 if causal else 0, 1.0 / math.sqrt(dh), p, site, _p(seed), _stream())
        return dq_t, (None if same else dkv_t), None, None, None, None, None, None, None


USE_Q1S = os.environ.get("SVPC_NO_Q1S", "") == ""
USE_P8W = os.environ.get("SVPC_NO_P8W", "") == ""                 # grouped stream wgrads on gemm_p8w.hip (else the round-1 ping-pong kernel)
USE_P8T = os.environ.get("SVPC_NO_P8T", "") == ""                 # stream dgrads on gemm_p8t.hip (else the round-1 ping-pong kernel)
P8T_MIN_TILES = int(os.environ.get("SVPC_P8T_MIN_TILES", "150"))   # 256² tiles needed to fill the chip
USE_S4T = os.environ.get("SVPC_NO_S4T", "") == ""                 # small-M stream dgrads (the decoder) on gemm_s4t.hip (128² tiles)
S4T_MIN_TILES = int(os.environ.get("SVPC_S4T_MIN_TILES", "64"))    # fewer 128² tiles: a long k-loop on a few CUs wants split-K (gemm_glds)
S4T_MAX_TILES = int(os.environ.get("SVPC_S4T_MAX_TILES", "640"))
X3_BIG_TILES = int(os.environ.get("SVPC_X3_BIG_TILES", "128"))      # fewer 256² tiles than this → the 128² split GEMM (gemm_s4x3.hip)


def attention(qt, kvt, cols, D, n_heads, seq, key_mask=None, causal=False, drop=None):
    require_split_tag(qt, "attention (queries)")
    require_split_tag(kvt, "attention (keys / values)")
    if qt.dtype == torch.float32 and kvt.dtype == torch.bfloat16 and kvt is not qt:
        # fp32 queries against bf16 / split keys and values: one query per sequence (the [CLS]-only layer) has its own exact kernels;
        # anything else joins one domain first
        dh = D // n_heads
        lo = lo_off(kvt) or 0
        if (USE_Q1S and _fast() and seq.max_q == 1 and not causal and qt.is_cuda and qt.stride(1) == 1 and kvt.stride(1) == 1 and
                (qt.data_ptr() | kvt.data_ptr()) % 16 == 0 and
                _lib.load().svpc_attn_q1s_supported(dh, seq.max_k, qt.stride(0), kvt.stride(0), kvt.stride(0), lo, lo) == 1):
            return _Attention.apply(qt, kvt, cols, D, n_heads, seq, key_mask, causal, drop)
        if lo_off(kvt) is not None:
            kvt = to_f32(kvt)
        else:
            qt = qt.to(torch.bfloat16)
    split = lo_off(qt) is not None or lo_off(kvt) is not None
    if ABLATE is not None and split and kvt is qt and ABLATE.get("attn"):      # (experiments: Q / K or V operands of the clip encoder's core
        flags = ABLATE["attn"] if seq.max_q > 32 else ""                      # without their lo planes)
        if flags:
            assert not torch.is_grad_enabled()
            c = new_split(qt.shape[0], qt.shape[1], qt.device)
            c.copy_(qt.detach()); _lo_view(c).copy_(_lo_view(qt))
            lo = _lo_view(c)
            if "q" in flags:
                lo[:, cols[0]:cols[0] + D].zero_()
            if "k" in flags:
                lo[:, cols[1]:cols[1] + D].zero_()
            if "v" in flags:
                lo[:, cols[2]:cols[2] + D].zero_()
            qt = kvt = c
    if split:
        dh = D // n_heads
        # (the same conditions _Attention.forward demands of a split stream, 16-byte aligned column blocks included: what fails them
        # leaves the split domain here instead of raising there)
        ok = ((not causal or (seq.max_q <= 32 and seq.max_k <= 32)) and lo_off(qt) is not None and lo_off(kvt) is not None and
              _lib.load().svpc_attn_mfma_supported(dh, seq.max_q, seq.max_k, 8, 8, 8, 1) == 1 and qt.stride(0) % 8 == 0 and kvt.stride(0) % 8 == 0 and
              ((qt.data_ptr() + cols[0] * 2) | (kvt.data_ptr() + cols[1] * 2) | (kvt.data_ptr() + cols[2] * 2)) % 16 == 0)
        if not ok:          # leave the split domain: exact fp32 attention on fp32 copies
            same = kvt is qt
            qt = to_f32(qt)
            kvt = qt if same else to_f32(kvt)
            split = False
    out = _Attention.apply(qt, kvt, cols, D, n_heads, seq, key_mask, causal, drop)
    if split:
        out._svpc_lo = out.shape[1]
    return out


class _SplitCols(Function):
    """One wide projection output (R, n·w) handed out as n column blocks (views, row stride n·w).  The gradient of the wide
    tensor is ONE buffer allocated up front: a consumer that finds ``_svpc_grad_into`` on its block (ops.attention does) writes
    its gradient straight into that block's columns and returns the view, so backward neither concatenates nor adds."""

    @staticmethod
    def forward(ctx, wide, n, gbuf):
        w = wide.shape[1] // n
        ctx.gbuf, ctx.n, ctx.w = gbuf, n, w
        return tuple(wide[:, i * w:(i + 1) * w] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        g, w = ctx.gbuf, ctx.w
        for i, gi in enumerate(grads):
            blk = g[:, i * w:(i + 1) * w]
            if gi is None:
                blk.zero_()
            elif gi.data_ptr() != blk.data_ptr() or gi.stride() != blk.stride():
                blk.copy_(gi)
        ctx.gbuf = None
        return g, None, None


def attn_q1_ln(q, kv, k_stride, n_keys, residual, gamma, beta, eps, n_heads, new_kv=None):
    """One decoding step of an attention block (reference decoder layer model.py:620-663 for one new position per sentence,
    translator.py:88-112): LayerNorm(residual + Attention(q; the sentence's key / value rows)), one launch.  ``q``: (T, ≥D) fp32, the
    query in its first D columns; ``kv``: (rows, 2D) fp32, K | V, sentence t owning rows t·k_stride … t·k_stride + n_keys − 1;
    ``new_kv`` (T, 2D view): the token's own K | V, stored as the sentence's row n_keys − 1 first (the cache append).  Forward only.
    Returns None when the shape is not taken (the caller then runs copy / attention / layernorm separately)."""
    T, D = residual.shape
    if torch.is_grad_enabled() and (q.requires_grad or residual.requires_grad):
        return None
    ts = (q, kv, residual) + ((new_kv,) if new_kv is not None else ())
    if not all(t.is_cuda and t.dtype == torch.float32 and lo_off(t) is None and t.stride(1) == 1 for t in ts) or D % n_heads:
        return None
    lib = _lib.load()
    ldn = new_kv.stride(0) if new_kv is not None else 0
    if lib.svpc_attn_q1_ln_supported(D, D // n_heads, n_keys, q.stride(0), kv.stride(0), ldn, residual.stride(0), D) != 1:
        return None
    if any(t.data_ptr() % 16 for t in ts) or kv.shape[1] != 2 * D or k_stride < n_keys:
        return None
    out = torch.empty(T, D, dtype=torch.float32, device=q.device)
    nk = new_kv.data_ptr() if new_kv is not None else None
    _lib.call("attn_q1_ln_fwd", _p(q), q.stride(0), kv.data_ptr(), kv.data_ptr() + 4 * D, kv.stride(0), k_stride, n_keys, nk,
              (nk + 4 * D) if nk is not None else None, ldn, _p(residual), residual.stride(0), _p(gamma), _p(beta), float(eps), _p(out), D, T, D,
              D // n_heads, 1.0 / math.sqrt(D // n_heads), _stream())
    return out


def split_cols(wide, n):
    """→ n column blocks of ``wide`` (R, n·w) whose gradients are gathered in place (see _SplitCols).  A split ``wide`` gives split
    blocks: the lo plane of a block lies the same number of columns behind it as the lo plane of the whole row."""
    lo = lo_off(wide)
    w = wide.shape[1] // n
    if not (torch.is_grad_enabled() and wide.requires_grad):
        outs = tuple(wide[:, i * w:(i + 1) * w] for i in range(n))
    else:
        gbuf = torch.empty(wide.shape, dtype=wide.dtype, device=wide.device)
        outs = _SplitCols.apply(wide, n, gbuf)
        for i, o in enumerate(outs):
            o._svpc_grad_into = gbuf[:, i * w:(i + 1) * w]
    if lo is not None:
        for o in outs:
            o._svpc_lo = lo
    return outs


def to_split(t):
    """split copy (two bf16 planes) of an fp32 2-D tensor joining a bf16x3 stream (a handful of rows: the decoder's memory slots);
    the gradient comes back as the dense bf16 tensor every split tensor has"""
    y = _ToSplit.apply(t)
    y._svpc_lo = t.shape[1]
    return y


class _ToSplit(Function):
    @staticmethod
    def forward(ctx, t):
        out = new_split(t.shape[0], t.shape[1], t.device)
        if t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32:
            _rows_move(t, None, None, out, out._svpc_lo, t.shape[0], t.shape[1])
            return out
        out.copy_(t)
        torch.as_strided(out, out.shape, out.stride(), out.storage_offset() + out._svpc_lo).copy_(t - out.float())
        return out

    @staticmethod
    def backward(ctx, g):
        return g.float()


# ------------------------------------------------------------------------------------------------ decoder cross-attention, fused
# Cross-attention of every sentence row to the ≤ 3 memory rows of its sentence + residual + LayerNorm in ONE launch, forward and backward
# (svpc_amd/csrc/cross_attn.hip; reference model.py:657-658 in :630-663; SURVEY §2.3 K6 "trivial; fuse").
USE_XATTN = os.environ.get("SVPC_NO_XATTN", "") == ""


def cross_attn_ln_usable(D, H, lt, nm, mem_mask=None):
    return (USE_XATTN and mem_mask is None and H > 0 and D % H == 0
            and _lib.load().svpc_cross_attn_ln_supported(int(D), int(H), int(lt), int(nm)) == 1)


def _kind(t):
    """storage code of the C-ABI: 0 fp32, 1 bf16, 2 split"""
    return 2 if lo_off(t) is not None else _dt(t)


class _CrossAttnLn(Function):
    @staticmethod
    def forward(ctx, q, x1, kv, gamma, beta, eps, H, lt, nm, drop, sink, rows=None):
        _need_gpu(q)
        D = gamma.shape[0]
        R = q.shape[0]
        T = R // lt if rows is None else rows[0].numel()       # rows = (row_off, row_len) int32 device tensors: ragged sentences
        ro, rl = rows if rows is not None else (None, None)
        dev = q.device
        out_kind = _kind(x1)
        if out_kind == 2:
            y = new_split(R, D, dev)
            loy = y._svpc_lo
        else:
            y = torch.empty(R, D, dtype=x1.dtype, device=dev)
            loy = 0
        probs = torch.empty(R, H, 4, dtype=torch.float32, device=dev)
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        p, site, seed = _drop_args(drop)
        scale = 1.0 / math.sqrt(D // H)
        es = kv.element_size()
        _lib.call("cross_attn_ln_fwd_r", _p(q), _kind(q), q.stride(0), lo_off(q) or 0, _p(x1), out_kind, x1.stride(0), lo_off(x1) or 0,
                  kv.data_ptr(), kv.data_ptr() + D * es, _kind(kv), kv.stride(0), lo_off(kv) or 0, _p(gamma), _p(beta), float(eps),
                  _p(y), out_kind, y.stride(0), loy, _p(probs), _p(mean), _p(rstd), T, lt, nm, D, H, scale, p, site, _p(seed), _p(ro), _p(rl),
                  _stream())
        ctx.save_for_backward(q, x1, kv, gamma, probs, mean, rstd, seed, ro, rl)
        ctx.cfg = (T, lt, nm, D, H, scale, p, site, _kind(q), lo_off(q) or 0, out_kind, lo_off(x1) or 0, _kind(kv), lo_off(kv) or 0)
        ctx.kv_into = getattr(kv, "_svpc_grad_into", None)
        ctx.direct = (_direct(gamma), _direct(beta))
        ctx.sink = bool(sink)
        return y

    @staticmethod
    def backward(ctx, dy):
        q, x1, kv, gamma, probs, mean, rstd, seed, ro, rl = ctx.saved_tensors
        T, lt, nm, D, H, scale, p, site, q_dt, q_lo, x_dt, x_lo, kv_dt, kv_lo = ctx.cfg
        dev = dy.device
        dy = _c(dy)
        R = q.shape[0]
        dq = torch.empty(R, D, dtype=dy.dtype, device=dev)
        dres = torch.empty(R, D, dtype=dy.dtype, device=dev)
        into = ctx.kv_into
        if into is not None and into.shape == kv.shape and into.dtype == kv.dtype:
            dkv = into                                 # the layer's block of the stacked projection's gradient buffer: written in place
        else:
            dkv = torch.empty(kv.shape, dtype=kv.dtype, device=dev)
        part_ln = torch.empty(T, 2 * D, dtype=torch.float32, device=dev)
        es = dkv.element_size()
        _lib.call("cross_attn_ln_bwd_r", _p(q), q_dt, q.stride(0), q_lo, _p(x1), x_dt, x1.stride(0), x_lo, kv.data_ptr(),
                  kv.data_ptr() + D * kv.element_size(), kv_dt, kv.stride(0), kv_lo, _p(gamma), _p(probs), _p(mean), _p(rstd), _p(dy), _dt(dy),
                  dy.stride(0), _p(dq), _p(dres), _dt(dq), dq.stride(0), dkv.data_ptr(), dkv.data_ptr() + D * es, _dt(dkv), dkv.stride(0),
                  _p(part_ln), T, lt, nm, D, H, scale, p, site, _p(seed), _p(ro), _p(rl), _stream())
        g_d, b_d = ctx.direct
        dgamma = dbeta = None
        if g_d is not None and b_d is not None and USE_MULTI_FINALIZE and not SIDE_WGRAD:
            defer_finalize(part_ln, T, 2 * D, g_d, b_d, D)
        else:
            sl = _colsum(part_ln).view(-1)
            dgamma, dbeta = sl[:D].clone(), sl[D:].clone()
            if g_d is not None:
                g_d.add_(dgamma); dgamma = None
            if b_d is not None:
                b_d.add_(dbeta); dbeta = None
        if ctx.sink and USE_RES_SINK and (dres.dtype == torch.bfloat16 or (_fast() and USE_L32)):
            _RES_SINK[x1.data_ptr()] = dres            # joins the dgrad of the query projection, the other consumer of x1 (see _RES_SINK)
            SINK_STATS[0] += 1
            _queue_end_of_backward_join()
            dres = None
        return dq, dres, dkv, dgamma, dbeta, None, None, None, None, None, None, None


def cross_attn_ln(q, x1, kv, gamma, beta, eps, n_heads, lt, nm, drop=None, sink=False, rows=None):
    """LayerNorm(x1 + CrossAttention(q; the sentence's nm memory rows)) in one launch — q, x1 (T·lt, D) fp32 / bf16 / split; kv (T·nm, 2D)
    [K | V] rows of the same storage family (may be a column block of the stacked memory projection).  Output in x1's storage kind.
    rows = (row_off, row_len): ragged sentences (valid tokens only) — sentence s owns rows [row_off[s], row_off[s] + row_len[s]), ≤ lt each.
    sink=True: x1's only other consumer is the ops.linear that produced q — its dgrad absorbs the residual-path gradient (_RES_SINK)."""
    require_split_tag(q, "cross_attn_ln (queries)")
    require_split_tag(x1, "cross_attn_ln (residual)")
    require_split_tag(kv, "cross_attn_ln (memory rows)")
    y = _CrossAttnLn.apply(q, x1, kv, gamma, beta, float(eps), int(n_heads), int(lt), int(nm), drop, sink, rows)
    if lo_off(x1) is not None:
        y._svpc_lo = y.shape[1]
    return y


# ------------------------------------------------------------------------------------------------ spans / rows
class _SpanMean(Function):
    @staticmethod
    def forward(ctx, x, starts, lens, weights, add, add_idx):
        _need_gpu(x)
        x = _c(x)
        G, D = len(starts), x.shape[1]
        dev = x.device
        out = torch.empty(G, D, dtype=torch.float32, device=dev)
        st, ln = starts.dev(dev), lens.dev(dev)
        ai = add_idx.dev(dev) if add is not None else None
        _lib.call("span_mean_fwd", _p(x), _p(st), _p(ln), _p(weights), _p(add), _p(ai), _p(out), G, D, _stream())
        ctx.save_for_backward(st, ln, weights)
        ctx.cfg = (tuple(x.shape), G, D)
        ctx.tiles = _spans_tile(starts, lens, x.shape[0])       # the spans cover every row exactly once: the backward writes all of dx
        return out

    @staticmethod
    def backward(ctx, dout):
        st, ln, weights = ctx.saved_tensors
        shape, G, D = ctx.cfg
        dout = _c(dout)
        dx = (torch.empty if ctx.tiles else torch.zeros)(shape, dtype=torch.float32, device=dout.device)
        _lib.call("span_mean_bwd", _p(dout), _p(st), _p(ln), _p(weights), _p(dx), G, D, _stream())
        return dx, None, None, None, None, None


_TILES = {}


def _spans_tile(starts, lens, n_rows):
    """whether the spans (host lists of two ``Idx``) are consecutive and cover rows 0..n_rows-1 exactly; cached per Idx pair"""
    key = (id(starts), id(lens), n_rows)
    hit = _TILES.get(key)
    if hit is not None and hit[0] is starts and hit[1] is lens:
        return hit[2]
    pos, ok = 0, True
    for s_, l_ in zip(starts.host, lens.host):
        if s_ != pos or l_ < 1:
            ok = False
            break
        pos += l_
    ok = ok and pos == n_rows
    if len(_TILES) > 256:
        _TILES.clear()
    _TILES[key] = (starts, lens, ok)
    return ok


def span_mean(x, starts, lens, weights=None, add=None, add_idx=None):
    return _SpanMean.apply(x, as_idx(starts), as_idx(lens), weights, add, as_idx(add_idx) if add_idx is not None else None)


class _RowNorm(Function):
    @staticmethod
    def forward(ctx, a, mode):
        _need_gpu(a)
        a = _c(a)
        y = torch.empty_like(a)
        _lib.call("rownorm_fwd", _p(a), _p(y), a.shape[0], a.shape[1], mode, _stream())
        ctx.save_for_backward(a, y)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, dy):
        a, y = ctx.saved_tensors
        dy = _c(dy)
        da = torch.empty_like(a)
        _lib.call("rownorm_bwd", _p(dy), _p(y), _p(a), _p(da), a.shape[0], a.shape[1], ctx.mode, _stream())
        return da, None


def row_normalize(a):
    return _RowNorm.apply(a, 0)


def softmax_rows(x):
    return _RowNorm.apply(x, 1)


class _SumAll(Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = _c(x)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.call("sum_all", _p(x), x.numel(), _p(out), 1.0, _stream())
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        _lib.call("fill_from", _p(dx), dx.numel(), _p(_c(dout)), _stream())
        return dx


def sum_all(x):
    return _SumAll.apply(x)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        c = torch.empty_like(a)
        _lib.call("add", _p(a), _p(b), _p(c), a.numel(), _stream())
        return c

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return _Add.apply(a, b)


def take_rows(x, idx):
    """Row gather (data movement only)."""
    return torch.index_select(x, 0, idx)


class _GcSeg(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("src_dt", ctypes.c_int),
                ("dst_dt", ctypes.c_int), ("n", ctypes.c_int)]


_GC_CODE = {torch.float32: 0, torch.int64: 1, torch.int32: 2}


def gather_cast_multi(items, rng=None):
    """items: [(src 1-D tensor, int32 row index or None, torch.float32 | torch.int32)] → [dst]; dst[i] = cast(src[idx[i]]) for all
    items in ONE launch (data movement only, no gradient) — the token staging of the batched forward.  rng: advance its seed in the
    same launch (``rng.begin_step(defer=True)`` was called)."""
    outs, segs = [], []
    for src, idx, dt in items:
        src = _c(src.reshape(-1))
        if src.dtype not in _GC_CODE:
            src = src.to(torch.float32 if src.dtype.is_floating_point else torch.int64)
        n = idx.numel() if idx is not None else src.numel()
        dst = torch.empty(n, dtype=dt, device=src.device)
        outs.append(dst)
        segs.append((src, idx, dst))
    for k in range(0, len(segs), 8):
        part = segs[k:k + 8]
        arr = (_GcSeg * len(part))()
        for i, (src, idx, dst) in enumerate(part):
            arr[i] = _GcSeg(src.data_ptr(), idx.data_ptr() if idx is not None else None, dst.data_ptr(), _GC_CODE[src.dtype],
                            _GC_CODE[dst.dtype], dst.numel())
        _lib.call("gather_cast_multi_seed", ctypes.addressof(arr), len(part), _p(rng.seed) if (rng is not None and k == 0) else None, _stream())
    if rng is not None and not segs:
        _lib.call("bump_seed", _p(rng.seed), _stream())
    return outs


def clamp_labels(labels, vocab, unk):
    out = torch.empty_like(labels)
    _lib.call("clamp_labels", _p(labels), _p(out), labels.numel(), int(vocab), int(unk), _stream())
    return out


def row_any_eq1(x):
    x = _c(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.call("row_any_eq1", _p(x), _p(out), x.shape[0], x.shape[1], _stream())
    return out


# ------------------------------------------------------------------------------------------------ simulator recurrence
class _SimRecur(Function):
    @staticmethod
    def forward(ctx, q, c, w4f, E0, step_off, step_len, ent_off, ent_len, e_max):
        _need_gpu(q)
        q, c, w4f, E0 = _c(q), _c(c), _c(w4f), _c(E0)
        T, D = q.shape
        dev = q.device
        idx = [v.dev(dev) for v in (step_off, step_len, ent_off, ent_len)]
        e = torch.empty(T, e_max, dtype=torch.float32, device=dev)
        ebar = torch.empty(T, D, dtype=torch.float32, device=dev)
        eall = torch.empty(T, e_max, D, dtype=torch.float32, device=dev)
        _lib.call("sim_recur_fwd", _p(q), _p(c), _p(w4f), _p(E0), _p(idx[0]), _p(idx[1]), _p(idx[2]), _p(idx[3]), len(step_off), e_max,
                  D, _p(e), _p(ebar), _p(eall), _stream())
        ctx.save_for_backward(q, c, w4f, E0, e, ebar, eall, *idx)
        ctx.cfg = (len(step_off), e_max, D)
        ctx.set_materialize_grads(False)
        return e, ebar, eall

    @staticmethod
    def backward(ctx, de, debar, deall):
        q, c, w4f, E0, e, ebar, eall, so, sl, eo, el = ctx.saved_tensors
        n, e_max, D = ctx.cfg
        dev = q.device
        de = _c(de) if de is not None else None
        debar = _c(debar) if debar is not None else None
        deall = _c(deall) if deall is not None else None
        dq = torch.empty_like(q)
        dc = torch.empty_like(c)
        dw = torch.empty_like(w4f)
        dE0 = torch.empty_like(E0)
        _lib.call("sim_recur_bwd", _p(q), _p(c), _p(w4f), _p(E0), _p(so), _p(sl), _p(eo), _p(el), n, e_max, D, _p(e), _p(ebar),
                  _p(eall), _p(de), _p(debar), _p(deall), _p(dq), _p(dc), _p(dw), _p(dE0), _stream())
        return dq, dc, dw, dE0, None, None, None, None, None


def sim_recur(q, c, w4f, E0, step_off, step_len, ent_off, ent_len, e_max):
    return _SimRecur.apply(q, c, w4f, E0, as_idx(step_off), as_idx(step_len), as_idx(ent_off), as_idx(ent_len), int(e_max))


class _SimHeads(Function):
    """c = softmax(ĥ·W3ᵀ + b3), w = f̄·W4ᵀ + b4 (reference model.py:801, :804-805) in one launch; backward in one launch (svpc_sim_heads_*)."""

    @staticmethod
    def forward(ctx, hh, fb, W3, b3, W4, b4):
        _need_gpu(hh)
        hh, fb, W3, W4 = _c(hh), _c(fb), _c(W3), _c(W4)
        T, D = hh.shape
        Wd = fb.shape[1]
        c = torch.empty(T, 3, dtype=torch.float32, device=hh.device)
        w = torch.empty(T, dtype=torch.float32, device=hh.device)
        _lib.call("sim_heads_fwd", _p(hh), _p(fb), _p(W3), _p(b3), _p(W4), _p(b4), _p(c), _p(w), T, D, Wd, _stream())
        ctx.save_for_backward(hh, fb, W3, W4, c)
        ctx.direct = (_direct(W3), _direct(b3), _direct(W4), _direct(b4))
        ctx.set_materialize_grads(False)
        return c, w

    @staticmethod
    def backward(ctx, dc, dw):
        hh, fb, W3, W4, c = ctx.saved_tensors
        T, D = hh.shape
        Wd = fb.shape[1]
        dev = hh.device
        dc = _c(dc) if dc is not None else None
        dw = _c(dw) if dw is not None else None
        G = _lib.load().svpc_sim_heads_groups(T)
        dhh, dfb = torch.empty_like(hh), torch.empty_like(fb)
        p3 = torch.empty(G, 3 * D + 3, dtype=torch.float32, device=dev)
        p4 = torch.empty(G, Wd + 1, dtype=torch.float32, device=dev)
        _lib.call("sim_heads_bwd", _p(hh), _p(fb), _p(W3), _p(W4), _p(c), _p(dc), _p(dw), _p(dhh), _p(dfb), _p(p3), _p(p4), T, D, Wd, _stream())
        w3d, b3d, w4d, b4d = ctx.direct
        if all(t is not None for t in ctx.direct) and USE_MULTI_FINALIZE and not SIDE_WGRAD:
            defer_finalize(p3, G, 3 * D + 3, w3d, b3d, 3 * D)
            defer_finalize(p4, G, Wd + 1, w4d, b4d, Wd)
            return dhh, dfb, None, None, None, None
        s3, s4 = _colsum(p3).view(-1), _colsum(p4).view(-1)
        outs = [s3[:3 * D].reshape(3, D), s3[3 * D:].clone(), s4[:Wd].reshape(1, Wd), s4[Wd:].clone()]
        res = []
        for tgt, val in zip(ctx.direct, outs):
            if tgt is not None:
                tgt.add_(val.view_as(tgt)); res.append(None)
            else:
                res.append(val)
        return (dhh, dfb) + tuple(res)


def sim_heads(hh, fb, W3, b3, W4, b4):
    """(c (T, 3), w (T,)) — the simulator's choice softmax and verb scalar; None when the shape is not taken"""
    if (not hh.is_cuda or hh.dtype != torch.float32 or fb.dtype != torch.float32 or tuple(W3.shape) != (3, hh.shape[1]) or
            tuple(W4.shape) != (1, fb.shape[1]) or hh.shape[1] > 1024 or fb.shape[1] > 512 or b3 is None or b4 is None):
        return None
    return _SimHeads.apply(hh, fb, W3, b3, W4, b4)


# ------------------------------------------------------------------------------------------------ pointer-generator
class _PtrAttn(Function):
    @staticmethod
    def forward(ctx, dec, proj, bank, step_ne, lt):
        _need_gpu(dec)
        dec, proj, bank = _c(dec), _c(proj), _c(bank)
        T, e_max, D = bank.shape
        dev = dec.device
        ne = step_ne.dev(dev)
        pi = torch.empty(T * lt, e_max, dtype=torch.float32, device=dev)
        att = torch.empty(T * lt, D, dtype=torch.float32, device=dev)
        _lib.call("ptr_attn_fwd", _p(dec), _p(proj), _p(bank), _p(ne), _p(pi), _p(att), T, lt, e_max, D, _stream())
        ctx.save_for_backward(dec, proj, bank, ne, pi)
        ctx.cfg = (T, lt, e_max, D)
        ctx.set_materialize_grads(False)
        return pi, att

    @staticmethod
    def backward(ctx, dpi, datt):
        dec, proj, bank, ne, pi = ctx.saved_tensors
        T, lt, e_max, D = ctx.cfg
        if datt is None:
            datt = torch.zeros(T * lt, D, dtype=torch.float32, device=dec.device)
        datt = _c(datt)
        dpi = _c(dpi) if dpi is not None else None
        ddec = torch.empty_like(dec)
        dproj = torch.empty_like(proj)
        dbank = torch.empty_like(bank)
        _lib.call("ptr_attn_bwd", _p(dec), _p(proj), _p(bank), _p(ne), _p(pi), _p(dpi), _p(datt), _p(ddec), _p(dproj), _p(dbank), T, lt,
                  e_max, D, _stream())
        return ddec, dproj, dbank, None, None


def ptr_attn(dec, proj, bank, step_ne, lt):
    return _PtrAttn.apply(dec, proj, bank, as_idx(step_ne), int(lt))


class _PtrAttnGate(Function):
    """Pointer attention + generation gate of every row in one launch, forward and backward (svpc_ptr_attn_gate_fwd/bwd): the attended
    vector feeds only the gate (model.py:903-908), so neither it nor the [dec ; att] concatenation nor the gate's 1,536-deep one-column
    projection (with its dgrad / wgrad / bias-sum launches) exist."""

    @staticmethod
    def forward(ctx, dec, proj, bank, w, b, step_ne, lt, rows=None):
        _need_gpu(dec)
        dec, proj, bank, w = _c(dec), _c(proj), _c(bank), _c(w)
        T, e_max, D = bank.shape
        dev = dec.device
        ne = step_ne.dev(dev)
        ro, rl = rows if rows is not None else (None, None)      # ragged sentences: (row_off, row_len) int32 device tensors
        R = dec.shape[0]
        pi = torch.empty(R, e_max, dtype=torch.float32, device=dev)
        g = torch.empty(R, 1, dtype=torch.float32, device=dev)
        _lib.call("ptr_attn_gate_fwd_r", _p(dec), _p(proj), _p(bank), _p(ne), _p(pi), _p(w), _p(b), _p(g), T, lt, e_max, D, _p(ro), _p(rl),
                  _stream())
        ctx.save_for_backward(dec, proj, bank, w, ne, pi, g, ro, rl)
        ctx.cfg = (T, lt, e_max, D)
        ctx.direct = (_direct(w), _direct(b))
        ctx.set_materialize_grads(False)
        return pi, g

    @staticmethod
    def backward(ctx, dpi, dg):
        dec, proj, bank, w, ne, pi, g, ro, rl = ctx.saved_tensors
        T, lt, e_max, D = ctx.cfg
        dev = dec.device
        dpi = _c(dpi) if dpi is not None else None
        dg = _c(dg) if dg is not None else None
        ddec, dproj, dbank = torch.empty_like(dec), torch.empty_like(proj), torch.empty_like(bank)
        wpart = torch.empty(T, 2 * D + 1, dtype=torch.float32, device=dev)
        _lib.call("ptr_attn_gate_bwd_r", _p(dec), _p(proj), _p(bank), _p(ne), _p(pi), _p(dpi), _p(g), _p(dg), _p(w), _p(ddec), _p(dproj),
                  _p(dbank), _p(wpart), T, lt, e_max, D, _p(ro), _p(rl), _stream())
        wg, bg = ctx.direct
        dw = db = None
        if wg is not None and bg is not None and USE_MULTI_FINALIZE and not SIDE_WGRAD:
            defer_finalize(wpart, T, 2 * D + 1, wg, bg, 2 * D)       # (column sums of the per-step partials, with every other pending tail)
        else:
            sums = _colsum(wpart).view(-1)
            dw, db = sums[:2 * D].reshape(1, 2 * D), sums[2 * D:]
            if wg is not None:
                wg.add_(dw); dw = None
            if bg is not None:
                bg.add_(db); db = None
        return ddec, dproj, dbank, dw, db, None, None, None


def ptr_attn_gate(dec, proj, bank, step_ne, lt, w, b, rows=None):
    """(pi (T·lt, e_max), p_gen (T·lt, 1)) with p_gen = sigmoid([dec ; att]·wᵀ + b), att = Σ_e pi·bank — training form (differentiable
    in dec, proj, bank, w, b); None when the shape is not taken (the caller then uses ptr_attn + linear).  rows = (row_off, row_len): ragged
    sentences (valid tokens only) — sentence j owns the dec rows [row_off[j], row_off[j] + row_len[j]), ≤ lt each."""
    T, e_max, D = bank.shape
    if (not dec.is_cuda or dec.dtype != torch.float32 or D % 4 or tuple(w.shape) != (1, 2 * D) or b is None or b.numel() != 1
            or lt > 32 or e_max > 32 or (rows is None and dec.shape != (T * lt, D)) or dec.shape[1] != D or w.dtype != torch.float32
            or (w.data_ptr() % 16) or lo_off(dec) is not None):
        return None
    return _PtrAttnGate.apply(dec, proj, bank, w, b, as_idx(step_ne), int(lt), rows)


def ptr_attn_pgen(dec, proj, bank, step_ne, w, b):
    """Decoding iteration (one position per step row, no gradients): (pi, p_gen) with p_gen = sigmoid([dec ; att]·w + b) computed inside
    the pointer-attention launch — replaces ptr_attn + concatenation + the 1,536-deep one-column projection.  None if not applicable."""
    T, e_max, D = bank.shape
    if (torch.is_grad_enabled() or not dec.is_cuda or dec.dtype != torch.float32 or D > 768 or D % 4 or w.shape != (1, 2 * D)
            or dec.shape != (T, D) or not (dec.is_contiguous() and proj.is_contiguous() and bank.is_contiguous() and w.is_contiguous())):
        return None
    dev = dec.device
    pi = torch.empty(T, e_max, dtype=torch.float32, device=dev)
    g = torch.empty(T, 1, dtype=torch.float32, device=dev)
    _lib.call("ptr_attn_pgen_fwd", _p(dec), _p(proj), _p(bank), _p(as_idx(step_ne).dev(dev)), _p(pi), None, _p(w), _p(b), _p(g), T, 1, e_max, D,
              _stream())
    return pi, g


class _PtrMixLoss(Function):
    @staticmethod
    def forward(ctx, logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, smoothing):
        _need_gpu(logits)
        logits = _c(logits)
        R, V = logits.shape
        dev = logits.device
        g = _c(g) if g is not None else None
        pi = _c(pi) if pi is not None else None
        e_max = pi.shape[1] if pi is not None else 0
        rc, rv = row_c.dev(dev), row_vid.dev(dev)
        co, ce, ci, cw = csr_off.dev(dev), csr_ent.dev(dev), csr_id.dev(dev), csr_w.dev(dev)
        P = torch.empty(R, c_max, dtype=torch.float32, device=dev)
        loss_rows = torch.empty(R, dtype=torch.float32, device=dev)
        row_w = None
        if not smoothing > 0:
            # label_smoothing == 0 (reference model.py:869-870): cross-entropy of the probabilities, a mean per video → per-row weights
            row_w = torch.empty(R, dtype=torch.float32, device=dev)
            _lib.call("ce_row_weights", _p(labels), _p(rv), R, len(csr_off) - 1, _p(row_w), _stream())
        _lib.call("ptr_mix_ce_fwd", _p(logits), _p(g), _p(pi), _p(labels), _p(rc), _p(rv), _p(co), _p(ce), _p(ci), _p(cw), _p(P),
                  _p(loss_rows), R, V, c_max, e_max, float(smoothing), _p(row_w), _stream())
        ctx.save_for_backward(logits, g, pi, labels, rc, rv, co, ce, ci, cw, P, row_w)
        ctx.cfg = (R, V, c_max, e_max, float(smoothing))
        ctx.set_materialize_grads(False)
        return P, loss_rows

    @staticmethod
    def backward(ctx, dP, dloss):
        logits, g, pi, labels, rc, rv, co, ce, ci, cw, P, row_w = ctx.saved_tensors
        R, V, c_max, e_max, smoothing = ctx.cfg
        dev = logits.device
        dP = _c(dP) if dP is not None else None
        if dloss is None:
            dloss = torch.zeros(R, dtype=torch.float32, device=dev)
        dloss = _c(dloss)
        dlogits = torch.empty_like(logits)
        dg = torch.empty_like(g) if g is not None else None
        dpi = torch.empty_like(pi) if pi is not None else None
        _lib.call("ptr_mix_ce_bwd", _p(logits), _p(g), _p(pi), _p(labels), _p(rc), _p(rv), _p(co), _p(ce), _p(ci), _p(cw), _p(P),
                  _p(dP), _p(dloss), _p(dlogits), _p(dg), _p(dpi), R, V, c_max, e_max, smoothing, _p(row_w), _stream())
        return dlogits, dg, dpi, None, None, None, None, None, None, None, None, None


def ptr_mix_loss(logits, g, pi, labels, row_c, row_vid, csr_off, csr_ent, csr_id, csr_w, c_max, smoothing):
    return _PtrMixLoss.apply(logits, g, pi, labels, as_idx(row_c), as_idx(row_vid), as_idx(csr_off), as_idx(csr_ent),
                             as_idx(csr_id), csr_w if isinstance(csr_w, FIdx) else FIdx(csr_w), int(c_max), float(smoothing))


# ------------------------------------------------------------------------------------------------ Gumbel bag of words
class _GumbelBow(Function):
    @staticmethod
    def forward(ctx, P, emb, row_c, tau, noise):
        _need_gpu(P)
        ctx.emb_direct = _direct(emb)
        P, emb, noise = _c(P), _c(emb), _c(noise)
        R, c_max = P.shape
        V, W = emb.shape
        dev = P.device
        rc = row_c.dev(dev)
        bow = torch.empty(R, W, dtype=torch.float32, device=dev)
        idx = torch.empty(R, dtype=torch.int32, device=dev)
        stats = torch.empty(R, 2, dtype=torch.float32, device=dev)
        _lib.call("gumbel_fwd", _p(P), _p(noise), _p(rc), _p(emb), _p(bow), _p(idx), _p(stats), R, c_max, V, W, float(tau), _stream())
        ctx.save_for_backward(P, emb, noise, rc, idx, stats)
        ctx.cfg = (R, c_max, V, W, float(tau))
        return bow

    @staticmethod
    def backward(ctx, dbow):
        P, emb, noise, rc, idx, stats = ctx.saved_tensors
        R, c_max, V, W, tau = ctx.cfg
        dev = P.device
        dbow = _c(dbow)
        dP = demb = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty(R, V, dtype=torch.float32, device=dev)
            _gemm(dbow, W, 1, emb, W, 1, dy, R, V, W)                      # dy = dbow @ embᵀ
            dP = torch.empty_like(P)
            _lib.call("gumbel_bwd", _p(P), _p(noise), _p(rc), _p(stats), _p(dy), _p(dP), R, c_max, V, tau, _stream())
        if ctx.needs_input_grad[1]:
            demb = ctx.emb_direct if ctx.emb_direct is not None else torch.zeros_like(emb)
            _lib.call("gumbel_emb_grad", _p(dbow), _p(idx), _p(stats), _p(demb), R, V, W, _stream())
            if ctx.emb_direct is not None:
                _ready(demb)
                demb = None
        return dP, demb, None, None, None


def gumbel_bow(P, row_c, emb, tau, noise=None, rng=None, site=0):
    if noise is None:
        noise = torch.empty_like(P)
        _lib.call("gumbel_noise", _p(noise), noise.numel(), int(site), _p(rng.seed), _stream())
    return _GumbelBow.apply(P, emb, as_idx(row_c), float(tau), noise)


# ------------------------------------------------------------------------------------------------ LSTM cell / losses
class _LstmCell(Function):
    @staticmethod
    def forward(ctx, gx, gh, c_prev, h_prev, active):
        _need_gpu(gx)
        gx, gh, c_prev, h_prev = _c(gx), _c(gh), _c(c_prev), _c(h_prev)
        N, D = c_prev.shape
        h = torch.empty_like(c_prev)
        c = torch.empty_like(c_prev)
        gates = torch.empty_like(gx)
        _lib.call("lstm_cell_fwd", _p(gx), _p(gh), _p(c_prev), _p(h_prev), _p(active), _p(h), _p(c), _p(gates), N, D, _stream())
        ctx.save_for_backward(gates, c_prev, active)
        ctx.cfg = (N, D)
        ctx.set_materialize_grads(False)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, active = ctx.saved_tensors
        N, D = ctx.cfg
        dev = gates.device
        dh = _c(dh) if dh is not None else torch.zeros(N, D, dtype=torch.float32, device=dev)
        dc = _c(dc) if dc is not None else torch.zeros(N, D, dtype=torch.float32, device=dev)
        dg = torch.empty_like(gates)
        dcp = torch.empty_like(c_prev)
        dhp = torch.empty_like(c_prev)
        _lib.call("lstm_cell_bwd", _p(dh), _p(dc), _p(gates), _p(c_prev), _p(active), _p(dg), _p(dcp), _p(dhp), N, D, _stream())
        return dg, dg, dcp, dhp, None


def lstm_cell(gx, gh, c_prev, h_prev, active):
    return _LstmCell.apply(gx, gh, c_prev, h_prev, active)


class _LstmSeq(Function):
    """One direction of the LSTM recurrence over every video's step sequence as a single autograd node.

    Per time step: one (N, D)×(D, 4D) GEMM + one fused cell kernel forward; one cell kernel + one dgrad GEMM (accumulating onto
    the pass-through gradient) backward.  The recurrent weight gradient is ONE GEMM over all time steps (K = S·N rows) on a side
    stream, the input-projection gradient one row gather.  No per-step gathers, adds or fills (reference: nn.LSTM model.py:859-860
    as used at :1022-1024; time-major state, inactive (padded) steps pass (h, c) through)."""

    @staticmethod
    def forward(ctx, gx_all, w_hh, rows_t, active_t, pick, wgrad):
        _need_gpu(gx_all)
        gx_all, w = _c(gx_all), _c(w_hh)
        S, N = len(rows_t), rows_t[0].numel()
        D = w.shape[1]
        dev = gx_all.device
        h_all = torch.empty(S + 1, N, D, dtype=torch.float32, device=dev)
        c_all = torch.empty(S + 1, N, D, dtype=torch.float32, device=dev)
        h_all[0].zero_(); c_all[0].zero_()
        gates = torch.empty(S, N, 4 * D, dtype=torch.float32, device=dev)
        gh = torch.empty(N, 4 * D, dtype=torch.float32, device=dev)
        for t in range(S):
            _gemm(h_all[t], D, 1, w, D, 1, gh, N, 4 * D, D)
            _lib.call("lstm_cell_fwd_idx", _p(gx_all), _p(rows_t[t]), _p(gh), _p(c_all[t]), _p(h_all[t]), _p(active_t[t]),
                      _p(h_all[t + 1]), _p(c_all[t + 1]), _p(gates[t]), N, D, _stream())
        out = torch.index_select(h_all[1:].reshape(S * N, D), 0, pick)
        ctx.save_for_backward(gates, c_all, h_all, w, pick)
        ctx.lists = (rows_t, active_t)
        ctx.direct = wgrad
        ctx.cfg = (S, N, D, gx_all.shape[0])
        return out

    @staticmethod
    def backward(ctx, dout):
        gates, c_all, h_all, w, pick = ctx.saved_tensors
        rows_t, active_t = ctx.lists
        S, N, D, T = ctx.cfg
        dev = dout.device
        dhs = torch.zeros(S * N, D, dtype=torch.float32, device=dev)
        dhs.index_copy_(0, pick.long() if pick.dtype != torch.int64 else pick, _c(dout))
        dhs = dhs.view(S, N, D)
        dG = torch.empty(S, N, 4 * D, dtype=torch.float32, device=dev)
        dh = torch.zeros(N, D, dtype=torch.float32, device=dev)
        dc = torch.zeros(N, D, dtype=torch.float32, device=dev)
        dh2, dc2 = torch.empty_like(dh), torch.empty_like(dc)
        wt = w.t().contiguous()         # (D, 4D): the S dgrad GEMMs then read both operands k-contiguously (12 tiles × K = 4D)
        for t in range(S - 1, -1, -1):
            _lib.call("lstm_cell_bwd_seq", _p(dhs[t]), _p(dh), _p(dc), _p(gates[t]), _p(c_all[t]), _p(active_t[t]), _p(dG[t]), _p(dc2),
                      _p(dh2), N, D, _stream())
            _gemm(dG[t], 4 * D, 1, wt, 4 * D, 1, dh2, N, D, 4 * D, accumulate=1)      # dh_{t-1} = pass-through + dgates · W_hh
            dh, dh2 = dh2, dh
            dc, dc2 = dc2, dc
        dG2, hp = dG.view(S * N, 4 * D), h_all[:S].reshape(S * N, D)
        wgrad = ctx.direct
        dw = None
        if wgrad is not None or ctx.needs_input_grad[1]:
            acc = 1 if wgrad is not None else 0
            dw = wgrad if wgrad is not None else torch.empty_like(w)
            with _side_of(wgrad, dG2, hp):
                _gemm(dG2, 4 * D, 0, hp, D, 0, dw, 4 * D, D, S * N, accumulate=acc)
            if wgrad is not None:
                _ready(wgrad, "w")
                dw = None
        dgx = torch.index_select(dG2, 0, pick) if ctx.needs_input_grad[0] else None
        return dgx, dw, None, None, None, None


class _GemmProblem(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int)]


def _gemm_pair(As, Bs, Cs, M, N, K, accumulate=0, x3=False):
    """C_z (+)= A_z · B_zᵀ for z = 0, 1 (all k-contiguous fp32) in one grouped launch"""
    probs = (_GemmProblem * 2)()
    for z in range(2):
        probs[z] = _GemmProblem(As[z].data_ptr(), Bs[z].data_ptr(), Cs[z].data_ptr(), M, N, K, As[z].stride(0), Bs[z].stride(0),
                                Cs[z].stride(0))
    _lib.call("gemm_group_x3" if x3 else "gemm_group", ctypes.addressof(probs), 2, 1, 1, accumulate, _stream())


def _ptr2(a, b):
    arr = (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    return ctypes.addressof(arr), arr          # keep `arr` alive until the call returns


LSTM_DGRAD_PARTS = int(os.environ.get("SVPC_LSTM_DGRAD_PARTS", "4"))
LSTM_FUSED_STEP = os.environ.get("SVPC_LSTM_FUSED_STEP", "1") != "0"      # recurrent projection and cell of a time step in one launch


class _BiLstmSeq(Function):
    """Both directions of the BiLSTM recurrence as ONE autograd node advancing in lockstep: per time step one grouped GEMM (the two
    recurrent projections) and one cell launch (both directions) forward, one cell launch and one grouped dgrad GEMM backward —
    half the launches of two independent direction nodes (see _LstmSeq for the per-direction scheme).  bf16-MFMA precision only."""

    @staticmethod
    def forward(ctx, gx_f, gx_b, w_f, w_b, rows_f, rows_b, active_t, pick_f, pick_b, wgrad_f, wgrad_b, summed=False):
        _need_gpu(gx_f)
        ctx.summed = bool(summed)
        gx = [_c(gx_f), _c(gx_b)]
        w = [_c(w_f), _c(w_b)]
        rows = [rows_f, rows_b]
        S, N = len(rows_f), rows_f[0].numel()
        D = w[0].shape[1]
        dev = gx_f.device
        mk = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        # states of both directions in ONE buffer (direction, h|c, time, N, D) with the four initial states adjacent: one fill
        state = mk(2, 2, S + 1, N, D)
        state[:, :, 0].zero_()
        h_all, c_all = [state[0, 0], state[1, 0]], [state[0, 1], state[1, 1]]
        gates = [mk(S, N, 4 * D) for _ in range(2)]
        gh = [mk(N, 4 * D) for _ in range(2)]
        st = _stream()
        fused = LSTM_FUSED_STEP and D % 16 == 0 and (w[0].data_ptr() | w[1].data_ptr()) % 16 == 0
        pw = _ptr2(w[0], w[1])
        pgx = _ptr2(gx[0], gx[1])
        for t in range(S):
            args = [pgx, _ptr2(rows[0][t], rows[1][t]), _ptr2(gh[0], gh[1]), _ptr2(c_all[0][t], c_all[1][t]),
                    _ptr2(h_all[0][t], h_all[1][t]), None, _ptr2(h_all[0][t + 1], h_all[1][t + 1]),
                    _ptr2(c_all[0][t + 1], c_all[1][t + 1]), _ptr2(gates[0][t], gates[1][t])]
            if fused:       # recurrent projection + cell in one launch per time step
                _lib.call("lstm_pair_step_fwd_x3" if is_x3() else "lstm_pair_step_fwd", args[4][0], args[3][0], pw[0], args[0][0],
                          args[1][0], _p(active_t[t]), args[6][0], args[7][0], args[8][0], N, D, st)
                continue
            _gemm_pair([h_all[0][t], h_all[1][t]], w, gh, N, 4 * D, D, x3=is_x3())
            _lib.call("lstm_pair_fwd", args[0][0], args[1][0], args[2][0], args[3][0], args[4][0], _p(active_t[t]), args[6][0],
                      args[7][0], args[8][0], N, D, st)
        ctx.save_for_backward(gates[0], gates[1], c_all[0], c_all[1], h_all[0], h_all[1], w[0], w[1], pick_f, pick_b)
        ctx.lists = active_t
        ctx.direct = (wgrad_f, wgrad_b)
        ctx.cfg = (S, N, D)
        pair_ok = (D % 4 == 0 and pick_f.dtype == torch.int32 and pick_b.dtype == torch.int32 and pick_f.numel() == pick_b.numel())
        ctx.pair_ok = pair_ok
        if summed and pair_ok:      # the two directions' outputs picked from the time-major states and summed in one launch
            out = mk(pick_f.numel(), D)
            _lib.call("pair_rows", _p(h_all[0][1:]), _p(pick_f), _p(h_all[1][1:]), _p(pick_b), _p(out), None, pick_f.numel(), D, 0, st)
            return out
        outs = [torch.index_select(h_all[z][1:].reshape(S * N, D), 0, pk) for z, pk in enumerate((pick_f, pick_b))]
        if summed:
            return outs[0] + outs[1]
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, dout_f, dout_b=None):
        if ctx.summed:
            dout_b = dout_f
        g0, g1, c0, c1, h0, h1, w0, w1, pick_f, pick_b = ctx.saved_tensors
        gates, c_all, h_all, w, picks = [g0, g1], [c0, c1], [h0, h1], [w0, w1], [pick_f, pick_b]
        active_t = ctx.lists
        S, N, D = ctx.cfg
        dev = g0.device
        mk = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        # every zero-initialised buffer of the backward in ONE allocation (one fill): the scattered output gradients of both
        # directions (2·S·N rows) and the four running state gradients (4·N rows)
        zeros = torch.zeros(2 * S * N + 4 * N, D, dtype=torch.float32, device=dev)
        dhs = []
        if ctx.summed and ctx.pair_ok and dout_f is not None:      # the one output gradient into both directions' time-major rows: one launch
            _lib.call("pair_rows", _p(zeros), _p(picks[0]), _p(zeros[S * N:]), _p(picks[1]), _p(_c(dout_f)), None, picks[0].numel(), D, 1,
                      _stream())
            dhs = [zeros[:S * N].view(S, N, D), zeros[S * N:2 * S * N].view(S, N, D)]
        else:
            for z, d in enumerate((dout_f, dout_b)):
                t_ = zeros[z * S * N:(z + 1) * S * N]
                if d is not None:
                    t_.index_copy_(0, picks[z].long(), _c(d))
                dhs.append(t_.view(S, N, D))
        dG = [mk(S, N, 4 * D) for _ in range(2)]
        tail = zeros[2 * S * N:].view(4, N, D)
        dh = [tail[0], tail[1]]
        dc = [tail[2], tail[3]]
        dh2, dc2 = [mk(N, D) for _ in range(2)], [mk(N, D) for _ in range(2)]
        st = _stream()
        # (the per-step dgrad dh = dG·W_hh reads W_hh (4D, D) k-strided IN PLACE — svpc_gemm_group with b_kc = 0 — instead of a transposed
        # copy of both directions' weights made every step: 2 × 14 µs of ATen copies at D = 768)
        # the per-step dgrad (N × D over K = 4D) is bound by how fast ONE workgroup can pull its weight columns: cut K into
        # LSTM_DGRAD_PARTS k-parts (separate problems of the grouped launch, each writing a slab) that the next cell launch adds
        P = LSTM_DGRAD_PARTS if (4 * D) % (32 * LSTM_DGRAD_PARTS) == 0 else 1
        Kp = 4 * D // P
        slabs = [mk(P, N, D) for _ in range(2)] if P > 1 else None
        have_parts = False
        for t in range(S - 1, -1, -1):
            a = [_ptr2(dhs[0][t], dhs[1][t]), _ptr2(dh[0], dh[1]), _ptr2(dc[0], dc[1]), _ptr2(gates[0][t], gates[1][t]),
                 _ptr2(c_all[0][t], c_all[1][t]), None, _ptr2(dG[0][t], dG[1][t]), _ptr2(dc2[0], dc2[1]), _ptr2(dh2[0], dh2[1])]
            if P > 1:
                sp = _ptr2(slabs[0], slabs[1])
                _lib.call("lstm_pair_bwd_parts", a[0][0], a[1][0], sp[0] if have_parts else None, P if have_parts else 0, a[2][0], a[3][0],
                          a[4][0], _p(active_t[t]), a[6][0], a[7][0], a[8][0], N, D, st)
                if t > 0:
                    probs = (_GemmProblem * (2 * P))()
                    for z in range(2):
                        for k in range(P):
                            probs[z * P + k] = _GemmProblem(dG[z][t].data_ptr() + 4 * k * Kp, w[z].data_ptr() + 4 * k * Kp * D,
                                                            slabs[z][k].data_ptr(), N, D, Kp, 4 * D, D, D)
                    _lib.call("gemm_group", ctypes.addressof(probs), 2 * P, 1, 0, 0, st)
                    have_parts = True
            else:
                _lib.call("lstm_pair_bwd", a[0][0], a[1][0], a[2][0], a[3][0], a[4][0], _p(active_t[t]), a[6][0], a[7][0], a[8][0], N, D, st)
                wt = [w[z].t().contiguous() for z in range(2)] if t == S - 1 else wt      # (unsplit fallback: a transposed copy, once)
                _gemm_pair([dG[0][t], dG[1][t]], wt, dh2, N, D, 4 * D, accumulate=1)      # dh_{t-1} = pass-through + dgates · W_hh
            dh, dh2 = dh2, dh
            dc, dc2 = dc2, dc
        dws, dgx = [None, None], [None, None]
        for z in range(2):
            dG2, hp = dG[z].view(S * N, 4 * D), h_all[z][:S].reshape(S * N, D)
            wgrad = ctx.direct[z]
            if wgrad is not None or ctx.needs_input_grad[2 + z]:
                if not (wgrad is not None and defer_wgrad(dG2, hp, wgrad, None)):
                    acc = 1 if wgrad is not None else 0
                    dw = wgrad if wgrad is not None else torch.empty_like(w[z])
                    _gemm(dG2, 4 * D, 0, hp, D, 0, dw, 4 * D, D, S * N, accumulate=acc)
                    if wgrad is not None:
                        _ready(wgrad, "w")
                    else:
                        dws[z] = dw
            if ctx.needs_input_grad[z] and not (ctx.pair_ok and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]):
                dgx[z] = torch.index_select(dG2, 0, picks[z])
        if ctx.pair_ok and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:       # both gate gradients gathered in one launch
            R_ = picks[0].numel()
            dgx = [mk(R_, 4 * D), mk(R_, 4 * D)]
            _lib.call("pair_rows", _p(dG[0]), _p(picks[0]), _p(dG[1]), _p(picks[1]), _p(dgx[0]), _p(dgx[1]), R_, 4 * D, 2, _stream())
        return dgx[0], dgx[1], dws[0], dws[1], None, None, None, None, None, None, None, None


def bilstm_sequences(gx_f, gx_b, w_f, w_b, rows_f, rows_b, active_t, pick_f, pick_b, summed=False):
    """Both LSTM directions (see lstm_sequence for the arguments) → (out_f, out_b), each (T, D); summed=True: their sum (what the model
    uses, model.py:1024) — picked and added in one launch, the backward scatters / gathers both directions in one launch each."""
    if _fast() and gx_f.is_cuda and not BWD_EXACT:
        return _BiLstmSeq.apply(gx_f, gx_b, w_f, w_b, rows_f, rows_b, active_t, pick_f, pick_b, _direct(w_f), _direct(w_b), summed)
    of, ob = lstm_sequence(gx_f, w_f, rows_f, active_t, pick_f), lstm_sequence(gx_b, w_b, rows_b, active_t, pick_b)
    return add(of, ob) if summed else (of, ob)


def lstm_sequence(gx_all, w_hh, rows_t, active_t, pick):
    """gx_all (T, 4D): input projections of every step row; rows_t[t] (N,) the step row each video consumes at time t;
    active_t[t] (N,) 1/0; pick (T,) position of every step row's output in the time-major (S·N) state → (T, D)."""
    return _LstmSeq.apply(gx_all, w_hh, rows_t, active_t, pick, _direct(w_hh))


class _BceRows(Function):
    @staticmethod
    def forward(ctx, p, y, widths):
        _need_gpu(p)
        p, y = _c(p), _c(y)
        R, C = p.shape
        w = widths.dev(p.device)
        out = torch.empty(R, dtype=torch.float32, device=p.device)
        _lib.call("bce_rows_fwd", _p(p), _p(y), _p(w), _p(out), R, C, _stream())
        ctx.save_for_backward(p, y, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, y, w = ctx.saved_tensors
        dp = torch.empty_like(p)
        _lib.call("bce_rows_bwd", _p(_c(dout)), _p(p), _p(y), _p(w), _p(dp), p.shape[0], p.shape[1], _stream())
        return dp, None, None


def bce_rows(p, y, widths):
    return _BceRows.apply(p, y, as_idx(widths))


class _AslRows(Function):
    @staticmethod
    def forward(ctx, p, y, active, gneg, gpos, clip, eps):
        _need_gpu(p)
        p, y = _c(p), _c(y)
        R, C = p.shape
        out = torch.empty(R, dtype=torch.float32, device=p.device)
        _lib.call("asl_rows_fwd", _p(p), _p(y), _p(active), _p(out), R, C, gneg, gpos, clip, eps, _stream())
        ctx.save_for_backward(p, y, active)
        ctx.cfg = (gneg, gpos, clip, eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, y, active = ctx.saved_tensors
        dp = torch.empty_like(p)
        _lib.call("asl_rows_bwd", _p(_c(dout)), _p(p), _p(y), _p(active), _p(dp), p.shape[0], p.shape[1], *ctx.cfg, _stream())
        return dp, None, None, None, None, None, None


def asl_rows(p, y, row_active, gamma_neg=4.0, gamma_pos=1.0, clip=0.05, eps=1e-8):
    return _AslRows.apply(p, y, row_active, float(gamma_neg), float(gamma_pos), float(clip), float(eps))


_TICKETS = {}


class _LossTail(Function):
    """total = Σ cap_rows + Σ BCE(e_p) + Σ ASL(a_p) + λ·(Σ BCE(r_e) + Σ ASL(r_a)) in one launch; backward in one launch."""

    @staticmethod
    def forward(ctx, cap_rows, e_p, a_p, r_e, r_a, align, act, widths, lam, gneg, gpos, clip, eps):
        _need_gpu(cap_rows)
        cap_rows = _c(cap_rows)
        e_p, a_p, r_e, r_a = [(_c(t) if t is not None else None) for t in (e_p, a_p, r_e, r_a)]
        align, act = _c(align), _c(act)
        dev = cap_rows.device
        R, Ce = align.shape
        Ca = act.shape[1]
        w = widths.dev(dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        scratch = torch.empty(_lib.load().svpc_loss_tail_ws_floats(cap_rows.numel(), R), dtype=torch.float32, device=dev)
        tkey = (dev, _stream())   # one counter per (device, stream): launches on one stream are ordered, two streams (or two models on
        counter = _TICKETS.get(tkey)    # their own streams) must not share the ticket word of a launch in flight
        if counter is None:       # zeroed once; every launch leaves it at zero
            counter = _TICKETS[tkey] = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("loss_tail_fwd", _p(cap_rows), cap_rows.numel(), _p(e_p), _p(align), _p(w), R, Ce, _p(a_p), _p(act), Ca, _p(r_e), _p(r_a),
                  float(lam), gneg, gpos, clip, eps, _p(out), _p(scratch), _p(counter), _stream())
        ctx.save_for_backward(e_p, a_p, r_e, r_a, align, act, w)
        ctx.cfg = (cap_rows.numel(), R, Ce, Ca, float(lam), gneg, gpos, clip, eps)
        ctx.parts = out
        return out[0]

    @staticmethod
    def backward(ctx, dout):
        e_p, a_p, r_e, r_a, align, act, w = ctx.saved_tensors
        n_cap, R, Ce, Ca, lam, gneg, gpos, clip, eps = ctx.cfg
        dev = align.device
        need = ctx.needs_input_grad
        mk = lambda t, on: torch.empty_like(t) if (t is not None and on) else None
        d_cap = torch.empty(n_cap, dtype=torch.float32, device=dev) if need[0] else None
        de, da, dre, dra = mk(e_p, need[1]), mk(a_p, need[2]), mk(r_e, need[3]), mk(r_a, need[4])
        _lib.call("loss_tail_bwd", _p(_c(dout)), n_cap, _p(e_p), _p(align), _p(w), R, Ce, _p(a_p), _p(act), Ca, _p(r_e), _p(r_a), lam, gneg,
                  gpos, clip, eps, _p(d_cap), _p(de), _p(da), _p(dre), _p(dra), _stream())
        return d_cap, de, da, dre, dra, None, None, None, None, None, None, None, None


def loss_tail(cap_rows, e_p, a_p, r_e, r_a, align, act, widths, lam, gamma_neg=4.0, gamma_pos=1.0, clip=0.05, eps=1e-8):
    """Scalar training loss from the caption-loss rows and the (re-)simulator probabilities; e_p / a_p / r_e / r_a may be None."""
    return _LossTail.apply(cap_rows, e_p, a_p, r_e, r_a, align, act, as_idx(widths), float(lam), float(gamma_neg), float(gamma_pos),
                           float(clip), float(eps))


def greedy_pick(scores, row_c, row_x, lt, pos, unk, append=None):
    """scores (T*lt, ≥C): → (next_ext, next_model) int32 (T,), see svpc_greedy_pick.  ``append`` = (text, ext, col): the picked ids are
    also stored as column ``col`` of the two (T, Lt) int32 id matrices of the decoding loop (one launch instead of three)."""
    scores = _c(scores)
    dev = scores.device
    n = scores.shape[0] // lt
    ext = torch.empty(n, dtype=torch.int32, device=dev)
    mod = torch.empty(n, dtype=torch.int32, device=dev)
    if append is not None:
        tm, em, col = append
        assert tm.dtype == em.dtype == torch.int32 and tm.is_contiguous() and em.is_contiguous() and tm.shape == em.shape and tm.shape[0] == n
        _lib.call("greedy_pick_append", _p(scores), scores.stride(0), _p(as_idx(row_c).dev(dev)), _p(as_idx(row_x).dev(dev)), n, lt, int(pos),
                  int(unk), _p(ext), _p(mod), _p(tm), _p(em), tm.shape[1], int(col), _stream())
        return ext, mod
    _lib.call("greedy_pick", _p(scores), scores.stride(0), _p(as_idx(row_c).dev(dev)), _p(as_idx(row_x).dev(dev)), n, lt, int(pos),
              int(unk), _p(ext), _p(mod), _stream())
    return ext, mod
