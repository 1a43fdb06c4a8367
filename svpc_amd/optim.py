"""Training-step tail on the MI355X: flat gradient arena + fused BertAdam/EMA + bucketed RCCL gradient all-reduce.

reference: src/rtransformer/optimization.py:219-338 (BertAdam: per-tensor clip to 1.0, no bias correction, decay added
to the update, warmup-linear schedule :162-171), :183-216 (EMA), src/train.py:140-147 (backward → global clip → step →
EMA), :338-343 (weight-decay grouping by name).

Design: every trainable tensor that receives a gradient gets its ``.grad`` re-pointed into ONE contiguous fp32 arena
(likewise m, v, EMA shadows).  That makes zero_grad one memset, the data-parallel exchange a handful of large
all-reduces over arena slices (xGMI is per-link bound: few big messages, overlapped with backward through
post-accumulate hooks), and clip + Adam + EMA three kernel launches over a chunk table (svpc_opt_step).
"""
from __future__ import annotations

import ctypes
import re

import torch

from . import _lib

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")  # src/train.py:339


def warmup_linear(progress, warmup):
    """optimization.py:166-171."""
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


# Bumped whenever parameter VALUES change without autograd's version counters noticing (the fused optimizer kernels, a replayed step
# graph, the EMA swap).  Whoever caches something derived from the weights (Translator's per-checkpoint embedding table and its decode
# graphs) keys it on this counter beside the tensors' own versions.
WEIGHTS_EPOCH = [0]


class _Meta(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("ema", ctypes.c_void_p), ("n", ctypes.c_longlong), ("wd", ctypes.c_float), ("pad", ctypes.c_int),
                ("shadow", ctypes.c_void_p), ("shadow_lo", ctypes.c_void_p)]


_QKV = {"query.weight": 0, "key.weight": 1, "value.weight": 2, "query.bias": 3, "key.bias": 4, "value.bias": 5}


_MEMKV = re.compile(r"^(.*\.layer\.)(\d+)\.dec_enc_attention\.(key|value)\.(weight|bias)$")


def _ordered(named_params):
    """Keep registration order, but lay each attention block out as [Wq Wk Wv | bq bk bv] so the packed projection
    ((3D, D) / (2D, D)) — weight, bf16 shadow and gradient alike — is one contiguous region a GEMM can use in place.
    The cross-attention key / value projections of a decoder stack all read the same memory rows (model.py:643-651), so
    they are laid out across the layers as [Wk0 Wv0 Wk1 Wv1 … | bk0 bv0 bk1 bv1 …]: one (L·2D, D) projection for the stack."""
    first, keyed = {}, []
    for i, (n, p) in enumerate(named_params):
        m = _MEMKV.match(n)
        if m is not None:
            first.setdefault(m.group(1), i)
            keyed.append(((first[m.group(1)], 0 if m.group(4) == "weight" else 1, int(m.group(2)), 0 if m.group(3) == "key" else 1), n, p))
            continue
        for suf, k in _QKV.items():
            if n.endswith("." + suf):
                pre = n[:-len(suf)]
                first.setdefault(pre, i)
                keyed.append(((first[pre], k), n, p))
                break
        else:
            keyed.append(((i, -1), n, p))
    keyed.sort(key=lambda t: t[0])
    return [(n, p) for _, n, p in keyed]


def _layout(named_params):
    named_params = _ordered(list(named_params))
    names = [n for n, _ in named_params]
    params = [p for _, p in named_params]
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 7) // 8 * 8          # keep every tensor 16-byte aligned, in the bf16 shadow too
    return names, params, offsets, off


def _run(ids):
    return ids == list(range(ids[0], ids[0] + len(ids)))


def _packed_groups(names, params, offsets):
    """→ [(query weight parameter, [6 member indices], kinds)] for every attention block; ``kinds`` ⊆ {"qkv", "kv", "q"} are
    the packed projections whose members are contiguous in this layout."""
    idx = {n: i for i, n in enumerate(names)}
    out = []
    for n in names:
        if not n.endswith(".query.weight"):
            continue
        pre = n[:-len("query.weight")]
        need = [pre + s for s in ("query.weight", "key.weight", "value.weight", "query.bias", "key.bias", "value.bias")]
        if not all(k in idx for k in need):
            continue
        ids = [idx[k] for k in need]
        D_out, D_in = params[ids[0]].shape
        if (D_out * D_in) % 8 or D_out % 8:
            continue
        kinds = ["q"]
        if _run(ids[:3]) and _run(ids[3:]):
            kinds.append("qkv")
        if _run(ids[1:3]) and _run(ids[4:]):
            kinds.append("kv")
        out.append((params[ids[0]], ids, kinds))
    return out


def _packed_views(flat, offsets, ids, D_out, D_in, kinds=("qkv", "kv", "q")):
    ow, ob = offsets[ids[0]], offsets[ids[3]]
    v = {
        "qkv": lambda: (flat[ow:ow + 3 * D_out * D_in].view(3 * D_out, D_in), flat[ob:ob + 3 * D_out]),
        "kv": lambda: (flat[offsets[ids[1]]:offsets[ids[1]] + 2 * D_out * D_in].view(2 * D_out, D_in),
                       flat[offsets[ids[4]]:offsets[ids[4]] + 2 * D_out]),
        "q": lambda: (flat[ow:ow + D_out * D_in].view(D_out, D_in), flat[ob:ob + D_out]),
    }
    return {k: v[k]() for k in kinds}


def _stack_groups(names, params):
    """→ [(anchor = layer 0's key weight, weight ids [k0 v0 k1 v1 …], bias ids)] for every decoder stack whose cross-attention
    key / value projections are contiguous across the layers (see _ordered)."""
    by_pre = {}
    for i, n in enumerate(names):
        m = _MEMKV.match(n)
        if m is not None:
            by_pre.setdefault(m.group(1), {})[(m.group(4), int(m.group(2)), m.group(3))] = i
    out = []
    for pre, d in by_pre.items():
        L = 1 + max(k[1] for k in d)
        try:
            wi = [d[("weight", l, kv)] for l in range(L) for kv in ("key", "value")]
            bi = [d[("bias", l, kv)] for l in range(L) for kv in ("key", "value")]
        except KeyError:
            continue
        D_out, D_in = params[wi[0]].shape
        if L < 2 or not (_run(wi) and _run(bi)) or (D_out * D_in) % 8 or D_out % 8:
            continue
        if any(tuple(params[i].shape) != (D_out, D_in) for i in wi):
            continue
        out.append((params[wi[0]], wi, bi))
    return out


def _stack_views(flat, offsets, wi, bi, D_out, D_in):
    n = len(wi)
    return (flat[offsets[wi[0]]:offsets[wi[0]] + n * D_out * D_in].view(n * D_out, D_in), flat[offsets[bi[0]]:offsets[bi[0]] + n * D_out])


class WeightStore:
    """All parameters of a model in ONE contiguous fp32 buffer (``p.data`` re-pointed into it, attention blocks laid out
    [Wq Wk Wv | bq bk bv]) plus a bf16 shadow of the same layout.

    * the packed Q/K/V (and K/V) projection weights/biases are plain views — no concatenation per forward;
    * the shadow is the B operand of the direct-to-LDS bf16 GEMMs (svpc_gemm_glds) in the bf16 activation stream; the
      fused optimizer rewrites it in the Adam kernel, anything else that changes a parameter is caught through the
      tensor version counter (``ops._shadow``) or refreshed wholesale with ``refresh()``.
    """

    def __init__(self, named_params, layout=None, with_lo=None):
        from . import ops
        self.names, self.params, self.offsets, self.numel = layout if layout is not None else _layout(named_params)
        dev = self.params[0].device
        # bf16x3 mode: a second shadow plane lo = bf16(w - bf16(w)), same layout, ``numel`` elements behind the first in ONE buffer —
        # every shadow view is tagged with that offset (``_svpc_lo``), which is how the x3 GEMM finds the lo plane of its B operand
        self.with_lo = ops.is_x3() if with_lo is None else bool(with_lo)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.shadow2 = torch.zeros((2 if self.with_lo else 1) * self.numel, dtype=torch.bfloat16, device=dev)
        self.shadow = self.shadow2[:self.numel]
        self.shadow_lo = self.shadow2[self.numel:] if self.with_lo else None

        def tag(t):
            if self.with_lo:
                t._svpc_lo = self.numel
            return t
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                view = self.flat[o:o + n].view_as(p)
                view.copy_(p.detach())
                p.data = view
                p._svpc_bf16 = tag(self.shadow[o:o + n].view_as(p))
                p._svpc_store = self        # (the owner: ``for_model`` never adopts a parameter that already lives in a store)
        for q, ids, kinds in _packed_groups(self.names, self.params, self.offsets):
            D_out, D_in = q.shape
            fw = _packed_views(self.flat, self.offsets, ids, D_out, D_in, kinds)
            sw = _packed_views(self.shadow, self.offsets, ids, D_out, D_in, kinds)
            q._svpc_packed_w = {k: (fw[k][0], fw[k][1], tag(sw[k][0])) for k in fw}
            q._svpc_packed_w_members = [self.params[i] for i in ids]
        for a, wi, bi in _stack_groups(self.names, self.params):
            D_out, D_in = a.shape
            fw = _stack_views(self.flat, self.offsets, wi, bi, D_out, D_in)
            sw = _stack_views(self.shadow, self.offsets, wi, bi, D_out, D_in)
            a._svpc_stack_w = (fw[0], fw[1], tag(sw[0]))
            a._svpc_stack_w_members = [self.params[i] for i in wi]
        self.refresh()

    @classmethod
    def for_model(cls, model):
        """Inference-time use: the store that owns ``model``'s parameters — the fused optimizer's when one has been built (the
        reference builds a Translator on the live training model after every epoch, src/train.py:284: adopting the parameters a
        second time would re-point ``p.data`` away from the addresses the optimizer's tensor table, the EMA swap and any captured
        step graph hold, and training would silently continue on an orphaned copy) — else a new one over every parameter.
        Parameters outside the owning store (frozen embedding tables, modules unused in the model's mode) stay where they are:
        a captured graph may hold their addresses too."""
        owners = {}
        for p in model.parameters():
            st = getattr(p, "_svpc_store", None)
            if st is not None and p.data_ptr() >= st.flat.data_ptr() and p.data_ptr() < st.flat.data_ptr() + 4 * st.numel:
                owners.setdefault(id(st), [st, 0])[1] += p.numel()
        if owners:
            store = max(owners.values(), key=lambda e: e[1])[0]
        else:
            store = cls(list(model.named_parameters()))
        model._svpc_weight_store = store
        return store

    def refresh(self):
        """shadow ← bf16(weights) for the whole store (after load_state_dict, EMA swap, manual edits …)."""
        with torch.no_grad():
            self.shadow.copy_(self.flat)
            if self.with_lo:
                self.shadow_lo.copy_(self.flat - self.shadow.float())
        for p in self.params:
            p._svpc_bf16_ver = p._version

    def versions(self):
        return sum(p._version for p in self.params)


class GradArena:
    """Contiguous fp32 gradient storage; ``p.grad`` of every member is a view into it."""

    def __init__(self, named_params):
        self.names, self.params, self.offsets, self.numel = _layout(named_params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            view = self.flat[o:o + p.numel()].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
            p._svpc_direct = True       # ops write this parameter's gradient in place from now on
        # packed views for the fused Q/K/V (and K/V) projections
        for q, ids, kinds in _packed_groups(self.names, self.params, self.offsets):
            D_out, D_in = q.shape
            q._svpc_packed = _packed_views(self.flat, self.offsets, ids, D_out, D_in, kinds)
            members = {"qkv": ids, "kv": [ids[1], ids[2], ids[4], ids[5]], "q": [ids[0], ids[3]]}
            q._svpc_packed_members = {k: members[k] for k in kinds}
        for a, wi, bi in _stack_groups(self.names, self.params):
            D_out, D_in = a.shape
            a._svpc_stack = _stack_views(self.flat, self.offsets, wi, bi, D_out, D_in)
            a._svpc_stack_members = (wi, bi)

    def layout(self):
        return self.names, self.params, self.offsets, self.numel

    def zero(self):
        self.flat.zero_()

    def slice_of(self, i):
        return self.offsets[i], self.params[i].numel()


class FusedBertAdam:
    """BertAdam + optional global clip + optional EMA in three launches (clip semantics identical to
    clip_grad_norm_(model.parameters(), grad_clip) followed by BertAdam.step())."""

    def __init__(self, named_params, lr=1e-4, warmup=0.1, t_total=-1, weight_decay=0.01, max_grad_norm=1.0, grad_clip=1.0,
                 ema_decay=-1.0, b1=0.9, b2=0.999, eps=1e-6):
        self.named = [(n, p) for n, p in named_params if p.requires_grad]
        self.lr, self.warmup, self.t_total = lr, warmup, t_total
        self.weight_decay, self.max_grad_norm, self.grad_clip = weight_decay, max_grad_norm, grad_clip
        self.ema_decay, self.b1, self.b2, self.eps = ema_decay, b1, b2, eps
        self.step_count = 0
        self.arena = None

    # -- lazily built after the first backward: tensors whose grad is None are skipped, as BertAdam does (:290-291)
    def _build(self):
        live = [(n, p) for n, p in self.named if p.grad is not None]
        assert live, "no parameter received a gradient"
        self.arena = GradArena(live)
        dev = self.arena.flat.device
        # parameters move into one contiguous buffer with the arena's layout, next to their bf16 shadow
        self.weights = WeightStore(None, layout=self.arena.layout())
        self.m = torch.zeros_like(self.arena.flat)
        self.v = torch.zeros_like(self.arena.flat)
        self.ema = None
        if self.ema_decay >= 0:
            self.ema = torch.zeros_like(self.arena.flat)
            for p, o in zip(self.arena.params, self.arena.offsets):
                self.ema[o:o + p.numel()].copy_(p.detach().reshape(-1))
        lib = _lib.load()
        chunk = lib.svpc_opt_chunk()
        assert lib.svpc_opt_meta_bytes() == ctypes.sizeof(_Meta)
        live = list(zip(self.arena.names, self.arena.params))      # arena order (attention blocks are regrouped)
        metas = (_Meta * len(live))()
        chunk_tid, chunk_start, tco = [], [], [0]
        for i, ((name, p), o) in enumerate(zip(live, self.arena.offsets)):
            n = p.numel()
            assert p.is_contiguous() and p.dtype == torch.float32
            wd = 0.0 if any(nd in name for nd in NO_DECAY) else self.weight_decay
            es = 4
            metas[i] = _Meta(p.data_ptr(), self.arena.flat.data_ptr() + o * es, self.m.data_ptr() + o * es,
                             self.v.data_ptr() + o * es, (self.ema.data_ptr() + o * es) if self.ema is not None else None,
                             n, wd, 0, self.weights.shadow.data_ptr() + o * 2,
                             (self.weights.shadow_lo.data_ptr() + o * 2) if self.weights.with_lo else None)
            for s in range(0, n, chunk):
                chunk_tid.append(i)
                chunk_start.append(s)
            tco.append(len(chunk_tid))
        raw = bytes(metas)
        self.meta = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.chunk_tid = torch.tensor(chunk_tid, dtype=torch.int32, device=dev)
        self.chunk_start = torch.tensor(chunk_start, dtype=torch.int64, device=dev)
        self.tensor_chunk_off = torch.tensor(tco, dtype=torch.int32, device=dev)
        self.n_tensors, self.n_chunks = len(live), len(chunk_tid)
        self.partial = torch.empty(self.n_chunks, dtype=torch.float32, device=dev)
        self.norms_sq = torch.empty(self.n_tensors + 1, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(8)

    def ensure_built(self):
        if self.arena is None:
            self._build()
        return self.arena

    def zero_grad(self):
        if self.arena is None:
            for _, p in self.named:
                p.grad = None
        else:
            self.arena.zero()

    def scheduled_lr(self):
        if self.t_total <= 0:
            return self.lr
        return self.lr * warmup_linear(self.step_count / self.t_total, self.warmup)

    def set_hyper(self):
        """Refresh the device hyper-parameter words (lr of this step, EMA decay of this step: optimization.py:197)."""
        h = self._hyper_host
        h[0] = self.scheduled_lr()
        h[1] = min(self.ema_decay, (1.0 + self.step_count) / (10.0 + self.step_count)) if self.ema_decay >= 0 else -1.0
        h[2] = self.grad_clip if self.grad_clip and self.grad_clip > 0 else -1.0
        h[3] = self.max_grad_norm if self.max_grad_norm and self.max_grad_norm > 0 else -1.0
        h[4], h[5], h[6] = self.b1, self.b2, self.eps
        self.hyper.copy_(h, non_blocking=True)

    def launch(self):
        """The three kernels only (graph-capturable)."""
        from . import clip_graphs, ops
        clip_graphs.check_consumed()  # (a replayed clip-encoder forward whose backward replay was never requested: loud, not silent)
        WEIGHTS_EPOCH[0] += 1         # (the kernels rewrite the parameters behind autograd's version counters)
        ops.join_side()        # parameter-gradient kernels forked onto side streams
        stream = torch.cuda.current_stream().cuda_stream
        _lib.call("opt_step", self.meta.data_ptr(), self.chunk_tid.data_ptr(), self.chunk_start.data_ptr(),
                  self.tensor_chunk_off.data_ptr(), self.n_tensors, self.n_chunks, self.partial.data_ptr(),
                  self.norms_sq.data_ptr(), self.hyper.data_ptr(), stream)

    def step(self):
        self.ensure_built()
        self.set_hyper()
        self.launch()
        self.step_count += 1

    # -- EMA evaluation swap (optimization.py:205-216: assign the shadow weights for validation, resume afterwards)
    def ema_assign(self):
        assert self.ema is not None and self.arena is not None
        WEIGHTS_EPOCH[0] += 1
        self._backup = self.weights.flat.clone()
        with torch.no_grad():
            self.weights.flat.copy_(self.ema)
        self.weights.refresh()

    def ema_resume(self):
        WEIGHTS_EPOCH[0] += 1
        with torch.no_grad():
            self.weights.flat.copy_(self._backup)
        self._backup = None
        self.weights.refresh()

    def grad_norm(self):
        """global gradient norm seen by the last step (device scalar, no sync)."""
        return self.norms_sq[self.n_tensors].sqrt()


class GradReducer:
    """Data-parallel gradient exchange: SUM all-reduce of the arena in large buckets over RCCL (backend "nccl" on
    ROCm) — or gloo in the CPU tests.  Buckets are contiguous arena slices; a bucket is launched asynchronously as soon
    as all of its tensors have accumulated their gradient (post-accumulate hooks), so the exchange overlaps the rest of
    backward; ``finish()`` waits before the clip/optimizer kernels read the arena.  SUM (not mean) keeps the reference's
    sum-over-videos loss semantics (model.py:1110-1115, :1188): N ranks × 16 videos ≡ one process with 16·N videos."""

    def __init__(self, arena, process_group=None, bucket_bytes=64 << 20, overlap=True, force=False, wire_dtype="fp32", timeline=False):
        """wire_dtype="bf16": every bucket travels as bf16 (cast → all-reduce → the summed values written back into the fp32 arena):
        half the bytes on the xGMI links for one 2⁻⁹ rounding of each rank's contribution and the collective's own bf16 adds — an
        option for the day the fp32 exchange (340 MB per step) is what holds weak scaling under 0.9; the default stays fp32, which
        keeps N ranks ≡ one process with the N-fold batch exactly (tests/test_dp_gloo.py bounds the bf16 error against it).
        timeline=True: per-bucket issue / completion timestamps (HIP events on the compute stream, or host clocks on CPU) for
        ``timeline_ms()`` — what bench.py prints in ``exchange.buckets`` so that an efficiency < 0.9 can be diagnosed from one line."""
        import torch.distributed as dist
        assert wire_dtype in ("fp32", "bf16")
        self.wire_dtype, self.timeline = wire_dtype, bool(timeline)
        self._wire, self._stamps, self._t0 = {}, [], None
        self.dist, self.arena, self.pg = dist, arena, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force: issue the collectives even with one rank (rehearsal of the data-parallel code path on a one-GPU box)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.skip = False          # timing aid (bench.py): walk the whole protocol but do not call all_reduce
        self.buckets = []          # (start, end, [param indices])
        cur, cur_start, cur_bytes = [], 0, 0
        # parameters are registered in forward order, so backward finishes them roughly last-to-first:
        # walk the arena from the end so the first bucket to complete is the first launched
        order = list(range(len(arena.params)))
        for i in reversed(order):
            cur.append(i)
            cur_bytes += arena.params[i].numel() * 4
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self.works = []
        self.overlap = overlap and self.active
        self._handles = []
        self._by_pack = {}
        if self.overlap:
            from . import ops
            bucket_of = {}
            for bi, (s, e, members) in enumerate(self.buckets):
                for i in members:
                    bucket_of[i] = bi
                    self._handles.append(arena.params[i].register_post_accumulate_grad_hook(self._make_hook(i, bi)))
            # Readiness signals.  (1) The post-accumulate hook of a parameter: autograd runs a leaf's AccumulateGrad node once, after
            # EVERY consumer of the leaf has run its backward — also when those backwards wrote the gradient in place (or queued
            # the write) and handed autograd no tensor (the hook fires on an undefined gradient too; tests/test_dp_gloo.py pins
            # that) — so it is the "last write has been enqueued or queued" signal, multi-use parameters included; the queues are
            # flushed before a bucket is exchanged (``_launch``).  (2) Packed Q/K/V (K/V) projections use views of the weight
            # store that have no autograd edge to their member parameters: their in-place writes are reported by pointer
            # (``ops._ready`` → ``_ready_ptr``), one write per view and step.
            for i, p in enumerate(arena.params):
                packed = getattr(p, "_svpc_packed", None)
                if packed is not None:
                    for which, (wg, bg) in packed.items():
                        if which == "q":       # the one-member view IS the query parameter, which reaches autograd (and its hook) whenever it is
                            continue           # used: a pointer report for it would count the same write twice (fused cross-attention, round 5)
                        mem = p._svpc_packed_members[which]
                        half = len(mem) // 2
                        self._by_pack[("w", wg.data_ptr(), wg.numel())] = [(j, bucket_of[j]) for j in mem[:half]]
                        self._by_pack[("b", bg.data_ptr(), bg.numel())] = [(j, bucket_of[j]) for j in mem[half:]]
                stack = getattr(p, "_svpc_stack", None)
                if stack is not None:
                    wi, bi = p._svpc_stack_members
                    self._by_pack[("w", stack[0].data_ptr(), stack[0].numel())] = [(j, bucket_of[j]) for j in wi]
                    self._by_pack[("b", stack[1].data_ptr(), stack[1].numel())] = [(j, bucket_of[j]) for j in bi]
            ops.GRAD_READY_HOOK = self._ready_ptr
        self.reset()

    def _close(self, members):
        lo = min(self.arena.offsets[i] for i in members)
        hi = max(self.arena.offsets[i] + (self.arena.params[i].numel() + 7) // 8 * 8 for i in members)
        self.buckets.append((lo, hi, list(members)))

    def reset(self):
        self.left = [set(m) for _, _, m in self.buckets]
        self.works = []
        self.launched = [False] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self._next = 0
        self._releasing = False
        self._finishing = False
        self._in_launch = 0

    def mark_all_unlaunched(self):
        """first step after construction: the hooks were not installed during the backward that has just run"""
        self.reset()

    def bytes_per_step(self):
        return sum((e - s) * (2 if self.wire_dtype == "bf16" else 4) for s, e, _ in self.buckets)

    # ---- per-bucket timeline (diagnostics)
    def _now(self):
        import time
        if self.arena.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return time.perf_counter()

    def mark_step_start(self):
        """call at the start of a step (before zero_grad) when ``timeline`` is on"""
        if self.timeline:
            self._t0 = self._now()
            self._stamps = []

    def timeline_ms(self):
        """→ [{bucket, bytes, issued_ms, complete_ms}] of the last finished step, times since ``mark_step_start`` (GPU: the compute
        stream's clock — ``complete`` is when the compute stream could pass the bucket's wait, issue order)"""
        if not self.timeline or self._t0 is None:
            return None
        if self.arena.flat.is_cuda:
            torch.cuda.synchronize()
            ms = lambda t: self._t0.elapsed_time(t)
        else:
            ms = lambda t: 1000.0 * (t - self._t0)
        return [{"bucket": bi, "bytes": nb, "issued_ms": ms(t_i), "complete_ms": ms(t_c) if t_c is not None else None}
                for bi, nb, t_i, t_c in self._stamps]

    def close(self):
        """remove the hooks (tests; a reducer normally lives as long as the model)"""
        from . import ops
        for h in self._handles:
            h.remove()
        self._handles = []
        if ops.GRAD_READY_HOOK == self._ready_ptr:
            ops.GRAD_READY_HOOK = None

    def _launch(self, bi):
        from . import ops
        self.launched[bi] = True     # first: flushing the queues below reports more gradients ready (re-entrant)
        self._in_launch += 1         # (those reports may name members of buckets already released by their hooks: the same write,
        try:                         #  seen by hook and by pointer — expected; only a report OUTSIDE a flush is a second write)
            ops.flush_pending()      # the bucket's gradients may still sit in a deferred-tail queue or on a side stream
        finally:
            self._in_launch -= 1
        s, e, _ = self.buckets[bi]
        if self.skip:
            return
        if self.wire_dtype == "bf16":
            buf = self._wire.get(bi)
            if buf is None:
                buf = self._wire[bi] = torch.empty(e - s, dtype=torch.bfloat16, device=self.arena.flat.device)
            buf.copy_(self.arena.flat[s:e])
            payload = buf
        else:
            payload = self.arena.flat[s:e]
        if self.timeline:
            if not self.works and self._stamps and all(st[3] is not None for st in self._stamps):
                self._stamps = []         # first launch of a step: the previous step's stamps go (a caller that never calls
                                          # mark_step_start must not grow the list; timeline_ms() then has no origin and returns None)
            self._stamps.append([bi, payload.numel() * payload.element_size(), self._now(), None])
        self.works.append((bi, self.dist.all_reduce(payload, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True)))

    def _release(self, bi):
        """Bucket ``bi`` is complete.  Collectives are issued strictly in bucket-index order (bucket i waits for 0..i-1): ranks
        whose backward completes buckets in a different order would otherwise pair mismatched all-reduces."""
        self.ready[bi] = True
        if self._releasing:          # called from inside a launch's queue flush: the outer loop picks the bucket up
            return
        self._releasing = True
        try:
            while self._next < len(self.buckets) and (self.ready[self._next] or self.launched[self._next]):
                if not self.launched[self._next]:
                    self._launch(self._next)
                self._next += 1
        finally:
            self._releasing = False

    def _done(self, i, bi):
        left = self.left[bi]
        if i in left:
            left.discard(i)
            if not left and not self.launched[bi]:
                self._release(bi)
        elif self.launched[bi] and not self._finishing and not self._in_launch:
            # a member of a bucket that is already being exchanged was written AGAIN in this step (a packed projection whose backward
            # ran twice: module reuse, a forward_step-style loop): its first report released the bucket too early and the all-reduce
            # raced the second accumulation.  Wrong gradients must not be silent.
            raise RuntimeError("GradReducer: gradient %r was reported ready again after its bucket %d had been launched — a parameter "
                               "view written by more than one backward kernel per step cannot be exchanged in overlap mode; use "
                               "overlap=False (exchange after backward)" % (self.arena.names[i], bi))

    def _ready_ptr(self, ptr, numel=None, kind=None):
        """a kernel has just finished (enqueued) writing the arena gradient view at ``ptr`` — only packed views are tracked here"""
        for i, bi in self._by_pack.get((kind, ptr, numel), ()):
            self._done(i, bi)

    def _make_hook(self, i, bi):
        from . import ops

        def hook(param):
            if not ops.HOOKS_PAUSED[0]:          # (a warm-up / capture pass of a replayed part of the step: clip_graphs._capture)
                self._done(i, bi)
        return hook

    CLIP_SIDE = ("video_embeddings.", "encoder.", "token_type_embeddings.")

    def start_early(self):
        """Two-phase backward (svpc_amd/graph.py): launch, without waiting, every bucket none of whose members belongs to the clip
        encoder — their gradients are final once the text-side backward has been enqueued.  (Index order within the subset; every
        rank selects the same subset from the same arena layout.)"""
        if not self.active:
            return
        for bi, (_, _, members) in enumerate(self.buckets):
            if self.launched[bi]:
                continue
            if not any(self.arena.names[i].startswith(self.CLIP_SIDE) for i in members):
                self._launch(bi)

    def finish(self):
        """Launch whatever has not been launched (tensors without a gradient this step never fire a hook) and wait."""
        if self.active:
            self._finishing = True       # (the queue flush of the launches below reports the remaining gradients: expected here)
            for bi in range(len(self.buckets)):
                if not self.launched[bi]:
                    self._launch(bi)
            for k, (bi, w) in enumerate(self.works):
                w.wait()
                if self.wire_dtype == "bf16":
                    s, e, _ = self.buckets[bi]
                    self.arena.flat[s:e].copy_(self._wire[bi])      # the summed bucket back into the fp32 arena (before the global clip)
                if self.timeline:
                    for st in self._stamps:
                        if st[0] == bi and st[3] is None:
                            st[3] = self._now()
                            break
        self.reset()
