"""Host-side descriptors shared by the HIP-backed primitives (svpc_amd/ops.py): activation codes, index
arrays that live both on the host (shape logic) and in HBM (kernel arguments), attention segmentation."""
from __future__ import annotations

import numpy as np
import torch

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3


def _upload(values, dtype, device):
    """host list → device tensor without draining the stream: ``torch.tensor(list, device=cuda)`` copies from pageable memory
    synchronously (the host waits for everything queued before it); a pinned staging tensor and a non-blocking copy do not (the caching
    host allocator keeps the staging block until the copy has run)"""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class Idx:
    """An int32 index array with a host copy (python list) and a lazily created, cached HBM copy.
    Built once per batch shape inside ``BatchPlan`` so that steady-state steps issue no H2D copies."""

    __slots__ = ("host", "_dev")

    def __init__(self, host):
        # (a list of Python ints is taken as it is: the plans build ≈20 of these per batch structure, thousands of entries each)
        self.host = host if (type(host) is list and (not host or type(host[0]) is int)) else [int(v) for v in host]
        self._dev = {}

    def __len__(self):
        return len(self.host)

    def dev(self, device):
        key = str(device)
        t = self._dev.get(key)
        if t is None:
            t = _upload(self.host, torch.int32, device)
            self._dev[key] = t
        return t


class FIdx:
    """float32 twin of ``Idx`` (CSR weights)."""

    __slots__ = ("host", "_dev")

    def __init__(self, host):
        self.host = [float(v) for v in host]
        self._dev = {}

    def __len__(self):
        return len(self.host)

    def dev(self, device):
        key = str(device)
        t = self._dev.get(key)
        if t is None:
            t = _upload(self.host, torch.float32, device)
            self._dev[key] = t
        return t


def as_idx(v):
    return v if isinstance(v, (Idx, FIdx)) else Idx(v)


class BulkUpload:
    """Every index array of a batch plan in ONE host→device copy.  A plan holds ≈70 small arrays (row maps, segment tables, CSR tables,
    per-time-step LSTM row lists); created one by one each is a pageable, synchronous ``torch.tensor(..., device=…)`` of ≈30 µs — ≈2.5 ms
    of host time per freshly structured batch (src/train.py:91-132 feeds a new structure every iteration).  Here the arrays are packed
    into one pinned staging slot (a ring of four per device, guarded by events), sent with one asynchronous copy on the current stream,
    and handed out as 16-byte aligned views of the one device buffer."""

    _RING = {}
    SLOT_WORDS = 1 << 19          # 2 MiB per slot

    def __init__(self, device):
        self.device = torch.device(device)
        self.items, self.total = [], 0          # (offset, n, is_float, sink)

    def _reserve(self, arr, is_float, sink):
        n = int(arr.shape[0])
        self.items.append((self.total, n, is_float, sink, arr))
        self.total += (n + 3) & ~3
        return len(self.items) - 1

    def add(self, values, sink=None):
        """int32 array → item id (``flush()[id]`` is its device tensor); ``sink(tensor)`` is called at flush"""
        return self._reserve(np.asarray(values, dtype=np.int32).reshape(-1), False, sink)

    def add_f(self, values, sink=None):
        return self._reserve(np.asarray(values, dtype=np.float32).reshape(-1), True, sink)

    def add_idx(self, idx):
        """an Idx / FIdx whose device copy becomes a view of the bulk buffer"""
        key = str(self.device)
        if isinstance(idx, FIdx):
            return self.add_f(idx.host, sink=lambda t, i=idx, k=key: i._dev.__setitem__(k, t))
        return self.add(idx.host, sink=lambda t, i=idx, k=key: i._dev.__setitem__(k, t))

    def add_seq(self, seq):
        """a SeqInfo built with ``device=None``: its (4, n) table"""
        return self.add(seq.h_q_off + seq.h_q_len + seq.h_k_off + seq.h_k_len, sink=lambda t, s_=seq: setattr(s_, "table", t.view(4, s_.n)))

    def flush(self):
        out = []
        if self.total == 0:
            return out
        if self.device.type != "cuda" or self.total > self.SLOT_WORDS:
            buf = torch.empty(self.total, dtype=torch.int32)
            host = buf.numpy()
        else:
            key = self.device.index
            ring = self._RING.get(key)
            if ring is None:
                ring = self._RING[key] = {"slots": [[torch.empty(self.SLOT_WORDS, dtype=torch.int32).pin_memory(), None] for _ in range(4)], "next": 0}
            slot = ring["slots"][ring["next"]]
            ring["next"] = (ring["next"] + 1) % len(ring["slots"])
            if slot[1] is not None:
                slot[1].synchronize()             # the copy that last read this staging slot (four plans ago) has long finished
            buf = slot[0][:self.total]
            host = buf.numpy()
        for off, n, is_float, _, arr in self.items:
            if is_float:
                host[off:off + n] = arr.view(np.int32)
            else:
                host[off:off + n] = arr
        if self.device.type == "cuda":
            dev = torch.empty(self.total, dtype=torch.int32, device=self.device)
            dev.copy_(buf, non_blocking=True)
            if self.total <= self.SLOT_WORDS:
                ev = torch.cuda.Event()
                ev.record()
                slot[1] = ev
        else:
            dev = buf.clone()
        for off, n, is_float, sink, _ in self.items:
            t = dev[off:off + n]
            if is_float:
                t = t.view(torch.float32)
            if sink is not None:
                sink(t)
            out.append(t)
        return out


class SeqInfo:
    """Segmentation of flat row arrays into attention sequences: sequence i owns query rows
    [q_off[i], q_off[i]+q_len[i]) and key rows [k_off[i], k_off[i]+k_len[i])."""

    def __init__(self, q_off, q_len, k_off, k_len, device="cpu"):
        self.n = len(q_off)
        self.h_q_off, self.h_q_len = [int(v) for v in q_off], [int(v) for v in q_len]
        self.h_k_off, self.h_k_len = [int(v) for v in k_off], [int(v) for v in k_len]
        packed = self.h_q_off + self.h_q_len + self.h_k_off + self.h_k_len
        # device=None: the table is uploaded later with the rest of a plan's arrays (BulkUpload.add_seq)
        self.table = torch.tensor(packed, dtype=torch.int32, device=device).view(4, self.n) if device is not None else None
        self.max_q, self.max_k = max(self.h_q_len), max(self.h_k_len)
        self.n_q_rows = max(o + l for o, l in zip(self.h_q_off, self.h_q_len))
        self.n_k_rows = max(o + l for o, l in zip(self.h_k_off, self.h_k_len))

    @classmethod
    def uniform(cls, n, lq, lk, device="cpu"):
        return cls([i * lq for i in range(n)], [lq] * n, [i * lk for i in range(n)], [lk] * n, device)
