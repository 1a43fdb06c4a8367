"""Host-side descriptors shared by the HIP-backed primitives (svpc_amd/ops.py): activation codes, index
arrays that live both on the host (shape logic) and in HBM (kernel arguments), attention segmentation."""
from __future__ import annotations

import torch

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3


class Idx:
    """An int32 index array with a host copy (python list) and a lazily created, cached HBM copy.
    Built once per batch shape inside ``BatchPlan`` so that steady-state steps issue no H2D copies."""

    __slots__ = ("host", "_dev")

    def __init__(self, host):
        self.host = [int(v) for v in host]
        self._dev = {}

    def __len__(self):
        return len(self.host)

    def dev(self, device):
        key = str(device)
        t = self._dev.get(key)
        if t is None:
            t = torch.tensor(self.host, dtype=torch.int32, device=device)
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
            t = torch.tensor(self.host, dtype=torch.float32, device=device)
            self._dev[key] = t
        return t


def as_idx(v):
    return v if isinstance(v, (Idx, FIdx)) else Idx(v)


class SeqInfo:
    """Segmentation of flat row arrays into attention sequences: sequence i owns query rows
    [q_off[i], q_off[i]+q_len[i]) and key rows [k_off[i], k_off[i]+k_len[i])."""

    def __init__(self, q_off, q_len, k_off, k_len, device="cpu"):
        self.n = len(q_off)
        self.h_q_off, self.h_q_len = [int(v) for v in q_off], [int(v) for v in q_len]
        self.h_k_off, self.h_k_len = [int(v) for v in k_off], [int(v) for v in k_len]
        packed = self.h_q_off + self.h_q_len + self.h_k_off + self.h_k_len
        self.table = torch.tensor(packed, dtype=torch.int32, device=device).view(4, self.n)
        self.max_q, self.max_k = max(self.h_q_len), max(self.h_k_len)
        self.n_q_rows = max(o + l for o, l in zip(self.h_q_off, self.h_q_len))
        self.n_k_rows = max(o + l for o, l in zip(self.h_k_off, self.h_k_len))

    @classmethod
    def uniform(cls, n, lq, lk, device="cpu"):
        return cls([i * lq for i in range(n)], [lq] * n, [i * lk for i in range(n)], [lk] * n, device)
