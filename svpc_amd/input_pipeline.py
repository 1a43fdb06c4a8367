"""Input staging for the MI355X training / decoding step.

reference: src/rtransformer/recursive_caption_dataset.py:187-189 (per-video feature = resnet ‖ bn), :380-416 (wall-clock window
→ feature rows, down-sampling, [CLS] [VID]… [SEP] [PAD]… layout), :536-575 (collate: pad every video to the batch's max step
count), src/train.py:91-112 (H2D copy of the collated batch).

The reference materialises, per clip, a float64 (Lv+Lt, F) matrix on the host, collates S×N of them and copies 288 MB per batch
over PCIe.  Here the feature matrices of the whole corpus stay **resident in HBM** (YouCook2: ≈12 GB of 288) as one row bank; a
batch is described by ≈N·S·Lv int32 row indices computed on the host (the only arithmetic: the window / down-sample rule), which
travel through a pinned staging buffer on a copy stream; one HIP gather kernel then lays the frame windows out in the
(S, N, Lv+Lt, F) buffer whose slices are the ``video_features_list`` the model consumes in place, and a second kernel writes the
video half of ``input_ids`` / ``input_mask`` from the valid-frame counts.  Two slots alternate, so batch t+1 is staged while
step t computes.  There is no CPU fallback: staging needs the HIP library.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _lib
from .synthetic import CLS, PAD, SEP, VID

IGNORE = -1


# ------------------------------------------------------------------------------------------------ index arithmetic (host, ints)
def frame_window(feat_len, timestamp, frm2sec):
    """(st, ed) inclusive feature-row window of a clip — recursive_caption_dataset.py:380-387."""
    st = int(math.floor(timestamp[0] / frm2sec))
    ed = int(math.ceil(timestamp[1] / frm2sec))
    ed = min(ed, feat_len - 1)
    st = min(st, ed - 1)
    if not (st <= ed <= feat_len):
        raise AssertionError("st {} <= ed {} <= feat_len {}".format(st, ed, feat_len))      # same failure as the reference (:386)
    return st, ed


def clip_frame_rows(feat_len, timestamp, frm2sec, max_v_len):
    """Feature rows placed at positions 1..n of a clip — :398-415: the whole window if it has ≤ max_v_len-2 rows, otherwise
    max_v_len-2 rows at ``linspace(st, ed)`` truncated towards zero."""
    cap = max_v_len - 2
    st, ed = frame_window(feat_len, timestamp, frm2sec)
    if ed - st + 1 > cap:
        rows = np.linspace(st, ed, cap, endpoint=True).astype(np.int64)
        if int(rows.max()) >= feat_len:
            raise AssertionError("down-sampled index beyond the feature matrix")                 # :405
        return rows
    return np.arange(st, ed + 1, dtype=np.int64)


# ------------------------------------------------------------------------------------------------ HBM-resident feature bank
class FeatureBank:
    """All videos' (frames, F) feature matrices as one (Σ frames + 1, F) fp32 tensor in HBM; the last row is zeros."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self._pending, self.offset, self.length = [], {}, {}
        self.rows = 0
        self.table = None

    def add(self, name, resnet=None, bn=None, feature=None):
        """One video: either its concatenated feature matrix or the (frames, 2048) / (frames, 1024) pair (:187-189)."""
        if feature is None:
            feature = np.concatenate([resnet, bn], axis=1)
        feature = np.ascontiguousarray(feature, dtype=np.float32)
        self.offset[name], self.length[name] = self.rows, feature.shape[0]
        self.rows += feature.shape[0]
        self._pending.append(feature)
        self.table = None

    def add_from_dir(self, feature_dir, names):
        for n in names:
            self.add(n, np.load(os.path.join(feature_dir, "{}_resnet.npy".format(n))),
                     np.load(os.path.join(feature_dir, "{}_bn.npy".format(n))))

    def finalize(self):
        """Upload (pinned staging, one copy per video) and free the host copies."""
        F = self._pending[0].shape[1]
        self.table = torch.zeros(self.rows + 1, F, dtype=torch.float32, device=self.device)
        o = 0
        for f in self._pending:
            t = torch.from_numpy(f)
            if self.device.type == "cuda":
                t = t.pin_memory()
            self.table[o:o + f.shape[0]].copy_(t, non_blocking=True)
            o += f.shape[0]
        if self.device.type == "cuda":
            torch.cuda.current_stream().synchronize()
        self._pending = []
        return self

    @property
    def width(self):
        return self.table.shape[1]


# ------------------------------------------------------------------------------------------------ per-batch staging
class _Slot:
    def __init__(self, S, N, L, F, Lv, device):
        pin = device.type == "cuda"
        self.feats = torch.empty(S, N, L, F, dtype=torch.float32, device=device)
        self.ids = torch.empty(S, N, L, dtype=torch.int64, device=device)
        self.mask = torch.empty(S, N, L, dtype=torch.float32, device=device)
        self.labels = torch.empty(S, N, L, dtype=torch.int64, device=device)
        self.row_idx = torch.empty(S * N * L, dtype=torch.int32, device=device)
        self.n_valid = torch.empty(S * N, dtype=torch.int32, device=device)
        mk = (lambda *sh, dt: torch.empty(*sh, dtype=dt).pin_memory()) if pin else (lambda *sh, dt: torch.empty(*sh, dtype=dt))
        self.h_row_idx = mk(S * N * L, dt=torch.int32)
        self.h_n_valid = mk(S * N, dt=torch.int32)
        self.h_text_ids = mk(S, N, L - Lv, dt=torch.int64)
        self.h_text_mask = mk(S, N, L - Lv, dt=torch.float32)
        self.h_labels = mk(S, N, L, dt=torch.int64)
        self.ready = torch.cuda.Event() if pin else None
        self.consumed = None


class ClipStager:
    """Builds the model's per-step input lists for a batch of videos from the feature bank.

    ``stage(examples)`` — one dict per video: ``name``, ``timestamps`` [[st, ed] seconds per step], ``frm2sec``, and the
    tokenised text half per step: ``text_ids`` / ``text_mask`` / ``text_labels`` arrays of shape (S_b, Lt) (tokenisation is string
    work and stays upstream).  Returns a dict with ``video_features_list``, ``input_ids_list``, ``input_masks_list``,
    ``token_type_ids_list``, ``input_labels_list`` (S tensors each, views of one buffer per kind) and ``batch_step_num``.
    Padded steps (videos shorter than the batch maximum) get zero features, PAD ids, mask 0 and IGNORE labels — the model
    never reads them (model.py:1038-1042); the reference fills them with a copy of the first clip (:561-566)."""

    def __init__(self, bank, max_v_len, max_t_len, n_slots=2):
        assert bank.table is not None, "finalize() the FeatureBank first"
        self.bank, self.Lv, self.Lt = bank, max_v_len, max_t_len
        self.device = bank.device
        self.n_slots = n_slots
        self._slots, self._turn = {}, 0
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _slot(self, S, N):
        key = (S, N, self._turn % self.n_slots)
        self._turn += 1
        sl = self._slots.get(key)
        if sl is None:
            sl = self._slots[key] = _Slot(S, N, self.Lv + self.Lt, self.bank.width, self.Lv, self.device)
        return sl

    def index_batch(self, examples):
        """Host part: → (row_idx (S, N, L) int32 bank rows or -1, n_valid (S, N) int32 (-1 = padded step), step counts)."""
        N = len(examples)
        steps = [len(e["timestamps"]) for e in examples]
        S, L = max(steps), self.Lv + self.Lt
        row_idx = np.full((S, N, L), -1, dtype=np.int32)
        n_valid = np.full((S, N), -1, dtype=np.int32)
        for b, e in enumerate(examples):
            off, flen = self.bank.offset[e["name"]], self.bank.length[e["name"]]
            for s_, ts in enumerate(e["timestamps"]):
                rows = clip_frame_rows(flen, ts, e["frm2sec"], self.Lv)
                row_idx[s_, b, 1:1 + len(rows)] = off + rows
                n_valid[s_, b] = len(rows)
        return row_idx, n_valid, steps

    def stage(self, examples):
        if self.device.type != "cuda":
            raise _lib.SvpcKernelError("svpc_amd.input_pipeline: staging runs on the GPU (no CPU fallback); index_batch() is the host part")
        row_idx, n_valid, steps = self.index_batch(examples)
        S, N, L = row_idx.shape
        Lv = self.Lv
        sl = self._slot(S, N)
        cuda = self.device.type == "cuda"
        if cuda and sl.consumed is not None:
            sl.consumed.synchronize()          # the step that used this slot last has finished reading it
        sl.h_row_idx.copy_(torch.from_numpy(row_idx.reshape(-1)))
        sl.h_n_valid.copy_(torch.from_numpy(n_valid.reshape(-1)))
        sl.h_text_ids.fill_(PAD); sl.h_text_mask.zero_(); sl.h_labels.fill_(IGNORE)
        for b, e in enumerate(examples):
            n = steps[b]
            sl.h_text_ids[:n, b] = torch.as_tensor(np.asarray(e["text_ids"]), dtype=torch.int64)
            sl.h_text_mask[:n, b] = torch.as_tensor(np.asarray(e["text_mask"]), dtype=torch.float32)
            sl.h_labels[:n, b, Lv:] = torch.as_tensor(np.asarray(e["text_labels"]), dtype=torch.int64)
        stream = self._copy_stream if cuda else None
        ctx = torch.cuda.stream(stream) if cuda else _Null()
        with ctx:
            sl.row_idx.copy_(sl.h_row_idx, non_blocking=True)
            sl.n_valid.copy_(sl.h_n_valid, non_blocking=True)
            sl.ids[:, :, Lv:].copy_(sl.h_text_ids, non_blocking=True)
            sl.mask[:, :, Lv:].copy_(sl.h_text_mask, non_blocking=True)
            sl.labels.copy_(sl.h_labels, non_blocking=True)
            sp = stream.cuda_stream if cuda else None
            _lib.call("gather_rows_f32", self.bank.table.data_ptr(), sl.row_idx.data_ptr(), sl.feats.data_ptr(), S * N * L,
                      self.bank.width, sp)
            _lib.call("video_tokens", sl.n_valid.data_ptr(), sl.ids.data_ptr(), sl.mask.data_ptr(), S * N, Lv, L, CLS, VID, SEP, PAD, sp)
            if cuda:
                sl.ready.record(stream)
        if cuda:
            torch.cuda.current_stream().wait_event(sl.ready)      # the consumer's stream waits, the host does not
            sl.consumed = torch.cuda.Event()
        tt = torch.cat([torch.zeros(N, Lv, dtype=torch.int64), torch.ones(N, self.Lt, dtype=torch.int64)], 1).to(self.device)
        return dict(video_features_list=[sl.feats[s_] for s_ in range(S)], input_ids_list=[sl.ids[s_] for s_ in range(S)],
                    input_masks_list=[sl.mask[s_] for s_ in range(S)], token_type_ids_list=[tt for _ in range(S)],
                    input_labels_list=[sl.labels[s_] for s_ in range(S)], batch_step_num=steps, _slot=sl)

    @staticmethod
    def release(batch):
        """Call after the step that consumed ``batch`` has been enqueued: marks the slot reusable once that work completes."""
        sl = batch.get("_slot")
        if sl is not None and sl.consumed is not None:
            sl.consumed.record(torch.cuda.current_stream())


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def balance_by_steps(step_counts, world_size):
    """Assign the videos of a GLOBAL batch to data-parallel ranks so that every rank gets the same number of videos and nearly the
    same number of clips (work ∝ Σ S_b: valid clips are the rows every stage runs over; padded steps cost nothing here).

    The reference batches whatever the sampler yields (``recursive_caption_dataset.py:528-576`` pads every video to the longest of
    the batch); real YouCook2 videos have 3-16 steps, so with one process per GPU an unlucky rank would hold the step's long videos and
    every other rank would wait for it at the gradient all-reduce ("sort/bucket by S if real data is used").
    Longest-processing-time greedy with a per-rank capacity: videos in decreasing S, each to the rank with the fewest clips so far
    that still has room.  → list of ``world_size`` index lists (positions into ``step_counts``), each of len(step_counts)//world_size.
    Deterministic (ties by index), so every rank computes the same assignment from the same global list."""
    n = len(step_counts)
    if world_size <= 0 or n % world_size:
        raise ValueError("balance_by_steps: %d videos cannot be split evenly over %d ranks" % (n, world_size))
    cap = n // world_size
    order = sorted(range(n), key=lambda i: (-int(step_counts[i]), i))
    ranks = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min((r for r in range(world_size) if len(ranks[r]) < cap), key=lambda r: (load[r], r))
        ranks[r].append(i)
        load[r] += int(step_counts[i])
    return [sorted(v) for v in ranks]
