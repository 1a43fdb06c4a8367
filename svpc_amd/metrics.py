"""Training / validation counters kept on the MI355X.

reference: src/train.py:32-38 (cal_performance: correct next-word predictions over labelled positions), :40-49 (calculate_f1:
hits / gold positives / predicted positives of the entity and action probabilities at threshold 0.5), :51-68 (compute_total_f1),
:150-180 (per-step accumulation with ≈10 ``.item()`` host synchronisations).  Here the nine running sums live in one device buffer
updated by two small kernels per video; ``result()`` is the only host read-back.
"""
from __future__ import annotations

import torch

from . import _lib

IGNORE = -1


def compute_total_f1(n_correct, n_recall, n_precision):
    """src/train.py:51-68."""
    recall = 0 if n_recall == 0 else n_correct / n_recall
    precision = 0 if n_precision == 0 else n_correct / n_precision
    f1 = 0 if (recall == 0 and precision == 0) else 2 * (recall * precision) / (recall + precision)
    return {"recall": recall, "precision": precision, "f1": f1}


class TrainMetrics:
    """counters: [n_word, n_word_correct, ent_correct, ent_recall, ent_precision, ac_correct, ac_recall, ac_precision, loss_sum]"""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SvpcKernelError("svpc_amd.metrics: counters live on the GPU (no CPU fallback)")
        self.c = torch.zeros(9, dtype=torch.float64, device=self.device)

    def reset(self):
        self.c.zero_()

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def add_words(self, pred_scores, labels):
        """pred_scores (S, Lt, C) probabilities / logits of one video, labels (S, Lt) int64 with IGNORE = -1."""
        C = pred_scores.shape[-1]
        p = pred_scores.detach().reshape(-1, C)
        if p.stride(1) != 1:
            p = p.contiguous()
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        _lib.call("metric_argmax", p.data_ptr(), p.stride(0), p.shape[0], C, lab.data_ptr(), IGNORE, self.c.data_ptr(), self._stream())

    def add_f1(self, prob, gold, which):
        """which = "entity" | "action"."""
        off = 2 if which == "entity" else 5
        pr = prob.detach().reshape(-1).float().contiguous()
        gd = gold.reshape(-1).float().contiguous()
        _lib.call("metric_f1", pr.data_ptr(), gd.data_ptr(), pr.numel(), self.c.data_ptr() + 8 * off, self._stream())

    def update(self, loss, pred_scores_list, labels_list, entity_prob_list=(), alignments=(), action_prob_list=(), actions=()):
        """One training / validation step (src/train.py:150-171), no host synchronisation."""
        for p, g in zip(pred_scores_list, labels_list):
            self.add_words(p, g)
        for p, g in zip(entity_prob_list, alignments):
            self.add_f1(p, g, "entity")
        for p, g in zip(action_prob_list, actions):
            self.add_f1(p, g, "action")
        if loss is not None:
            self.c[8] += loss.detach().double()

    def result(self):
        """The single host read-back: totals and the derived numbers the reference logs (src/train.py:178-185)."""
        v = [float(x) for x in self.c.cpu()]
        n_word, n_corr = v[0], v[1]
        return dict(n_word_total=n_word, n_word_correct=n_corr, total_loss=v[8],
                    loss_per_word=(v[8] / n_word) if n_word else 0.0, accuracy=(n_corr / n_word) if n_word else 0.0,
                    entity=compute_total_f1(v[2], v[3], v[4]), action=compute_total_f1(v[5], v[6], v[7]),
                    counts=v[:8])
