"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's clip feature windowing, the only arithmetic of the input
pipeline.  Parity status: PINNED — checked against outputs of the reference's own functions run in this
container (oracle/make_golden_input.py → tests/golden/input_pipeline.npz).

Follows src/rtransformer/recursive_caption_dataset.py:
  :380-387  _convert_to_feat_index_st_ed   wall-clock [st, ed] seconds → inclusive feature-row window
  :389-416  _load_indexed_video_feature    window → (Lv+Lt, F) matrix laid out [CLS] [VID]…[VID] [SEP] [PAD]…, tokens, mask
  :187-189  feature = concat(resnet (·,2048), bn (·,1024)) along the channel axis
Only product code under svpc_amd/ is shipped; nothing here is imported by it.
"""
import math

import numpy as np


def window(feat_len, timestamp, frm2sec):
    """:380-387 — st = floor(t0 / frm2sec), ed = ceil(t1 / frm2sec), ed ≤ feat_len-1, st ≤ ed-1."""
    st = int(math.floor(timestamp[0] / frm2sec))
    ed = int(math.ceil(timestamp[1] / frm2sec))
    ed = min(ed, feat_len - 1)
    st = min(st, ed - 1)
    return st, ed


def frame_rows(feat_len, timestamp, frm2sec, max_v_len):
    """:398-415 — the feature rows that fill positions 1.. of the clip: the whole window when it has ≤ max_v_len-2 rows,
    else max_v_len-2 rows at linspace(st, ed) truncated to integers."""
    cap = max_v_len - 2
    st, ed = window(feat_len, timestamp, frm2sec)
    if ed - st + 1 > cap:
        return [int(v) for v in np.linspace(st, ed, cap, endpoint=True).astype(np.int64)]
    return list(range(st, ed + 1))


def clip_matrix(raw_feat, timestamp, frm2sec, max_v_len, max_t_len, tokens=(1, 3, 2, 0)):
    """:389-416 → (feat (Lv+Lt, F) float64 as the reference builds it, token ids of the video half, mask)."""
    cls_id, vid_id, sep_id, pad_id = tokens
    rows = frame_rows(len(raw_feat), timestamp, frm2sec, max_v_len)
    n = len(rows)
    feat = np.zeros((max_v_len + max_t_len, raw_feat.shape[1]))
    feat[1:n + 1] = raw_feat[rows]
    cap = max_v_len - 2
    ids = [cls_id] + [vid_id] * n + [sep_id] + [pad_id] * (cap - n)
    mask = [1] * (n + 2) + [0] * (cap - n)
    return feat, ids, mask
