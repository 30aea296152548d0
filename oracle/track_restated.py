"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame-to-frame IoU stitcher.

The reference's `stitch` tracker (src/aliby/track/trackers.py:14-90) wraps `cellpose.utils.stitch3D` but cannot be
imported (it needs `agora.utils.masks.labels_from_masks`, which does not exist: SURVEY.md §2 row 15), and cellpose
4.0.6 is not vendored, so parity here is UNPINNED: this restates the published stitch3D rule for one pair of frames

    iou[cur, prev] = |cur ∩ prev| / (|cur| + |prev| - |cur ∩ prev|)           (float64)
    iou[iou < stitch_threshold] = 0 ; iou[iou < iou.max(axis=0)] = 0            (column-wise winners, ties kept)
    matched: label = argmax over prev (first maximum = smallest previous label) ; unmatched: a new label

with the previous frame carrying its TRACKED labels (trackers.py:70-71 `update_labels`), and with one deliberate,
documented difference: new labels continue from the running `max_label` of the tile instead of from the largest
label present in the previous frame, so the identity of a cell that disappeared is never handed to a new one.
"""

import numpy as np


def stitch_pair(prev, cur, prev_tracked=None, max_label=None, stitch_threshold=0.25):
    """prev, cur: int label images [Y,X] with labels 1..n.  prev_tracked[i] = tracked label of previous object i+1
    (None: its own label).  Returns (tracked label of every current object 1..n_cur as int64 [n_cur], new max_label)."""
    prev = np.asarray(prev).astype(np.int64)
    cur = np.asarray(cur).astype(np.int64)
    n_prev, n_cur = int(prev.max(initial=0)), int(cur.max(initial=0))
    if prev_tracked is None:
        prev_tracked = np.arange(1, n_prev + 1, dtype=np.int64)
    prev_tracked = np.asarray(prev_tracked, dtype=np.int64)
    if max_label is None:
        max_label = int(prev_tracked.max(initial=0))
    max_label = max(int(max_label), int(prev_tracked.max(initial=0)))
    out = np.zeros(n_cur, np.int64)
    if n_cur == 0:
        return out, max_label
    overlap = np.zeros((n_cur + 1, n_prev + 1), np.int64)
    np.add.at(overlap, (cur.ravel(), prev.ravel()), 1)
    area_cur = overlap.sum(axis=1, keepdims=True)
    area_prev = overlap.sum(axis=0, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = overlap / (area_cur + area_prev - overlap)
    iou[np.isnan(iou)] = 0.0
    iou = iou[1:, 1:]
    # columns in TRACKED-label order: stitch3D sees the previous frame relabelled with its tracked labels
    order = np.argsort(prev_tracked, kind="stable") if n_prev else np.zeros(0, np.int64)
    present = area_prev[0, 1:][order] > 0 if n_prev else np.zeros(0, bool)
    iou = iou[:, order]
    if iou.size:
        iou[iou < stitch_threshold] = 0.0
        iou[iou < iou.max(axis=0)] = 0.0
    for i in range(n_cur):
        if area_cur[i + 1, 0] == 0:
            continue  # label missing from the current frame: stays 0
        row = iou[i] if iou.size else np.zeros(0)
        if row.size and row.max() > 0.0:
            out[i] = prev_tracked[order][int(np.argmax(row))]
        else:
            max_label += 1
            out[i] = max_label
    del present
    return out, max_label


def stitch_rois(masks, track_info=None, stitch_threshold=0.25):
    """Reference-shaped entry (trackers.py:14-55): masks[k] = (previous, current) label images of tile k;
    track_info[k] = {"labels": tracked labels of the previous frame's objects, "max_label": int}."""
    result = {}
    for k, pair in enumerate(masks):
        pair = np.asarray(pair)
        assert pair.ndim == 3, "Masks are in wrong dimensions"
        info = (track_info or {}).get(k) if isinstance(track_info, dict) else None
        labels, mx = stitch_pair(pair[0], pair[1], None if info is None else info["labels"],
                                 None if info is None else info["max_label"], stitch_threshold)
        result[k] = {"labels": [int(v) for v in labels], "max_label": int(mx)}
    return result
