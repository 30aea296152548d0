"""
Frame-to-frame tracking by IoU stitching on the GPU — the working stand-in for the reference's `stitch` tracker.

Reference (src/aliby/track/trackers.py:14-90, dispatch.py:8-13): `stitch_rois(masks, track_info)` receives, per tile, the
label images of the last two timepoints (wired by pipe_core.py:195-200) and the previous call's result, relabels the
older frame with its tracked labels (`update_labels`) and lets `cellpose.utils.stitch3D` carry labels over to the newer
frame.  As shipped it cannot be imported (`agora.utils.masks.labels_from_masks` does not exist) and cellpose is not
vendored, so this module keeps the INTERFACE — same arguments, same `{tile: {"labels": [...], "max_label": n}}` result,
`labels[i]` = tracked label of the newer frame's object i+1 — and implements the stitch3D rule in one batched HIP call
(`aliby_track_stitch`, csrc/track.hip).  One deliberate difference: new labels continue from the tile's running
`max_label`, so the identity of a vanished cell is never reused (oracle/track_restated.py documents the rule).
"""

from __future__ import annotations

import numpy as np
import torch

from aliby_amd import devcache
from aliby_amd.extraction.engine import FeatureEngine, to_device_u16


class TrackResult(dict):
    """{tile: {"labels": [...], "max_label": n}}.  The engine hands a dict-valued step output to its consumers by
    keyword (pipe_core.py:203-205 `last_value[kwd]`): asking this one for "track_info" returns the whole mapping, so
    `passed_data["track"] = [("masks", "segment_x"), ("track_info", "track")]` feeds the tracker its own last result."""

    def __missing__(self, key):
        if key == "track_info":
            return self
        raise KeyError(key)


class StitchTracker:
    """Callable with the reference's `stitch_rois` signature; keeps nothing between calls (the state travels in
    `track_info`, as in the reference)."""

    def __init__(self, stitch_threshold: float = 0.25, engine: FeatureEngine | None = None):
        self.stitch_threshold = float(stitch_threshold)
        self._eng = engine

    @property
    def eng(self) -> FeatureEngine:
        if self._eng is None:
            self._eng = FeatureEngine()
        return self._eng

    @staticmethod
    def _dev(a):
        hit = devcache.lookup(a) if isinstance(a, np.ndarray) else None
        return hit[0] if hit is not None else to_device_u16(np.asarray(a))

    def __call__(self, masks, track_info=None):
        """masks[k] = (older, newer) label images of tile k; track_info = the previous call's return value (or None /
        empty on the first call)."""
        pairs = []
        if len(masks) and all(len(pair) == 1 for pair in masks):
            # first timepoint: nothing to stitch against, every object keeps its own label
            first = [self._dev(pair[0]) for pair in masks]
            if any(a.ndim != 2 for a in first):
                # (a monotile segmenter without per_tile=True hands on ONE [Y,X] image, which the engine's regrouping takes for Y
                # tiles of one row each: pipe_core.py:195-200 — the tracker wants the per-tile list)
                raise AssertionError("Masks are in wrong dimensions")
            cur = torch.stack(first).contiguous()
            tc = self.eng.object_table(cur)
            out = TrackResult()
            for k in range(len(masks)):
                rows = tc.host[tc.offsets[k] : tc.offsets[k + 1]]
                out[k] = {"labels": [int(r["label"]) if r["area"] > 0 else 0 for r in rows], "max_label": int(len(rows))}
            return out
        for pair in masks:
            a, b = self._dev(pair[0]), self._dev(pair[1])
            if a.ndim != 2 or b.ndim != 2 or a.shape != b.shape:
                raise AssertionError("Masks are in wrong dimensions")
            pairs.append((a, b))
        if not pairs:
            return TrackResult()
        prev = torch.stack([p[0] for p in pairs]).contiguous()
        cur = torch.stack([p[1] for p in pairs]).contiguous()
        eng = self.eng
        tp, tc = eng.object_table(prev), eng.object_table(cur)
        prev_tracked = max_in = None
        if track_info:
            flat = np.zeros(max(tp.n_obj, 1), np.int32)
            max_in = np.zeros(len(pairs), np.int32)
            for k in range(len(pairs)):
                info = track_info[k]
                lab = np.asarray(info["labels"], dtype=np.int32)
                n_k = int(tp.offsets[k + 1] - tp.offsets[k])
                if lab.size < n_k:
                    raise ValueError(f"tile {k}: {n_k} objects in the older frame but {lab.size} tracked labels")
                flat[tp.offsets[k] : tp.offsets[k] + n_k] = lab[:n_k]
                max_in[k] = int(info["max_label"])
            prev_tracked = torch.from_numpy(flat).to(prev.device)
        tracked, max_out = eng.track_stitch(prev, cur, tp, tc, prev_tracked, max_in, self.stitch_threshold)
        host = eng.to_host(tracked) if tc.n_obj else np.zeros(0, np.int32)
        return TrackResult(
            (k, {"labels": [int(v) for v in host[tc.offsets[k] : tc.offsets[k + 1]]], "max_label": int(max_out[k])})
            for k in range(len(pairs))
        )


def dispatch_tracker(kind: str = "stitch", **kwargs):
    """dispatch.py:8-28: only the mask-based `stitch` tracker is meaningful here (BABY's comes from its own server)."""
    if kind == "stitch":
        return StitchTracker(**{k: v for k, v in kwargs.items() if k in ("stitch_threshold", "engine")})
    raise Exception("A tracker must be defined.")
