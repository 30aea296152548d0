"""
The reference's in-repo functions that are not reached through the feature tree, with their own signatures:

  ratio(cell_mask, trap_image)            src/extraction/core/functions/cell.py:268-279
  imBackground(cell_masks, trap_image)    src/extraction/core/functions/trap.py:6-23
  background_max5(cell_masks, trap_image) src/extraction/core/functions/trap.py:26-43

Host arrays in, Python floats out, the arithmetic in csrc/feat_extra.hip (batched forms: FeatureEngine.cell_ratio /
FeatureEngine.trap_background, which take a whole [F,Y,X] label stack).  No CPU fallback.
"""

from __future__ import annotations

import numpy as np
import torch

from aliby_amd.extraction.engine import FeatureEngine, to_device_planes, to_device_u16


def _as_pixels(a):
    """Float arrays that hold uint16-valued pixels go up as uint16: the kernels then divide / order them exactly as NumPy
    does in float64; anything else is float32 on the device."""
    a = np.asarray(a)
    if a.dtype.kind == "f" and a.size and np.isfinite(a).all() and a.min() >= 0 and a.max() <= 65535 and (a == np.rint(a)).all():
        return a.astype(np.uint16)
    return a


def ratio(cell_mask, trap_image) -> float:
    """Median ratio between two fluorescence channels of one cell; NaN unless trap_image is [Y, X, 2] (cell.py:270-279)."""
    trap_image = _as_pixels(trap_image)
    if not (trap_image.ndim == 3 and trap_image.shape[-1] == 2):
        return float("nan")
    eng = FeatureEngine()
    labels = to_device_u16(np.asarray(cell_mask, dtype=bool).astype(np.uint16)[None])
    planes, dt = to_device_planes(np.moveaxis(trap_image, -1, 0)[None])  # [1, 2, Y, X]
    table = eng.object_table(labels)
    if table.n_obj == 0:
        return float("nan")  # np.median of an empty selection
    return float(eng.cell_ratio(labels, planes, dt, 0, 1, table)[0])


def _background(cell_masks, trap_image):
    trap_image = _as_pixels(trap_image)
    if not len(cell_masks):
        covered = np.zeros(trap_image.shape, bool)  # "create cell_masks if none are given"
    else:
        covered = np.asarray(cell_masks).sum(axis=2).astype(bool)
    eng = FeatureEngine()
    labels = to_device_u16(covered.astype(np.uint16)[None])
    planes, dt = to_device_planes(trap_image[None, None])
    return eng.trap_background(labels, planes, dt, 0)[0]


def imBackground(cell_masks, trap_image) -> float:
    """Median of the pixels not comprising cells (cell_masks [Y, X, N], one mask per cell)."""
    return float(_background(cell_masks, trap_image)[0])


def background_max5(cell_masks, trap_image) -> float:
    """Mean of the maximum five pixels of the background."""
    return float(_background(cell_masks, trap_image)[1])
