"""
The reference's in-repo functions that are not reached through the feature tree, with their own signatures:

  ratio(cell_mask, trap_image)            src/extraction/core/functions/cell.py:268-279
  imBackground(cell_masks, trap_image)    src/extraction/core/functions/trap.py:6-23
  background_max5(cell_masks, trap_image) src/extraction/core/functions/trap.py:26-43

  reduce_z(trap_image, fun, axis=0)       src/extraction/core/functions/distributors.py:6-24

Host arrays in, Python floats (reduce_z: a NumPy array of NumPy's own result dtype) out, the arithmetic in csrc/feat_extra.hip (batched forms: FeatureEngine.cell_ratio /
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


_UFUNCS = {"maximum": 0, "add": 1, "divide": 2, "true_divide": 2}


def reduce_z(trap_image, fun, axis: int = 0):
    """`fun.reduce(trap_image, axis=axis)` for the reducers the reference registers (loaders.py:110-127: np.maximum, np.add,
    np.divide) with NumPy's result dtypes — uint16 input: uint16 / uint64 / float64; float32 input: float32 — and the
    reference's error for anything that is not a ufunc (distributors.py:20-24).  A 2-D image comes back as it is."""
    from aliby_amd import _lib
    from aliby_amd.extraction.engine import _ptr, _stream_ptr

    if isinstance(fun, np.ufunc):
        op = _UFUNCS.get(fun.__name__)
        if op is None:
            raise NotImplementedError(f"reduce_z: ufunc {fun.__name__} is not one of the reducers the pipeline registers (max / add / div)")
    else:
        raise Exception(f"Operator {fun} is an invalid reducer.")
    a = np.asarray(trap_image)
    if a.ndim <= 2:
        return a
    if a.dtype not in (np.uint16, np.float32):
        a = a.astype(np.uint16 if a.dtype in (np.uint8, np.bool_) else np.float32)
    moved = np.ascontiguousarray(np.moveaxis(a, axis, 0))  # [Z, ...]
    Z, rest = moved.shape[0], moved.shape[1:]
    eng = FeatureEngine()
    dev = torch.from_numpy(moved).cuda()
    u16 = moved.dtype == np.uint16
    in_dt = _lib.U16 if u16 else _lib.F32
    out_dt, t_dt = ((in_dt, dev.dtype) if op == 0 else (_lib.F32, torch.float32) if not u16
                    else (_lib.U64, torch.uint64) if op == 1 else (_lib.F64, torch.float64))
    out = torch.empty(rest, dtype=t_dt, device="cuda")
    _lib.check(eng.lib.aliby_reduce_z(eng.ctx.handle, _ptr(dev), in_dt, 1, Z, int(np.prod(rest, dtype=np.int64)), op, _ptr(out), out_dt,
                                      _stream_ptr()))
    return out.cpu().numpy()
