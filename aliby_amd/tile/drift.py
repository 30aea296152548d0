"""
Whole-pixel drift between two frames by phase cross-correlation, on the GPU.

Reference: `Tiler.find_drift` (src/aliby/tile/tiler.py:284-307) =
`skimage.registration.phase_cross_correlation(previous, current)` with its defaults; the algorithm is restated in
oracle/drift_restated.py.  Once per timepoint on one 2-D plane: two forward FFTs, a pointwise product (optionally
phase-normalised), one inverse FFT and an argmax — the FFTs are rocFFT through `torch.fft` (float64, like NumPy's),
the rest a handful of pointwise passes; nothing here is on the per-tile hot path.
"""

from __future__ import annotations

import numpy as np
import torch


def phase_cross_correlation(reference_image, moving_image, normalization="phase") -> np.ndarray:
    """-> float64 [ndim] shift (in pixels) to register `moving_image` onto `reference_image`, like skimage's first
    return value with upsample_factor=1.  Inputs: host arrays or device tensors of equal shape."""
    ref = _plane(reference_image)
    mov = _plane(moving_image)
    if ref.shape != mov.shape:
        raise ValueError("images must be same shape")
    prod = torch.fft.fftn(ref) * torch.conj(torch.fft.fftn(mov))
    if normalization == "phase":
        eps = torch.finfo(torch.float64).eps
        prod = prod / torch.clamp(prod.abs(), min=100 * eps)
    elif normalization is not None:
        raise ValueError("normalization must be either phase or None")
    cc = torch.fft.ifftn(prod)
    peak = int(torch.argmax(cc.abs()))  # first maximum in raster order, like numpy.argmax
    maxima = np.array(np.unravel_index(peak, tuple(cc.shape)), dtype=np.float64)
    shape = np.array(cc.shape, dtype=np.float64)
    mid = np.fix(shape / 2)
    wrap = maxima > mid
    maxima[wrap] -= shape[wrap]
    return maxima


def _plane(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a
    else:
        arr = np.asarray(a)
        if arr.dtype == np.uint16:
            arr = arr.astype(np.int32)  # torch has no arithmetic on uint16
        t = torch.from_numpy(np.ascontiguousarray(arr))
    if t.dtype == torch.uint16:
        t = t.to(torch.int32)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("aliby_amd.tile.drift needs a GPU: there is no CPU fallback")
        t = t.cuda()
    return t.to(torch.float64)
