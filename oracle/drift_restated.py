"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the whole-pixel drift estimate.

Reference call site: Tiler.find_drift (src/aliby/tile/tiler.py:284-307) =
skimage.registration.phase_cross_correlation(previous, current) with defaults (upsample_factor=1, space="real").
scikit-image is pinned at 0.26.0 by the reference (uv.lock) and is not importable here; the published algorithm is

    P = fft2(reference) * conj(fft2(moving));  normalization="phase" (the default since 0.19): P /= max(|P|, 100 eps)
    cc = ifft2(P);  peak = argmax |cc| (first maximum in raster order);  shift = peak, minus the axis length where
    peak > fix(axis length / 2)

PINNED for normalization=None against scikit-image 0.18.3 (which has no normalisation step) by
tests/golden/skimage_drift.json; the "phase" branch is restated from the 0.19+ source and UNPINNED.
"""

import numpy as np


def phase_cross_correlation(reference_image, moving_image, normalization="phase"):
    src = np.fft.fftn(np.asarray(reference_image))
    tgt = np.fft.fftn(np.asarray(moving_image))
    prod = src * tgt.conj()
    if normalization == "phase":
        eps = np.finfo(prod.real.dtype).eps
        prod /= np.maximum(np.abs(prod), 100 * eps)
    elif normalization is not None:
        raise ValueError("normalization must be either phase or None")
    cc = np.fft.ifftn(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    shape = np.array(cc.shape)
    mid = np.array([np.fix(s / 2) for s in cc.shape])
    shifts = np.stack(maxima).astype(np.float64)
    shifts[shifts > mid] -= shape[shifts > mid]
    return shifts
