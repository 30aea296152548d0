"""
TEST INFRASTRUCTURE — restatement of cp_measure's "granularity" feature (CellProfiler MeasureGranularity), bound like every
core measurement at src/extraction/core/functions/loaders.py:71-73 (not in the builder's default list, pipe_builder.py:49-56).

cp_measure 0.1.17 is not available: PARITY UNPINNED as a whole.  The primitives it is written in ARE pinned: grey erosion /
dilation with a disk footprint and morphological reconstruction by dilation against scikit-image 0.18.3, bilinear
`map_coordinates` is SciPy's own (tests/golden/skimage_granularity.json, tests/test_oracle_golden.py).

CellProfiler's algorithm (granular spectrum of Matlab's `granspectr`), per image:
  1. subsample by `subsample_size` (0.25): map_coordinates(order=1) at (i, j) / subsample_size;
  2. background: subsample again by `image_sample_size` (0.25), grey erosion then dilation with disk(`element_size` = 10),
     bilinear resize back, subtract, clamp at 0;
  3. `granular_spectrum_length` (16) times: erode with disk(1), reconstruct by dilation under the background-subtracted
     image, resize the reconstruction to the original shape (bilinear) and take the mean under every object;
        Granularity_i = (mean_{i-1} - mean_i) * 100 / max(mean_0, eps),   mean_0 = mean of the ORIGINAL pixels under the object.

Two conventions cannot be established offline and are keyword arguments (CellProfiler's defaults):
  image_mask = "frame": the image mask is the whole frame (CellProfiler without a masking module).  "objects": the mask is
               labels > 0 — then `mask_order` (3 = map_coordinates' default spline order, as in CellProfiler's un-annotated
               call; 1 = bilinear) decides which subsampled pixels count.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


def disk(radius):
    r = int(radius)
    yy, xx = np.mgrid[-r : r + 1, -r : r + 1]
    return (yy * yy + xx * xx) <= r * r


def reconstruction_by_dilation(seed, mask, footprint):
    """skimage.morphology.reconstruction(seed, mask, method='dilation', footprint): the largest image <= mask reachable from
    seed by geodesic dilations.  Plain fixed-point iteration (pixels outside the frame never contribute)."""
    rec = np.minimum(seed, mask).astype(np.float64)
    while True:
        grown = np.minimum(ndi.grey_dilation(rec, footprint=footprint, mode="constant", cval=-np.inf), mask)
        if np.array_equal(grown, rec):
            return rec
        rec = grown


def names(granular_spectrum_length=16):
    return [f"Granularity_{i}" for i in range(1, granular_spectrum_length + 1)]


def _resize_coords(src_shape, dst_shape):
    i, j = np.mgrid[0 : dst_shape[0], 0 : dst_shape[1]].astype(float)
    i *= float(src_shape[0] - 1) / float(dst_shape[0] - 1)
    j *= float(src_shape[1] - 1) / float(dst_shape[1] - 1)
    return i, j


def get_granularity(masks, pixels, subsample_size=0.25, image_sample_size=0.25, element_size=10, granular_spectrum_length=16,
                    image_mask="frame", mask_order=3):
    labels = np.asarray(masks)
    orig = np.asarray(pixels).astype(np.float64)
    n = int(labels.max()) if labels.size else 0
    idx = np.arange(1, n + 1)
    out = {k: np.full(n, np.nan) for k in names(granular_spectrum_length)}
    if n == 0:
        return out
    mask = np.ones(orig.shape, bool) if image_mask == "frame" else labels > 0
    # 1. subsample
    new_shape = np.array(orig.shape)
    if subsample_size < 1:
        new_shape = new_shape * subsample_size
        i, j = np.mgrid[0 : new_shape[0], 0 : new_shape[1]].astype(float) / subsample_size
        pix = ndi.map_coordinates(orig, (i, j), order=1)
        msk = ndi.map_coordinates(mask.astype(float), (i, j), order=mask_order) > 0.9
    else:
        pix, msk = orig.copy(), mask.copy()
    new_shape = np.array(pix.shape)
    # 2. background
    if image_sample_size < 1:
        back_shape = new_shape * image_sample_size
        i, j = np.mgrid[0 : back_shape[0], 0 : back_shape[1]].astype(float) / image_sample_size
        back = ndi.map_coordinates(pix, (i, j), order=1)
        bmask = ndi.map_coordinates(msk.astype(float), (i, j), order=mask_order) > 0.9
    else:
        back, bmask = pix, msk
    selem = disk(element_size)
    tmp = np.zeros_like(back)
    tmp[bmask] = back[bmask]
    back = ndi.grey_erosion(tmp, footprint=selem)  # skimage.morphology.erosion: ndimage's default 'reflect' border
    tmp = np.zeros_like(back)
    tmp[bmask] = back[bmask]
    back = ndi.grey_dilation(tmp, footprint=selem)
    if image_sample_size < 1:
        back = ndi.map_coordinates(back, _resize_coords(back.shape, pix.shape), order=1)
    pix = pix - back
    pix[pix < 0] = 0
    # 3. granular spectrum
    with np.errstate(invalid="ignore"):
        current = np.array([orig[labels == l].mean() if (labels == l).any() else np.nan for l in idx])
    start = np.maximum(current, np.finfo(float).eps)
    ero = pix.copy()
    ero[~msk] = 0
    footprint = disk(1)
    up = _resize_coords(pix.shape, orig.shape)
    for step in range(1, granular_spectrum_length + 1):
        tmp = np.zeros_like(ero)
        tmp[msk] = ero[msk]
        ero = ndi.grey_erosion(tmp, footprint=footprint)
        rec = reconstruction_by_dilation(ero, pix, footprint)
        rec_full = ndi.map_coordinates(rec, up, order=1)
        with np.errstate(invalid="ignore"):
            new = np.array([rec_full[labels == l].mean() if (labels == l).any() else np.nan for l in idx])
        out[f"Granularity_{step}"] = (current - new) * 100 / start
        current = new
    return out
