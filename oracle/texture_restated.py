"""
TEST INFRASTRUCTURE — restatement of cp_measure's "texture" feature (CellProfiler MeasureTexture on
top of mahotas.features.haralick), bound at src/extraction/core/functions/loaders.py:71-73 and listed
in the builder's default features (pipe_builder.py:49-56).

mahotas 1.4.18 (uv.lock:993-994) and cp_measure are not available: PARITY UNPINNED.  Restated:
  * pixels -> 8-bit grey levels the way skimage.util.img_as_ubyte does: uint16 -> v >> 8;
    float in [0,1] -> rint(255 v) (clipped);
  * per object: bbox crop with non-object pixels set to 0 (regionprops `intensity_image`), symmetric
    co-occurrence matrix at distance `scale` for the 4 2-D directions (0,1),(1,1),(1,0),(1,-1),
    zero grey level ignored (row/column 0 cleared), 13 Haralick statistics with mahotas' conventions
    (SumVariance without the "haralick bug", DifferenceVariance = variance of the p_{x-y} VECTOR,
    entropies in bits, matrix side = max grey level in the crop + 1).
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

from oracle.cp_measure_restated import _indices

DELTAS_2D = [(0, 1), (1, 1), (1, 0), (1, -1)]
HARALICK = [
    "AngularSecondMoment", "Contrast", "Correlation", "Variance", "InverseDifferenceMoment", "SumAverage",
    "SumVariance", "SumEntropy", "Entropy", "DifferenceVariance", "DifferenceEntropy", "InfoMeas1", "InfoMeas2",
]


def img_as_ubyte(pixels):
    pixels = np.asarray(pixels)
    if pixels.dtype == np.uint8:
        return pixels
    if pixels.dtype == np.uint16:
        return (pixels >> 8).astype(np.uint8)
    x = np.rint(pixels.astype(np.float64) * 255.0)
    return np.clip(x, 0, 255).astype(np.uint8)


def cooccurence(f, direction, distance):
    fm1 = int(f.max()) + 1
    cmat = np.zeros((fm1, fm1), np.int64)
    dy, dx = DELTAS_2D[direction]
    dy, dx = dy * distance, dx * distance
    h, w = f.shape
    y0, y1 = max(0, -dy), min(h, h - dy)
    x0, x1 = max(0, -dx), min(w, w - dx)
    if y1 > y0 and x1 > x0:
        a = f[y0:y1, x0:x1].ravel()
        b = f[y0 + dy : y1 + dy, x0 + dx : x1 + dx].ravel()
        np.add.at(cmat, (a, b), 1)
    return cmat + cmat.T


def _entropy(p):
    p = p.ravel()
    p1 = p.copy()
    p1 += p == 0
    return -np.dot(np.log2(p1), p)


def haralick_features(cmat):
    cmat = cmat.copy()
    cmat[0] = 0
    cmat[:, 0] = 0
    T = cmat.sum()
    if not T:
        raise ValueError("empty co-occurrence matrix")
    maxv = len(cmat)
    k = np.arange(maxv)
    k2 = k**2
    tk = np.arange(2 * maxv)
    tk2 = tk**2
    p = cmat / float(T)
    pravel = p.ravel()
    px, py = p.sum(0), p.sum(1)
    ux, uy = np.dot(px, k), np.dot(py, k)
    vx, vy = np.dot(px, k2) - ux**2, np.dot(py, k2) - uy**2
    sx, sy = np.sqrt(vx), np.sqrt(vy)
    px_plus_y = np.zeros(2 * maxv)
    px_minus_y = np.zeros(maxv)
    ii, jj = np.nonzero(p)
    np.add.at(px_plus_y, ii + jj, p[ii, jj])
    np.add.at(px_minus_y, np.abs(ii - jj), p[ii, jj])
    feats = np.zeros(13)
    feats[0] = np.dot(pravel, pravel)
    feats[1] = np.dot(k2, px_minus_y)
    feats[2] = 1.0 if (sx == 0.0 or sy == 0.0) else (1.0 / sx / sy) * (np.dot(np.dot(k, p), k) - ux * uy)
    feats[3] = vx
    feats[4] = np.dot(1.0 / (1.0 + k2), px_minus_y)
    feats[5] = np.dot(tk, px_plus_y)
    feats[7] = _entropy(px_plus_y)
    feats[6] = np.dot(tk2, px_plus_y) - feats[5] ** 2
    feats[8] = _entropy(pravel)
    feats[9] = px_minus_y.var()
    feats[10] = _entropy(px_minus_y)
    HX, HY = _entropy(px), _entropy(py)
    cross = np.outer(px, py)
    cross += cross == 0
    cross = cross.ravel()
    HXY1 = -np.dot(pravel, np.log2(cross))
    HXY2 = _entropy(cross)
    feats[11] = (feats[8] - HXY1) if max(HX, HY) == 0.0 else (feats[8] - HXY1) / max(HX, HY)
    feats[12] = np.sqrt(max(0, 1 - np.exp(-2.0 * (HXY2 - feats[8]))))
    return feats


def get_texture(masks, pixels, scale=3, gray_levels=256):
    labels = np.asarray(masks)
    idx = _indices(labels)
    q = img_as_ubyte(pixels)
    if gray_levels != 256:
        # skimage.exposure.rescale_intensity(in_range=(0,255), out_range=(0,gray_levels-1)).astype(uint8)
        q = (q.astype(np.float64) / 255.0 * (gray_levels - 1)).astype(np.uint8)
    out = {f"{h}_{scale}_{d:02d}_{gray_levels}": np.full(len(idx), np.nan) for d in range(4) for h in HARALICK}
    for i, sl in enumerate(ndi.find_objects(labels.astype(np.int32), max_label=len(idx))):
        if sl is None:
            continue
        crop = np.where(labels[sl] == (i + 1), q[sl], 0).astype(np.int64)
        for d in range(4):
            try:
                feats = haralick_features(cooccurence(crop, d, scale))
            except ValueError:
                continue
            for h, v in zip(HARALICK, feats):
                out[f"{h}_{scale}_{d:02d}_{gray_levels}"][i] = v
    return out
