"""
TEST INFRASTRUCTURE — restatement of centrosome.zernike (zernike / construct_zernike_polynomials /
score_zernike / get_zernike_indexes) and centrosome.cpmorphology.minimum_enclosing_circle, as used by
cp_measure's "zernike" and "radial_zernikes" features (names bound at
src/extraction/core/functions/loaders.py:71-73; default list pipe_builder.py:49-56).

centrosome 1.3.3 (uv.lock:154-155) is not available: PARITY UNPINNED.  The minimum enclosing circle is
unique, so any exact algorithm agrees with centrosome's up to rounding; it is cross-checked against a
brute-force search in tests/test_oracle_units.py.
"""

from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi

from oracle.cp_measure_restated import _hull_ccw, _indices


def get_zernike_indexes(limit=10):
    return np.array([(n, m) for n in range(limit) for m in range(n % 2, n + 1, 2)], dtype=int)


def construct_zernike_lookuptable(zernike_indexes):
    n_max = int(np.max(zernike_indexes[:, 0]))
    factorial = np.ones((1 + n_max,), dtype=float)
    factorial[1:] = np.cumprod(np.arange(1, 1 + n_max, dtype=float))
    width = int(n_max // 2 + 1)
    lut = np.zeros((zernike_indexes.shape[0], width), dtype=float)
    for idx, (n, m) in enumerate(zernike_indexes):
        alt = 1
        npmh, nmmh = (n + m) // 2, (n - m) // 2
        for k in range(0, nmmh + 1):
            lut[idx, k] = (alt * factorial[n - k]) / (factorial[k] * factorial[npmh - k] * factorial[nmmh - k])
            alt = -alt
    return lut


def construct_zernike_polynomials(x, y, zernike_indexes, weight=None):
    """x, y: 1-D coordinates inside the unit circle -> complex array [npts, K]."""
    lut = construct_zernike_lookuptable(zernike_indexes)
    r_square = np.square(x) + np.square(y)
    z = y + 1j * x
    zf = np.zeros((len(zernike_indexes),) + x.shape, complex)
    for idx, (n, m) in enumerate(zernike_indexes):
        s = np.zeros_like(x, dtype=float)
        for k in range((n - m) // 2 + 1):
            s *= r_square
            s += lut[idx, k]
        # pixels defining the enclosing circle sit at r^2 = 1 +- 1ulp: a 1e-9 guard band keeps their
        # membership independent of rounding (centrosome itself tests r_square > 1 on its own rounding)
        s[r_square > 1 + 1e-9] = 0
        if weight is not None:
            s = s * weight.astype(s.dtype)
        zf[idx] = s if m == 0 else s * (z**m)
    return zf.T


def _circle2(a, b):
    c = (a + b) / 2.0
    return c, float(np.hypot(*(a - c)))


def _circle3(a, b, c):
    ax, ay, bx, by, cx, cy = *a, *b, *c
    d = 2.0 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by))
    if d == 0:
        return None
    ux = ((ax * ax + ay * ay) * (by - cy) + (bx * bx + by * by) * (cy - ay) + (cx * cx + cy * cy) * (ay - by)) / d
    uy = ((ax * ax + ay * ay) * (cx - bx) + (bx * bx + by * by) * (ax - cx) + (cx * cx + cy * cy) * (bx - ax)) / d
    ctr = np.array([ux, uy])
    return ctr, float(np.hypot(*(a - ctr)))


def _inside(p, c, r):
    return np.hypot(*(p - c)) <= r * (1 + 1e-12) + 1e-12


def minimum_enclosing_circle_points(pts):
    """Exact smallest enclosing circle (Welzl, move-free incremental form) of float points [K,2]."""
    pts = np.asarray(pts, float)
    k = len(pts)
    if k == 0:
        return np.zeros(2), 0.0
    c, r = pts[0].copy(), 0.0
    for i in range(1, k):
        if _inside(pts[i], c, r):
            continue
        c, r = pts[i].copy(), 0.0
        for j in range(i):
            if _inside(pts[j], c, r):
                continue
            c, r = _circle2(pts[i], pts[j])
            for l in range(j):
                if _inside(pts[l], c, r):
                    continue
                cc = _circle3(pts[i], pts[j], pts[l])
                if cc is not None:
                    c, r = cc
    return c, r


def minimum_enclosing_circle(labels, indexes=None):
    """centres [n,2] as (i, j) and radii [n] of the per-object minimum enclosing circle of the pixel centres."""
    idx = _indices(labels) if indexes is None else np.asarray(indexes)
    centres, radii = np.zeros((len(idx), 2)), np.zeros(len(idx))
    slices = ndi.find_objects(labels.astype(np.int32), max_label=int(idx.max()) if len(idx) else 0)
    for k, lab in enumerate(idx):
        sl = slices[lab - 1]
        if sl is None:
            centres[k], radii[k] = np.nan, np.nan
            continue
        rr, cc = np.nonzero(labels[sl] == lab)
        hv = _hull_ccw(np.stack([rr + sl[0].start, cc + sl[1].start], 1)).astype(float)
        centres[k], radii[k] = minimum_enclosing_circle_points(hv)
    return centres, radii


def _names(prefix, zi):
    return [f"{prefix}_{n}_{m}" for n, m in zi]


def get_zernike(masks, pixels=None, zernike_numbers=9):
    """centrosome.zernike.zernike: |sum of Z_nm over the object's pixels| / (pi r^2)."""
    labels = np.asarray(masks)
    zi = get_zernike_indexes(zernike_numbers + 1)
    idx = _indices(labels)
    centres, radii = minimum_enclosing_circle(labels, idx)
    out = {name: np.full(len(idx), np.nan) for name in _names("Zernike", zi)}
    for k, lab in enumerate(idx):
        ii, jj = np.nonzero(labels == lab)
        if len(ii) == 0:
            continue
        with np.errstate(invalid="ignore", divide="ignore"):
            y = (ii - centres[k, 0]) / radii[k]
            x = (jj - centres[k, 1]) / radii[k]
            zf = construct_zernike_polynomials(x, y, zi)
            area = np.pi * radii[k] ** 2
            for col, name in enumerate(_names("Zernike", zi)):
                v = zf[:, col].sum()
                out[name][k] = np.sqrt(v.real**2 + v.imag**2) / area
    return out


def get_radial_zernikes(masks, pixels, zernike_degree=9):
    """MeasureObjectIntensityDistribution.calculate_zernikes: intensity-weighted moments,
    magnitude = |v| / n_pixels, phase = arctan2(real, imag)."""
    labels = np.asarray(masks)
    img = np.asarray(pixels, dtype=float)
    zi = get_zernike_indexes(zernike_degree + 1)
    idx = _indices(labels)
    centres, radii = minimum_enclosing_circle(labels, idx)
    mag = {n: np.full(len(idx), np.nan) for n in _names("RadialDistribution_ZernikeMagnitude", zi)}
    pha = {n: np.full(len(idx), np.nan) for n in _names("RadialDistribution_ZernikePhase", zi)}
    for k, lab in enumerate(idx):
        ii, jj = np.nonzero(labels == lab)
        if len(ii) == 0:
            continue
        with np.errstate(invalid="ignore", divide="ignore"):
            y = (ii - centres[k, 0]) / radii[k]
            x = (jj - centres[k, 1]) / radii[k]
            zf = construct_zernike_polynomials(x, y, zi)
            w = img[ii, jj]
            for col, (n, m) in enumerate(zi):
                vr = (w * zf[:, col].real).sum()
                vi = (w * zf[:, col].imag).sum()
                mag[f"RadialDistribution_ZernikeMagnitude_{n}_{m}"][k] = np.sqrt(vr * vr + vi * vi) / len(ii)
                pha[f"RadialDistribution_ZernikePhase_{n}_{m}"][k] = math.atan2(vr, vi)
    return {**mag, **pha}
