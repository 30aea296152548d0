"""
TEST INFRASTRUCTURE — restatement of the reference's in-repo per-cell metrics
(src/extraction/core/functions/cell.py:18-303) and trap metrics (trap.py:6-43).

PINNED: tests/golden/cell_metrics.json holds the outputs of the reference module itself
(imported from /root/reference/src in the build container by tests/golden/make_golden.py);
tests/test_oracle_golden.py checks this file against them.
"""

from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi


def area(cell_mask):  # cell.py:18-27
    return np.sum(cell_mask)


def _edt_cone(cell_mask):
    padded = np.pad(cell_mask, 1, mode="constant", constant_values=0)
    nn = ndi.distance_transform_edt(padded == 1) * padded
    return padded, nn


def min_maj_approximation(cell_mask):  # cell.py:207-229
    padded, nn = _edt_cone(cell_mask)
    dn = ndi.distance_transform_edt(nn - nn.max()) * padded
    cone_top = ndi.distance_transform_edt(dn == 0) * padded
    min_ax = np.round(np.max(nn))
    maj_ax = np.round(np.max(dn) + np.sum(cone_top) / 2)
    return min_ax, maj_ax


def eccentricity(cell_mask):  # cell.py:30-40
    min_ax, maj_ax = min_maj_approximation(cell_mask)
    return np.sqrt(maj_ax**2 - min_ax**2) / maj_ax


def volume(cell_mask):  # cell.py:159-172
    min_ax, maj_ax = min_maj_approximation(cell_mask)
    return (4 * np.pi * min_ax**2 * maj_ax) / 3


def conical_volume(cell_mask):  # cell.py:175-186
    _, nn = _edt_cone(cell_mask)
    return 4 * np.sum(nn)


def spherical_volume(cell_mask):  # cell.py:189-204
    r = math.sqrt(area(cell_mask) / np.pi)
    return (4 * np.pi * r**3) / 3


def centroid(cell_mask):  # cell.py:282-293 (1-based coordinate weights)
    wc = np.arange(1, cell_mask.shape[1] + 1).reshape(1, -1)
    wv = np.arange(1, cell_mask.shape[0] + 1).reshape(-1, 1)
    m00 = np.sum(cell_mask)
    return np.sum(cell_mask * wc) / m00, np.sum(cell_mask * wv) / m00


def centroid_x(cell_mask):
    return centroid(cell_mask)[0]


def centroid_y(cell_mask):
    return centroid(cell_mask)[1]


def mean(cell_mask, trap_image):  # cell.py:43-52
    return np.mean(trap_image[cell_mask])


def total(cell_mask, trap_image):
    return np.sum(trap_image[cell_mask])


def total_squared(cell_mask, trap_image):
    return np.sum(trap_image[cell_mask] ** 2)


def median(cell_mask, trap_image):
    return np.median(trap_image[cell_mask])


def std(cell_mask, trap_image):
    return np.std(trap_image[cell_mask])


def max2p5pc(cell_mask, trap_image):  # cell.py:102-119
    npixels = np.sum(cell_mask)
    n_top = int(np.ceil(npixels * 0.025))
    pixels = trap_image[cell_mask]
    top = np.partition(pixels, len(pixels) - n_top)[-n_top:]
    return np.mean(top)


def max5px_median(cell_mask, trap_image):  # cell.py:122-145
    pixels = trap_image[cell_mask]
    if len(pixels) > 5:
        top = np.partition(pixels, len(pixels) - 5)[-5:]
        med = np.median(pixels)
        if med == 0:
            return np.nan
        return np.mean(top) / med
    return np.nan


def moment_of_inertia(cell_mask, trap_image):  # cell.py:232-265 (note: zeroes the image outside the cell)
    x = np.array(trap_image, copy=True)
    x[~cell_mask] = 0
    if not np.any(x):
        return np.nan
    col = np.arange(1, x.shape[1] + 1)[None, :]
    row = np.arange(1, x.shape[0] + 1)[:, None]
    m00 = np.sum(x)
    xm = np.sum(x * col) / m00
    ym = np.sum(x * row) / m00
    mu20 = np.sum(x * (col - xm) ** 2)
    mu02 = np.sum(x * (row - ym) ** 2)
    return mu20 / m00**2 + mu02 / m00**2


def ratio(cell_mask, trap_image):  # cell.py:268-279
    if trap_image.ndim == 3 and trap_image.shape[-1] == 2:
        fl_0 = trap_image[..., 0][cell_mask]
        fl_1 = trap_image[..., 1][cell_mask]
        if np.any(fl_1 == 0):
            return np.nan
        return np.median(fl_0 / fl_1)
    return np.nan


def imBackground(cell_masks, trap_image):  # trap.py:6-23
    if not len(cell_masks):
        cell_masks = np.zeros_like(trap_image)
    background = ~cell_masks.sum(axis=2).astype(bool)
    return np.median(trap_image[np.where(background)])


def background_max5(cell_masks, trap_image):  # trap.py:26-43
    if not len(cell_masks):
        cell_masks = np.zeros_like(trap_image)
    background = ~cell_masks.sum(axis=2).astype(bool)
    return np.mean(np.sort(trap_image[np.where(background)])[-5:])


ONE_ARG = {
    "area": area,
    "centroid": centroid,
    "centroid_x": centroid_x,
    "centroid_y": centroid_y,
    "conical_volume": conical_volume,
    "eccentricity": eccentricity,
    "min_maj_approximation": min_maj_approximation,
    "spherical_volume": spherical_volume,
    "volume": volume,
}
TWO_ARG = {
    "max2p5pc": max2p5pc,
    "max5px_median": max5px_median,
    "mean": mean,
    "median": median,
    "moment_of_inertia": moment_of_inertia,
    "ratio": ratio,
    "std": std,
    "total": total,
    "total_squared": total_squared,
}
