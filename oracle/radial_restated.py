"""
TEST INFRASTRUCTURE — restatement of cp_measure's "radial_distribution" feature (CellProfiler
MeasureObjectIntensityDistribution, centre = the object itself, scaled bins), bound at
src/extraction/core/functions/loaders.py:71-73, default feature list pipe_builder.py:49-56.

cp_measure 0.1.17 / centrosome 1.3.3 are not available: PARITY UNPINNED.  Restated:
  d_to_edge   : Euclidean distance of every object pixel to the nearest pixel outside the object
                (centrosome.cpmorphology.distance_to_edge; the image border is NOT background);
  centre      : pixel of maximal d_to_edge (ties: LAST in raster order, i.e. a stable sort —
                scipy.ndimage.maximum_position leaves ties to an unstable argsort);
  d_from_centre: centrosome.propagate.propagate(zeros, centre, mask, weight=1): Dijkstra over the
                8-neighbourhood inside the object with step cost sqrt((|di|+|dj|) l^2 / (1+l^2)),
                l = 1: sqrt(1/2) for edge steps, 1 for diagonal steps;
  normalised  : d_from_centre / (d_from_centre + d_to_edge + 0.001); bin = int(normalised*bin_count);
  FracAtD     : intensity in bin / intensity in object;
  MeanFrac    : FracAtD / (pixel fraction in bin + eps);
  RadialCV    : coefficient of variation of the mean intensities of the 8 wedges
                (i>ic) + 2 (j>jc) + 4 (|i-ic|>|j-jc|) that have pixels in the bin.
"""

from __future__ import annotations

import heapq

import numpy as np
from scipy import ndimage as ndi

from oracle.cp_measure_restated import _indices

STEP_EDGE = np.sqrt(1.0 / 2.0)
STEP_DIAG = 1.0


def distance_to_edge_object(mask_full):
    """EDT of one object's full-frame mask (no padding: the image border is not background)."""
    return ndi.distance_transform_edt(mask_full)


def propagate_from(mask, ci, cj):
    """Geodesic distance inside `mask` from (ci,cj); inf where unreachable."""
    h, w = mask.shape
    dist = np.full((h, w), np.inf)
    dist[ci, cj] = 0.0
    heap = [(0.0, ci, cj)]
    while heap:
        d, i, j = heapq.heappop(heap)
        if d > dist[i, j]:
            continue
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                if di == 0 and dj == 0:
                    continue
                a, b = i + di, j + dj
                if 0 <= a < h and 0 <= b < w and mask[a, b]:
                    nd = d + (STEP_DIAG if (di != 0 and dj != 0) else STEP_EDGE)
                    if nd < dist[a, b]:
                        dist[a, b] = nd
                        heapq.heappush(heap, (nd, a, b))
    return dist


def names(bin_count=4, scaled=True):
    out = []
    for stat in ("FracAtD", "MeanFrac", "RadialCV"):
        out += [f"RadialDistribution_{stat}_{b}of{bin_count}" for b in range(1, bin_count + 1)]
        if not scaled:  # CellProfiler's overflow ring of the unscaled mode: everything beyond maximum_radius
            out.append(f"RadialDistribution_{stat}_Overflow")
    return out


def get_radial_distribution(masks, pixels, scaled=True, bin_count=4, maximum_radius=100):
    labels = np.asarray(masks)
    img = np.asarray(pixels).astype(np.float64)
    idx = _indices(labels)
    res = {n: np.full(len(idx), np.nan) for n in names(bin_count, scaled)}
    nb = bin_count + 1
    for k, lab in enumerate(idx):
        full = labels == lab
        if not full.any():
            continue
        d_edge_full = distance_to_edge_object(full)
        sl = ndi.find_objects(full.astype(np.int32))[0]
        m = full[sl]
        d_edge = d_edge_full[sl]
        px = img[sl]
        # centre: maximum of d_to_edge, last raster occurrence among ties
        flat = np.where(m.ravel(), d_edge.ravel(), -1.0)
        order = np.argsort(flat, kind="stable")
        ci, cj = np.unravel_index(order[-1], m.shape)
        d_from = propagate_from(m, ci, cj)
        good = m & np.isfinite(d_from)
        norm = np.zeros(m.shape)
        if scaled:
            norm[good] = d_from[good] / (d_from[good] + d_edge[good] + 0.001)
        else:
            norm[good] = d_from[good] / maximum_radius
        bins = (norm * bin_count).astype(int)
        bins[bins > bin_count] = bin_count
        hist = np.zeros(nb)
        cnt = np.zeros(nb)
        np.add.at(hist, bins[good], px[good])
        np.add.at(cnt, bins[good], 1.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            frac_at_d = hist / hist.sum()
            frac_at_bin = cnt / cnt.sum()
            mean_frac = frac_at_d / (frac_at_bin + np.finfo(float).eps)
        ii, jj = np.mgrid[0 : m.shape[0], 0 : m.shape[1]]
        wedge = (ii > ci).astype(int) + 2 * (jj > cj).astype(int) + 4 * (np.abs(ii - ci) > np.abs(jj - cj)).astype(int)
        n_out = bin_count if scaled else bin_count + 1
        for b in range(n_out):
            sel = good & (bins == b)
            vals = np.zeros(8)
            cts = np.zeros(8)
            np.add.at(vals, wedge[sel], px[sel])
            np.add.at(cts, wedge[sel], 1.0)
            have = cts > 0
            if have.any():
                means = vals[have] / cts[have]
                with np.errstate(invalid="ignore", divide="ignore"):
                    cv = np.std(means) / np.mean(means)
            else:
                cv = 0.0
            tag = f"{b + 1}of{bin_count}" if b < bin_count else "Overflow"
            res[f"RadialDistribution_FracAtD_{tag}"][k] = frac_at_d[b]
            res[f"RadialDistribution_MeanFrac_{tag}"][k] = mean_frac[b]
            res[f"RadialDistribution_RadialCV_{tag}"][k] = cv
    return res
