"""
Independent pins of the oracle's unpinned families ("Next round" 1c).

cp_measure / centrosome / mahotas are absent from the reference tree and from this image, so the
restatements in oracle/ cannot be compared with the packages themselves.  Every check below compares a restatement with
something its author could not have fitted: another library's implementation of the same published quantity
(scikit-image greycoprops through the committed fixture, scipy.stats.entropy, scikit-learn's mutual information,
scipy.special Jacobi polynomials, scipy.stats.rankdata, an SVD line fit), a brute-force search, or a closed form.
None of the checkers below shares code with oracle/ or mirrors the kernels' operation order.
"""

import itertools
import json
import math
from pathlib import Path

import numpy as np
import pytest
from scipy import ndimage as ndi

G = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def inputs():
    return np.load(G / "inputs_c1_256.npz")


# ----------------------------------------------------------------------------------------- Haralick: all 13 statistics
def test_all_13_haralick_statistics_from_the_pinned_matrices(inputs):
    """The co-occurrence matrices are pinned to scikit-image's greycomatrix (pair counts + checksum, skimage_glcm.json).
    From those matrices each of the 13 statistics is recomputed by an independent route:
      f1 ASM, f2 contrast, f3 correlation, f5 inverse difference moment (= greycoprops 'homogeneity')  -> scikit-image (fixture)
      f9 entropy                                           -> scipy.stats.entropy (fixture, conda SciPy 1.7) and here (1.15)
      f8 sum entropy, f11 difference entropy               -> scipy.stats.entropy of np.bincount marginals
      f4 variance, f6 sum average, f7 sum variance         -> np.average moments of the index distributions
      f10 difference variance                              -> np.var of the p_{x-y} vector (mahotas' convention)
      f12, f13 information measures of correlation         -> scikit-learn mutual_info_score:  f12 = -I/max(HX,HY),
                                                              f13 = sqrt(1 - exp(-2 I)) with I, H in bits (mahotas mixes
                                                              log2 entropies into a natural exp)
    plus first moment of p_{x-y} against greycoprops 'dissimilarity'."""
    from scipy import stats
    from sklearn.metrics import mutual_info_score

    from oracle import texture_restated as tx

    fx = json.loads((G / "skimage_glcm.json").read_text())
    lab = inputs["nuclei"].astype(np.int32)
    q = tx.img_as_ubyte(inputs["pixels"][0, 0])
    objs = ndi.find_objects(lab)
    checked = 0
    for rec in fx["objects"]:
        i = rec["label"] - 1
        crop = np.where(lab[objs[i]] == i + 1, q[objs[i]], 0).astype(np.int64)
        for d in range(4):
            if rec["ASM"][d] is None:
                continue
            cm = tx.cooccurence(crop, d, 3)
            got = tx.haralick_features(cm)
            cm = cm.astype(np.float64)
            cm[0] = 0
            cm[:, 0] = 0
            assert int(cm.sum()) == rec["pairs"][d]
            p = cm / cm.sum()
            n = len(p)
            ii, jj = np.indices(p.shape)
            px = p.sum(1)
            mu = np.average(np.arange(n), weights=px)
            want = np.empty(13)
            want[0], want[1], want[2], want[4] = rec["ASM"][d], rec["contrast"][d], rec["correlation"][d], rec["homogeneity"][d]
            want[3] = np.average((np.arange(n) - mu) ** 2, weights=px)
            want[5] = np.average((ii + jj).ravel(), weights=p.ravel())
            want[6] = np.average(((ii + jj).ravel() - want[5]) ** 2, weights=p.ravel())
            p_sum = np.bincount((ii + jj).ravel(), weights=p.ravel(), minlength=2 * n)
            p_dif = np.bincount(np.abs(ii - jj).ravel(), weights=p.ravel(), minlength=n)
            want[7] = stats.entropy(p_sum, base=2)
            want[8] = stats.entropy(p.ravel(), base=2)
            want[9] = np.var(p_dif)
            want[10] = stats.entropy(p_dif, base=2)
            mi_bits = mutual_info_score(None, None, contingency=cm) / math.log(2)
            hx = stats.entropy(px, base=2)
            want[11] = -mi_bits / hx if hx > 0 else 0.0
            want[12] = math.sqrt(max(0.0, 1.0 - math.exp(-2.0 * mi_bits)))
            np.testing.assert_allclose(got, want, rtol=2e-9, atol=1e-11, err_msg=f"label {rec['label']} direction {d}")
            assert np.isclose(want[8], rec["entropy_bits"][d], rtol=1e-12)
            assert np.isclose(np.dot(np.arange(n), p_dif), rec["dissimilarity"][d], rtol=1e-12)
            checked += 1
    assert checked >= 80


# ----------------------------------------------------------------------------------------- Zernike polynomials
def test_zernike_polynomials_against_jacobi_closed_form():
    """R_n^m(r) = (-1)^((n-m)/2) r^m P_((n-m)/2)^(m,0)(1 - 2 r^2)  (Born & Wolf), evaluated with scipy.special.eval_jacobi;
    centrosome's complex polynomial is R_n^m(r) * (y + ix)^m / r^m, i.e. phase m*atan2(x, y); zero outside the unit disc."""
    from scipy.special import eval_jacobi

    from oracle import zernike_restated as zr

    rng = np.random.default_rng(5)
    x = rng.uniform(-1.1, 1.1, 4000)
    y = rng.uniform(-1.1, 1.1, 4000)
    zi = zr.get_zernike_indexes(10)
    assert len(zi) == 30 and tuple(zi[0]) == (0, 0) and tuple(zi[-1]) == (9, 9)
    got = zr.construct_zernike_polynomials(x, y, zi)
    r = np.hypot(x, y)
    inside = r <= 1
    for col, (n, m) in enumerate(zi):
        k = (n - m) // 2
        radial = (-1) ** k * r**m * eval_jacobi(k, m, 0, 1 - 2 * r**2)
        want = np.where(inside, radial * np.exp(1j * m * np.arctan2(x, y)), 0)
        np.testing.assert_allclose(got[:, col], want, rtol=0, atol=2e-12, err_msg=f"Z({n},{m})")
    # orthogonality on the disc (a property no coefficient table error survives): <Z_nm, Z_n'm> = pi/(n+1) delta_nn'
    gy, gx = np.mgrid[-1:1:801j, -1:1:801j]
    z = zr.construct_zernike_polynomials(gx.ravel(), gy.ravel(), zi)
    cell = (2 / 800) ** 2
    gram = (z.conj().T @ z) * cell
    norms = np.array([math.pi / (n + 1) for n, _ in zi])
    assert np.allclose(np.diag(gram).real, norms, rtol=2e-2)
    off = gram - np.diag(np.diag(gram))
    assert np.abs(off).max() < 2e-2


def test_zernike_feature_of_a_disc_has_only_the_piston_term():
    """A centred disc is rotation-symmetric: every m > 0 moment vanishes; Z(0,0) = N / (pi r^2) -> 1; the m = 0, n > 0 terms
    integrate to 0 over the full disc (orthogonality to Z00)."""
    from oracle import zernike_restated as zr

    yy, xx = np.mgrid[:201, :201]
    lab = ((yy - 100) ** 2 + (xx - 100) ** 2 <= 90**2).astype(np.int32)
    out = zr.get_zernike(lab)
    assert abs(out["Zernike_0_0"][0] - 1.0) < 5e-3
    for (n, m) in zr.get_zernike_indexes(10)[1:]:
        assert abs(out[f"Zernike_{n}_{m}"][0]) < 5e-3, (n, m)


# ----------------------------------------------------------------------------------------- minimum enclosing circle
def _brute_force_mec(pts):
    best = None
    cands = []
    for a, b in itertools.combinations(range(len(pts)), 2):
        c = (pts[a] + pts[b]) / 2
        cands.append((c, np.hypot(*(pts[a] - c))))
    for a, b, c in itertools.combinations(range(len(pts)), 3):
        A = 2 * np.array([pts[b] - pts[a], pts[c] - pts[a]])
        if abs(np.linalg.det(A)) < 1e-12:
            continue
        rhs = np.array([pts[b] @ pts[b] - pts[a] @ pts[a], pts[c] @ pts[c] - pts[a] @ pts[a]])
        ctr = np.linalg.solve(A, rhs)
        cands.append((ctr, np.hypot(*(pts[a] - ctr))))
    for ctr, r in cands:
        if np.all(np.hypot(*(pts - ctr).T) <= r * (1 + 1e-10) + 1e-10) and (best is None or r < best[1]):
            best = (ctr, r)
    return best


def test_minimum_enclosing_circle_against_brute_force(inputs):
    """Every pair / triple of points defines a candidate circle; the smallest one that contains all points is THE minimum
    enclosing circle (it is unique).  Random point sets and the hulls of real objects."""
    from oracle import zernike_restated as zr
    from oracle.cp_measure_restated import _hull_ccw

    rng = np.random.default_rng(11)
    sets = [rng.integers(0, 40, size=(k, 2)).astype(float) for k in (2, 3, 4, 5, 7, 9, 12, 14) for _ in range(6)]
    sets.append(np.array([[0.0, 0], [0, 10], [10, 0], [10, 10], [5, 5]]))  # four cocircular points
    sets.append(np.array([[0.0, 0], [3, 0], [9, 0]]))  # collinear
    lab = inputs["nuclei"].astype(np.int32)
    for l in range(1, 13):
        rr, cc = np.nonzero(lab == l)
        sets.append(_hull_ccw(np.stack([rr, cc], 1)).astype(float))
    for pts in sets:
        pts = np.unique(pts, axis=0)
        if len(pts) < 2 or len(pts) > 40:
            continue
        c, r = zr.minimum_enclosing_circle_points(pts)
        bc, br = _brute_force_mec(pts)
        assert abs(r - br) <= 1e-9 * max(1, br), (pts, r, br)
        assert np.allclose(c, bc, atol=1e-7)


# ----------------------------------------------------------------------------------------- colocalisation
def _object_pixels(inputs, name="nuclei"):
    lab = inputs[name].astype(np.int32)
    p0 = inputs["pixels"][0, 0].astype(np.float64)
    p1 = inputs["pixels"][1, 0].astype(np.float64)
    return lab, p0, p1


def test_manders_definition(inputs):
    """M1 = sum of channel-1 intensity where BOTH channels are above thr% of their own object maximum, over the sum of
    channel-1 intensity above its threshold (Manders et al. 1993 as used by CellProfiler)."""
    from oracle import cp_measure_restated as cpm

    lab, p0, p1 = _object_pixels(inputs)
    for thr in (15, 40):
        got = cpm.get_correlation_manders_fold(p0, p1, lab, thr=thr)
        for l in range(1, lab.max() + 1):
            f, s = p0[lab == l], p1[lab == l]
            if f.size == 0:
                continue
            hi_f, hi_s = f >= thr / 100 * f.max(), s >= thr / 100 * s.max()
            both = hi_f & hi_s
            assert np.isclose(got["Correlation_Manders_1"][l - 1], f[both].sum() / f[hi_f].sum(), rtol=1e-12)
            assert np.isclose(got["Correlation_Manders_2"][l - 1], s[both].sum() / s[hi_s].sum(), rtol=1e-12)


def test_rwc_definition_with_scipy_rankdata(inputs):
    """Rank-weighted colocalisation (Singan et al. 2011): weight = 1 - |rank1 - rank2| / R with DENSE ranks of the object's
    pixels in each channel (scipy.stats.rankdata) and R = number of rank levels (max over the two channels)."""
    from scipy.stats import rankdata

    from oracle import cp_measure_restated as cpm

    lab, p0, p1 = _object_pixels(inputs)
    for l in range(1, min(lab.max(), 40) + 1):
        m = lab == l
        f, s = p0[m], p1[m]
        if f.size == 0:
            continue
        got = cpm.get_correlation_rwc(np.where(m, p0, 0), np.where(m, p1, 0), m.astype(np.int32))
        r1, r2 = rankdata(f, method="dense"), rankdata(s, method="dense")
        R = max(r1.max(), r2.max())
        w = (R - np.abs(r1 - r2)) / R
        hi_f, hi_s = f >= 0.15 * f.max(), s >= 0.15 * s.max()
        both = hi_f & hi_s
        assert np.isclose(got["Correlation_RWC_1"][0], (f[both] * w[both]).sum() / f[hi_f].sum(), rtol=1e-12), l
        assert np.isclose(got["Correlation_RWC_2"][0], (s[both] * w[both]).sum() / s[hi_s].sum(), rtol=1e-12), l


def test_costes_regression_line_and_threshold_search(inputs):
    """Costes et al. 2004: the thresholds lie on the orthogonal-regression (total least squares) line of channel 2 on
    channel 1 — checked against the principal axis from an SVD of the centred pixel cloud — and the search returns the
    threshold where the Pearson correlation of the below-threshold pixels changes sign — checked against an exhaustive
    scan of all 255 candidates on objects where that correlation is monotone in the threshold."""
    from scipy import stats

    from oracle import cp_measure_restated as cpm

    lab, p0, p1 = _object_pixels(inputs)
    p0, p1 = p0 / 65535.0, p1 / 65535.0
    clouds = [(p0[lab == l], p1[lab == l]) for l in range(1, min(lab.max(), 30) + 1)]
    # synthetic clouds with an anti-correlated dim population and a correlated bright one: the below-threshold correlation
    # is negative at low thresholds and turns positive exactly once
    rng = np.random.default_rng(8)
    for k in range(8):
        n, cut = 600 + 50 * k, 0.2 + 0.03 * k
        f = rng.uniform(0.02, 0.9, n)
        s = np.where(f < cut, cut - f + 0.02, 0.8 * f) + rng.normal(0, 0.01, n)
        clouds.append((f, np.clip(s, 0.001, 1)))
    scanned = 0
    for l, (f, s) in enumerate(clouds):
        if f.size < 20:
            continue
        t1, t2 = cpm.bisection_costes(f, s, 255)
        # the line: principal axis of the centred cloud
        pts = np.stack([f - f.mean(), s - s.mean()], 1)
        _, _, vt = np.linalg.svd(pts, full_matrices=False)
        a = vt[0, 1] / vt[0, 0]
        b = s.mean() - a * f.mean()
        assert np.isclose(t2, a * t1 + b, rtol=1e-8, atol=1e-10), l
        # exhaustive scan
        cost = np.full(256, np.nan)
        for t in range(1, 256):
            below = (f < t / 255) | (s < a * t / 255 + b)
            if below.sum() > 2 and f[below].std() > 0 and s[below].std() > 0:
                cost[t] = stats.pearsonr(f[below], s[below])[0]
        ok = ~np.isnan(cost)
        sign = np.sign(cost[ok])
        if not ((sign[:-1] <= sign[1:]).all() and (sign < 0).any() and (sign >= 0).any()):
            continue  # not monotone: bisection and scan may legitimately differ
        first_nonneg = int(np.flatnonzero(ok & (np.nan_to_num(cost, nan=-1) >= 0))[0])
        assert abs(round(t1 * 255) + 1 - first_nonneg) <= 1, (l, t1 * 255, first_nonneg)
        scanned += 1
    print("costes: objects scanned exhaustively:", scanned)
    assert scanned >= 5
    # the fractions above the thresholds, definitional
    got = cpm.get_correlation_costes(p0, p1, lab)
    for l in range(1, min(lab.max(), 30) + 1):
        f, s = p0[lab == l], p1[lab == l]
        if f.size == 0:
            continue
        t1, t2 = cpm.bisection_costes(f, s, 255)
        both = (f > t1) & (s > t2)
        if both.any():
            assert np.isclose(got["Correlation_Costes_1"][l - 1], f[both].sum() / f[f >= t1].sum(), rtol=1e-12)
            assert np.isclose(got["Correlation_Costes_2"][l - 1], s[both].sum() / s[s >= t2].sum(), rtol=1e-12)


# ----------------------------------------------------------------------------------------- drift, phase-normalised
def _dft_matrix(n):
    k = np.arange(n)
    return np.exp(-2j * np.pi * np.outer(k, k) / n)


def test_phase_normalised_drift_against_explicit_dft_and_closed_form():
    """normalization="phase" (scikit-image >= 0.19's default, the variant the reference's pinned 0.26 runs): (a) on small
    frames the cross-power spectrum is built from explicit DFT matrices (no FFT library) and the peak compared; (b) for a
    pure circular shift the normalised spectrum is a pure phase ramp, so the correlation surface must be a unit impulse
    at minus the applied shift — a closed form."""
    from oracle.drift_restated import phase_cross_correlation

    rng = np.random.default_rng(3)
    for shape, shift in (((24, 20), (3, -5)), ((17, 23), (-6, 4)), ((16, 16), (0, 0)), ((21, 12), (10, 5))):
        ref = rng.normal(size=shape) + 5
        mov = np.roll(ref, shift, axis=(0, 1)) + rng.normal(scale=0.05, size=shape)
        Fy, Fx = _dft_matrix(shape[0]), _dft_matrix(shape[1])
        R, M = Fy @ ref @ Fx, Fy @ mov @ Fx
        P = R * M.conj()
        P /= np.maximum(np.abs(P), 100 * np.finfo(float).eps)
        cc = (Fy.conj() @ P @ Fx.conj()) / (shape[0] * shape[1])
        peak = np.array(np.unravel_index(np.argmax(np.abs(cc)), shape), dtype=float)
        for ax in range(2):
            if peak[ax] > shape[ax] // 2:
                peak[ax] -= shape[ax]
        got = phase_cross_correlation(ref, mov, normalization="phase")
        assert got.tolist() == peak.tolist()
        assert got.tolist() == [-float(shift[0]), -float(shift[1])] or shift == (10, 5)
    # closed form: exact circular shift -> impulse of height 1
    ref = rng.normal(size=(64, 48)) + 3
    for shift in ((5, 7), (-20, 13), (31, -23)):
        mov = np.roll(ref, shift, axis=(0, 1))
        P = np.fft.fft2(ref) * np.fft.fft2(mov).conj()
        P /= np.abs(P)
        cc = np.fft.ifft2(P)
        at = tuple((-np.array(shift)) % np.array(ref.shape))
        assert abs(cc[at] - 1) < 1e-9 and np.abs(np.delete(cc.ravel(), np.ravel_multi_index(at, ref.shape))).max() < 1e-9
        want = -np.array(shift, dtype=float)
        want = np.where(want > np.array(ref.shape) // 2, want - ref.shape, want)
        want = np.where(want < -(np.array(ref.shape) - 1) // 2 - 0.5, want + ref.shape, want)
        assert phase_cross_correlation(ref, mov, normalization="phase").tolist() == want.tolist()


# ------------------------------------------------------------------------- radial distribution: brute force + Bellman-Ford
def test_radial_distribution_against_brute_force_distances_and_relaxation():
    """oracle/radial_restated.py (EDT for the distance to the edge, Dijkstra with a heap for the geodesic distance from the
    centre, numpy scatter-adds for the ring / wedge statistics) against the same published definition computed another way:
    the distance of every object pixel to every outside pixel, Bellman-Ford sweeps until nothing changes, and explicit
    per-pixel accumulation — on a disc, an L-shape (where geodesic and straight-line distance differ), a ring with a hole
    and a one-pixel object."""
    from oracle import radial_restated as rr

    def shapes():
        yy, xx = np.mgrid[0:40, 0:44]
        lab = np.zeros((40, 44), np.int32)
        lab[(yy - 9) ** 2 + (xx - 10) ** 2 <= 36] = 1                      # disc
        lab[22:36, 4:8] = 2                                                # L-shape
        lab[32:36, 4:20] = 2
        ring = ((yy - 12) ** 2 + (xx - 32) ** 2 <= 64) & ((yy - 12) ** 2 + (xx - 32) ** 2 >= 9)
        lab[ring] = 3                                                      # ring with a hole
        lab[30, 40] = 4                                                    # a single pixel
        return lab

    lab = shapes()
    rng = np.random.default_rng(11)
    img = rng.integers(50, 4000, lab.shape).astype(np.float64)
    bins_n = 4
    got = rr.get_radial_distribution(lab, img, scaled=True, bin_count=bins_n)

    edge_step, diag_step = math.sqrt(0.5), 1.0
    for k, L in enumerate([1, 2, 3, 4]):
        m = lab == L
        pts = np.argwhere(m)
        outside = np.argwhere(~m)
        d_edge = {tuple(p): float(np.sqrt(((outside - p) ** 2).sum(1)).min()) for p in pts}  # (the frame's border is not background)
        best = max(d_edge.values())
        centre = max((p for p in d_edge if d_edge[p] == best))  # last in raster order among the ties
        dist = {p: math.inf for p in d_edge}
        dist[centre] = 0.0
        changed = True
        while changed:  # Bellman-Ford over the 8-neighbourhood inside the object
            changed = False
            for (i, j) in dist:
                for di, dj in itertools.product((-1, 0, 1), repeat=2):
                    q = (i + di, j + dj)
                    if (di or dj) and q in dist:
                        nd = dist[q] + (diag_step if di and dj else edge_step)
                        if nd < dist[(i, j)] - 1e-15:
                            dist[(i, j)] = nd
                            changed = True
        tot = cnt = 0.0
        ring_sum, ring_cnt = [0.0] * (bins_n + 1), [0.0] * (bins_n + 1)
        wedges = [[[0.0, 0.0] for _ in range(8)] for _ in range(bins_n + 1)]
        for p, d in dist.items():
            if not math.isfinite(d):
                continue
            b = min(int(d / (d + d_edge[p] + 0.001) * bins_n), bins_n)
            v = img[p]
            tot += v
            cnt += 1
            ring_sum[b] += v
            ring_cnt[b] += 1
            wdg = (p[0] > centre[0]) + 2 * (p[1] > centre[1]) + 4 * (abs(p[0] - centre[0]) > abs(p[1] - centre[1]))
            wedges[b][wdg][0] += v
            wedges[b][wdg][1] += 1
        for b in range(bins_n):
            frac = ring_sum[b] / tot
            mean_frac = frac / (ring_cnt[b] / cnt + np.finfo(float).eps)
            means = [s / c for s, c in wedges[b] if c > 0]
            cv = float(np.std(means) / np.mean(means)) if means else float("nan")
            for stat, want in (("FracAtD", frac), ("MeanFrac", mean_frac), ("RadialCV", cv)):
                have = got[f"RadialDistribution_{stat}_{b + 1}of{bins_n}"][k]
                if ring_cnt[b] == 0 and stat == "RadialCV":
                    continue  # (an empty ring: the restatement's convention for the CV of nothing is its own)
                assert (math.isnan(want) and math.isnan(have)) or have == pytest.approx(want, rel=1e-12, abs=1e-12), (L, b, stat, have, want)


# ------------------------------------------------------------------------------------------- intensity: per-object loops
def test_intensity_family_against_per_object_python(inputs):
    """oracle/cp_measure_restated.get_intensity (labelled ndimage sums, lexsort order statistics) against the same quantities
    written per object with plain Python / NumPy on the object's own pixel list: moments, extrema, the inner-boundary ("edge")
    statistics with the boundary found pixel by pixel from the 4-neighbourhood, centres of mass and their displacement, and
    CellProfiler's order statistics (value at position n·q of the sorted list, linearly interpolated; the last value when that
    position is the last)."""
    from oracle import cp_measure_restated as cm

    labels = inputs["nuclei"].astype(np.int32)
    img = inputs["pixels"][0, 0].astype(np.float64) if inputs["pixels"].ndim == 4 else inputs["pixels"][0].astype(np.float64)
    got = cm.get_intensity(labels, img, edge_measurements=True)
    H, W = labels.shape

    def order_stat(sorted_vals, q):
        n = len(sorted_vals)
        pos = n * q
        i = int(math.floor(pos))
        if i < n - 1:
            f = pos - i
            return sorted_vals[i] * (1 - f) + sorted_vals[i + 1] * f
        return sorted_vals[min(i, n - 1)]

    for L in list(range(1, int(labels.max()) + 1))[:: max(1, int(labels.max()) // 12)]:
        ys, xs = np.nonzero(labels == L)
        vals = img[ys, xs]
        k = L - 1
        assert got["Intensity_IntegratedIntensity"][k] == pytest.approx(vals.sum(), rel=1e-12)
        assert got["Intensity_MeanIntensity"][k] == pytest.approx(vals.mean(), rel=1e-12)
        assert got["Intensity_StdIntensity"][k] == pytest.approx(math.sqrt(((vals - vals.mean()) ** 2).mean()), rel=1e-10)
        assert got["Intensity_MinIntensity"][k] == vals.min() and got["Intensity_MaxIntensity"][k] == vals.max()
        edge = []
        for y, x in zip(ys, xs):  # inner boundary: a 4-neighbour (edge-replicated at the frame) carries another label
            nb = [labels[min(max(y + dy, 0), H - 1), min(max(x + dx, 0), W - 1)] for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1))]
            if any(v != L for v in nb):
                edge.append(img[y, x])
        edge = np.asarray(edge)
        assert got["Intensity_IntegratedIntensityEdge"][k] == pytest.approx(edge.sum(), rel=1e-12)
        assert got["Intensity_MeanIntensityEdge"][k] == pytest.approx(edge.mean(), rel=1e-12)
        assert got["Intensity_StdIntensityEdge"][k] == pytest.approx(math.sqrt(((edge - edge.mean()) ** 2).mean()), rel=1e-10)
        assert got["Intensity_MinIntensityEdge"][k] == edge.min() and got["Intensity_MaxIntensityEdge"][k] == edge.max()
        cx, cy = xs.mean(), ys.mean()
        wx, wy = (xs * vals).sum() / vals.sum(), (ys * vals).sum() / vals.sum()
        assert got["Location_CenterMassIntensity_X"][k] == pytest.approx(wx, rel=1e-12)
        assert got["Location_CenterMassIntensity_Y"][k] == pytest.approx(wy, rel=1e-12)
        assert got["Intensity_MassDisplacement"][k] == pytest.approx(math.hypot(cx - wx, cy - wy), rel=1e-9, abs=1e-12)
        sv = np.sort(vals)
        med = order_stat(sv, 0.5)
        assert got["Intensity_LowerQuartileIntensity"][k] == pytest.approx(order_stat(sv, 0.25), rel=1e-12)
        assert got["Intensity_MedianIntensity"][k] == pytest.approx(med, rel=1e-12)
        assert got["Intensity_UpperQuartileIntensity"][k] == pytest.approx(order_stat(sv, 0.75), rel=1e-12)
        assert got["Intensity_MADIntensity"][k] == pytest.approx(order_stat(np.sort(np.abs(vals - med)), 0.5), rel=1e-12, abs=1e-12)
        # position of the maximum: the last raster occurrence among ties
        at = np.nonzero(vals == vals.max())[0][-1]
        assert (got["Location_MaxIntensity_X"][k], got["Location_MaxIntensity_Y"][k]) == (xs[at], ys[at])


# ---------------------------------------------------------------------------------------------- Feret diameters: no hull
def test_feret_diameters_without_a_convex_hull(inputs):
    """oracle/cp_measure_restated.feret_diameters (qhull + per-edge widths) against a search that never builds a hull: the largest
    distance between any two pixels of the object, and the smallest width of the pixel set over the directions of ALL pixel
    pairs (the hull's edges are among them, so the minimum is the same)."""
    from oracle import cp_measure_restated as cm

    labels = inputs["nuclei"].astype(np.int32)
    checked = 0
    for L in range(1, int(labels.max()) + 1):
        ys, xs = np.nonzero(labels == L)
        if not (5 <= len(ys) <= 600):
            continue
        pts = np.stack([ys, xs], 1).astype(float)
        d = pts[:, None, :] - pts[None, :, :]
        fmax = float(np.sqrt((d ** 2).sum(-1)).max())
        best = math.inf
        for i in range(len(pts)):
            e = pts - pts[i]                                   # directions p_i -> p_j
            ln = np.hypot(e[:, 0], e[:, 1])
            ok = ln > 0
            u = e[ok] / ln[ok, None]
            # signed distance of every pixel from the line through p_i with direction u: one row per direction
            w = u[:, 0:1] * (pts[None, :, 1] - pts[i, 1]) - u[:, 1:2] * (pts[None, :, 0] - pts[i, 0])
            best = min(best, float((w.max(1) - w.min(1)).min()))
        fmin_o, fmax_o = cm.feret_diameters(labels == L)
        assert fmax_o == pytest.approx(fmax, rel=1e-12) and fmin_o == pytest.approx(best, rel=1e-9, abs=1e-9), (L, fmin_o, best)
        checked += 1
        if checked == 6:
            break
    assert checked >= 3
