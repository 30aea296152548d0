"""
TEST INFRASTRUCTURE — CPU restatement of the cp_measure feature functions the reference binds
at src/extraction/core/functions/loaders.py:71-77 (`get_core_measurements()` /
`get_correlation_measurements()`), with the call conventions of
`wrap_cp_measure_features` (loaders.py:135-150: `fun(mask.astype(uint16), pixels, **kw)`) and
`wrap_cp_corr_features` (loaders.py:153-167: `fun(pixels1, pixels2, mask, **kw)`).

cp_measure 0.1.17 (uv.lock:441-442) and its dependencies centrosome 1.3.3 / mahotas 1.4.18 /
scikit-image 0.26.0 are NOT in /root/reference nor installed: each function below restates the
published CellProfiler 4 measurement definition that cp_measure ports, written in terms of the
same SciPy primitives.  PARITY UNPINNED for the cp_measure level; the primitives are pinned
against scikit-image 0.18.3 / SciPy in tests/golden (see oracle/__init__.py).

Every function takes a label image `masks` (the reference passes a full-frame binary mask of ONE
object, so labels are {0,1}) and returns {feature_name: float64 array of length n_objects}.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi
from scipy.spatial import ConvexHull

# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def _fix(x):
    """centrosome.cpmorphology.fixup_scipy_ndimage_result: always a 1-d float array."""
    return np.atleast_1d(np.asarray(x, dtype=float))


def _indices(masks):
    n = int(masks.max()) if masks.size else 0
    return np.arange(1, n + 1, dtype=np.int32)


def find_boundaries_inner(labels):
    """skimage.segmentation.find_boundaries(labels, connectivity=1, mode="inner").

    boundaries = (grey_dilation != grey_erosion) & (labels != 0) with the 4-neighbour cross and
    skimage's default border handling (scipy mode="reflect": the edge pixel is replicated)."""
    fp = ndi.generate_binary_structure(labels.ndim, 1)
    lab = labels.astype(np.int32)
    dil = ndi.grey_dilation(lab, footprint=fp, mode="reflect")
    ero = ndi.grey_erosion(lab, footprint=fp, mode="reflect")
    return (dil != ero) & (lab != 0)


# --------------------------------------------------------------------------------------
# intensity  (CellProfiler MeasureObjectIntensity)
# --------------------------------------------------------------------------------------


def get_intensity(masks, pixels, edge_measurements=True):
    labels = np.asarray(masks)
    img = np.asarray(pixels)
    lindexes = _indices(labels)
    n = len(lindexes)
    z = lambda: np.zeros(n)  # noqa: E731
    integrated, mean_i, std_i, min_i, max_i = z(), z(), z(), z(), z()
    e_int, e_mean, e_std, e_min, e_max = z(), z(), z(), z(), z()
    mass_disp, lq, med, mad, uq = z(), z(), z(), z(), z()
    cmi_x, cmi_y, cmi_z = z(), z(), z()
    max_x, max_y, max_z = z(), z(), z()
    lmask = labels > 0
    if n and lmask.any():
        limg = img[lmask]
        llabels = labels[lmask].astype(np.int32)
        mesh_y, mesh_x = np.mgrid[0 : labels.shape[0], 0 : labels.shape[1]]
        mesh_x = mesh_x[lmask]
        mesh_y = mesh_y[lmask]
        lcount = _fix(ndi.sum(np.ones(len(limg)), llabels, lindexes))
        integrated[:] = _fix(ndi.sum(limg, llabels, lindexes))
        with np.errstate(invalid="ignore", divide="ignore"):
            mean_i[:] = integrated / lcount
            std_i[:] = np.sqrt(_fix(ndi.mean((limg - mean_i[llabels - 1]) ** 2, llabels, lindexes)))
        min_i[:] = _fix(ndi.minimum(limg, llabels, lindexes))
        max_i[:] = _fix(ndi.maximum(limg, llabels, lindexes))
        # position of the maximum: scipy's maximum_position sorts with an unstable argsort, so ties are
        # implementation-defined there; this restatement fixes ties to the LAST raveled occurrence
        # (what a stable sort yields).
        order = np.argsort(limg, kind="stable")
        mp = np.zeros(n + 1, dtype=np.int64)
        mp[llabels[order]] = order
        mp = mp[1:]
        present = lcount > 0
        max_x[present] = mesh_x[mp[present]]
        max_y[present] = mesh_y[mp[present]]
        cm_x = _fix(ndi.mean(mesh_x, llabels, lindexes))
        cm_y = _fix(ndi.mean(mesh_y, llabels, lindexes))
        i_x = _fix(ndi.sum(mesh_x * limg.astype(float), llabels, lindexes))
        i_y = _fix(ndi.sum(mesh_y * limg.astype(float), llabels, lindexes))
        with np.errstate(invalid="ignore", divide="ignore"):
            cmi_x[:] = i_x / integrated
            cmi_y[:] = i_y / integrated
            cmi_z[:] = 0.0 * integrated / integrated  # mesh_z == 0 for a 2-D plane
        dx, dy = cm_x - cmi_x, cm_y - cmi_y
        mass_disp[:] = np.sqrt(dx * dx + dy * dy)
        # order statistics: sort by label then intensity, interpolate at area*fraction
        order = np.lexsort((limg, llabels))
        areas = lcount.astype(int)
        indices = np.cumsum(areas) - areas
        for dest, fraction in ((lq, 0.25), (med, 0.5), (uq, 0.75)):
            qindex = indices.astype(float) + areas * fraction
            qfraction = qindex - np.floor(qindex)
            qindex = qindex.astype(int)
            qmask = qindex < indices + areas - 1
            qi, qf = qindex[qmask], qfraction[qmask]
            dest[lindexes[qmask] - 1] = limg[order[qi]] * (1 - qf) + limg[order[qi + 1]] * qf
            qmask = (~qmask) & (areas > 0)
            dest[lindexes[qmask] - 1] = limg[order[qindex[qmask]]]
        madimg = np.abs(limg - med[llabels - 1])
        order = np.lexsort((madimg, llabels))
        qindex = indices.astype(float) + areas / 2.0
        qfraction = qindex - np.floor(qindex)
        qindex = qindex.astype(int)
        qmask = qindex < indices + areas - 1
        qi, qf = qindex[qmask], qfraction[qmask]
        mad[lindexes[qmask] - 1] = madimg[order[qi]] * (1 - qf) + madimg[order[qi + 1]] * qf
        qmask = (~qmask) & (areas > 0)
        mad[lindexes[qmask] - 1] = madimg[order[qindex[qmask]]]
        absent = ~present
        for arr in (integrated, mean_i, std_i, min_i, max_i, mass_disp, lq, med, mad, uq, cmi_x, cmi_y, cmi_z, max_x, max_y, max_z):
            arr[absent] = np.nan
    if edge_measurements and n:
        emask = find_boundaries_inner(labels)
        eimg = img[emask]
        elabels = labels[emask].astype(np.int32)
        if len(eimg):
            ecount = _fix(ndi.sum(np.ones(len(eimg)), elabels, lindexes))
            e_int[:] = _fix(ndi.sum(eimg, elabels, lindexes))
            with np.errstate(invalid="ignore", divide="ignore"):
                e_mean[:] = e_int / ecount
                e_std[:] = np.sqrt(_fix(ndi.mean((eimg - e_mean[elabels - 1]) ** 2, elabels, lindexes)))
            e_min[:] = _fix(ndi.minimum(eimg, elabels, lindexes))
            e_max[:] = _fix(ndi.maximum(eimg, elabels, lindexes))
            none = ecount == 0
            for arr in (e_int, e_mean, e_std, e_min, e_max):
                arr[none] = 0.0
    out = {
        "Intensity_IntegratedIntensity": integrated,
        "Intensity_MeanIntensity": mean_i,
        "Intensity_StdIntensity": std_i,
        "Intensity_MinIntensity": min_i,
        "Intensity_MaxIntensity": max_i,
    }
    if edge_measurements:
        out.update(
            {
                "Intensity_IntegratedIntensityEdge": e_int,
                "Intensity_MeanIntensityEdge": e_mean,
                "Intensity_StdIntensityEdge": e_std,
                "Intensity_MinIntensityEdge": e_min,
                "Intensity_MaxIntensityEdge": e_max,
            }
        )
    out.update(
        {
            "Intensity_MassDisplacement": mass_disp,
            "Intensity_LowerQuartileIntensity": lq,
            "Intensity_MedianIntensity": med,
            "Intensity_MADIntensity": mad,
            "Intensity_UpperQuartileIntensity": uq,
            "Location_CenterMassIntensity_X": cmi_x,
            "Location_CenterMassIntensity_Y": cmi_y,
            "Location_CenterMassIntensity_Z": cmi_z,
            "Location_MaxIntensity_X": max_x,
            "Location_MaxIntensity_Y": max_y,
            "Location_MaxIntensity_Z": max_z,
        }
    )
    return out


# --------------------------------------------------------------------------------------
# regionprops primitives (scikit-image definitions, restated with SciPy)
# --------------------------------------------------------------------------------------

_STREL_4 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)


def perimeter(image):
    """skimage.measure.perimeter(image, neighbourhood=4)."""
    image = image.astype(np.uint8)
    eroded = ndi.binary_erosion(image, _STREL_4, border_value=0)
    border = image - eroded
    w = np.zeros(50)
    w[[5, 7, 15, 17, 25, 27]] = 1
    w[[21, 33]] = np.sqrt(2)
    w[[13, 23]] = (1 + np.sqrt(2)) / 2
    pimg = ndi.convolve(border, np.array([[10, 2, 10], [2, 1, 2], [10, 2, 10]]), mode="constant", cval=0)
    hist = np.bincount(pimg.ravel(), minlength=50)
    return float(hist[:50] @ w)


def euler_number(image):
    """skimage.measure.euler_number(image, connectivity=2) (bit-quad counts; scikit-image >= 0.19)."""
    img = np.pad((image > 0).astype(int), 1, mode="constant")
    config = np.array([[0, 0, 0], [0, 1, 4], [0, 2, 8]])
    coefs = np.array([0, 0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, -1, 0])
    xf = ndi.convolve(img, config, mode="constant", cval=0)
    h = np.bincount(xf.ravel(), minlength=16)
    return int(coefs @ h)


def moments_central(image, center, order=3):
    """skimage.measure.moments_central via successive dot products (float64)."""
    calc = image.astype(float)
    for dim, dim_length in enumerate(image.shape):
        delta = np.arange(dim_length, dtype=float) - center[dim]
        powers = delta[:, np.newaxis] ** np.arange(order + 1)
        calc = np.rollaxis(calc, dim, image.ndim)
        calc = np.dot(calc, powers)
        calc = np.rollaxis(calc, -1, dim)
    return calc


def moments_normalized(mu, order=3):
    nu = np.zeros_like(mu)
    mu0 = mu[0, 0]
    for p in range(order + 1):
        for q in range(order + 1):
            nu[p, q] = np.nan if p + q < 2 else mu[p, q] / mu0 ** ((p + q) / 2 + 1)
    return nu


def moments_hu(nu):
    t0 = nu[3, 0] + nu[1, 2]
    t1 = nu[2, 1] + nu[0, 3]
    q0, q1 = t0 * t0, t1 * t1
    n4 = 4 * nu[1, 1]
    s = nu[2, 0] + nu[0, 2]
    d = nu[2, 0] - nu[0, 2]
    hu = np.zeros(7)
    hu[0] = s
    hu[1] = d * d + n4 * nu[1, 1]
    hu[3] = q0 + q1
    hu[5] = d * (q0 - q1) + n4 * t0 * t1
    t0 *= q0 - 3 * q1
    t1 *= 3 * q0 - q1
    q0 = nu[3, 0] - 3 * nu[1, 2]
    q1 = 3 * nu[2, 1] - nu[0, 3]
    hu[2] = q0 * q0 + q1 * q1
    hu[4] = q0 * t0 + q1 * t1
    hu[6] = q1 * t0 - q0 * t1
    return hu


def _hull_ccw(points):
    """Convex hull (counter-clockwise in (x=col, y=row) sense) of integer points via qhull."""
    pts = np.unique(np.asarray(points), axis=0)
    if len(pts) < 3:
        return pts
    try:
        hull = ConvexHull(pts)
    except Exception:  # collinear
        order = np.lexsort((pts[:, 1], pts[:, 0]))
        return pts[[order[0], order[-1]]]
    return pts[hull.vertices]


def convex_hull_image(image):
    """skimage.morphology.convex_hull_image(image, offset_coordinates=True, include_borders=True).

    Hull of the diamond-offset coordinates (r±0.5,c),(r,c±0.5) of the object's pixels; a pixel belongs to the
    hull image when its centre is inside or on the hull.  Coordinates are doubled so the test is exact."""
    rr, cc = np.nonzero(image)
    if len(rr) == 0:
        return np.zeros(image.shape, bool)
    pts = np.concatenate(
        [
            np.stack([2 * rr - 1, 2 * cc], 1),
            np.stack([2 * rr + 1, 2 * cc], 1),
            np.stack([2 * rr, 2 * cc - 1], 1),
            np.stack([2 * rr, 2 * cc + 1], 1),
        ]
    ).astype(np.int64)
    hv = _hull_ccw(pts)
    # orientation-agnostic inside test: all cross products share a sign (or are zero)
    R, Cc = np.mgrid[0 : image.shape[0], 0 : image.shape[1]]
    R2, C2 = 2 * R.astype(np.int64), 2 * Cc.astype(np.int64)
    pos = np.ones(image.shape, bool)
    neg = np.ones(image.shape, bool)
    k = len(hv)
    for i in range(k):
        a, b = hv[i], hv[(i + 1) % k]
        cross = (b[0] - a[0]) * (C2 - a[1]) - (b[1] - a[1]) * (R2 - a[0])
        pos &= cross >= 0
        neg &= cross <= 0
    return pos | neg


def feret_diameters(image):
    """centrosome.cpmorphology.feret_diameter on the convex hull of the pixel centres:
    max = diameter of the hull; min = minimum width (min over hull edges of the farthest vertex)."""
    rr, cc = np.nonzero(image)
    pts = np.stack([rr, cc], 1).astype(np.int64)
    hv = _hull_ccw(pts).astype(float)
    k = len(hv)
    if k == 1:
        return 0.0, 0.0
    d = hv[:, None, :] - hv[None, :, :]
    dist = np.sqrt((d**2).sum(-1))
    fmax = float(dist.max())
    if k == 2:
        return 0.0, fmax
    fmin = np.inf
    for i in range(k):
        a, b = hv[i], hv[(i + 1) % k]
        e = b - a
        L = np.hypot(*e)
        w = np.abs(e[0] * (hv[:, 1] - a[1]) - e[1] * (hv[:, 0] - a[0])) / L
        fmin = min(fmin, float(w.max()))
    return fmin, fmax


def sizeshape_names():
    names = [
        "Area", "BoundingBoxArea", "BoundingBoxMaximum_X", "BoundingBoxMaximum_Y", "BoundingBoxMinimum_X",
        "BoundingBoxMinimum_Y", "Center_X", "Center_Y", "Compactness", "ConvexArea", "Eccentricity",
        "EquivalentDiameter", "EulerNumber", "Extent", "FormFactor", "MajorAxisLength", "MaxFeretDiameter",
        "MaximumRadius", "MeanRadius", "MedianRadius", "MinFeretDiameter", "MinorAxisLength", "Orientation",
        "Perimeter", "Solidity",
    ]
    names += [f"SpatialMoment_{p}_{q}" for p in range(3) for q in range(4)]
    names += [f"CentralMoment_{p}_{q}" for p in range(3) for q in range(4)]
    names += [f"NormalizedMoment_{p}_{q}" for p in range(4) for q in range(4)]
    names += [f"HuMoment_{k}" for k in range(7)]
    names += [f"InertiaTensor_{i}_{j}" for i in range(2) for j in range(2)]
    names += [f"InertiaTensorEigenvalues_{k}" for k in range(2)]
    return names


def get_sizeshape(masks, pixels=None):
    """CellProfiler MeasureObjectSizeShape (2-D, calculate_advanced=True, zernikes split out)."""
    labels = np.asarray(masks)
    idx = _indices(labels)
    n = len(idx)
    names = sizeshape_names()
    res = {k: np.full(n, np.nan) for k in names}
    slices = ndi.find_objects(labels.astype(np.int32), max_label=n)
    for i, sl in enumerate(slices):
        if sl is None:
            continue
        img = labels[sl] == (i + 1)
        area = float(img.sum())
        h, w = img.shape
        y0, x0 = sl[0].start, sl[1].start
        M = moments_central(img, (0.0, 0.0), 3)
        rbar, cbar = M[1, 0] / M[0, 0], M[0, 1] / M[0, 0]
        mu = moments_central(img, (rbar, cbar), 3)
        nu = moments_normalized(mu, 3)
        hu = moments_hu(nu)
        T = np.array([[mu[0, 2], -mu[1, 1]], [-mu[1, 1], mu[2, 0]]]) / mu[0, 0]
        ev = np.clip(np.linalg.eigvalsh(T), 0, None)
        l1, l2 = float(ev.max()), float(ev.min())
        a, b, c = T[0, 0], T[0, 1], T[1, 1]
        if a - c == 0:
            orient = -np.pi / 4 if b < 0 else np.pi / 4
        else:
            orient = 0.5 * np.arctan2(-2 * b, c - a)
        per = perimeter(img)
        cvx = float(convex_hull_image(img).sum())
        dist = ndi.distance_transform_edt(np.pad(img, 1))
        dvals = dist[np.pad(img, 1)]
        fmin, fmax = feret_diameters(img)
        fpa = 4.0 * np.pi * area
        r = res
        r["Area"][i] = area
        r["BoundingBoxArea"][i] = h * w
        r["BoundingBoxMaximum_X"][i] = x0 + w
        r["BoundingBoxMaximum_Y"][i] = y0 + h
        r["BoundingBoxMinimum_X"][i] = x0
        r["BoundingBoxMinimum_Y"][i] = y0
        r["Center_X"][i] = x0 + cbar
        r["Center_Y"][i] = y0 + rbar
        with np.errstate(divide="ignore", invalid="ignore"):
            r["Compactness"][i] = per**2 / max(fpa, 1.0)
            r["FormFactor"][i] = np.float64(fpa) / np.float64(per**2)
        r["ConvexArea"][i] = cvx
        r["Eccentricity"][i] = 0.0 if l1 == 0 else np.sqrt(1 - l2 / l1)
        r["EquivalentDiameter"][i] = np.sqrt(4 * area / np.pi)
        r["EulerNumber"][i] = euler_number(img)
        r["Extent"][i] = area / (h * w)
        r["MajorAxisLength"][i] = 4 * np.sqrt(l1)
        r["MinorAxisLength"][i] = 4 * np.sqrt(l2)
        r["MaxFeretDiameter"][i] = fmax
        r["MinFeretDiameter"][i] = fmin
        r["MaximumRadius"][i] = dvals.max()
        r["MeanRadius"][i] = dvals.mean()
        r["MedianRadius"][i] = np.median(dvals)
        r["Orientation"][i] = orient * 180.0 / np.pi
        r["Perimeter"][i] = per
        r["Solidity"][i] = area / cvx
        for p in range(3):
            for q in range(4):
                r[f"SpatialMoment_{p}_{q}"][i] = M[p, q]
                r[f"CentralMoment_{p}_{q}"][i] = mu[p, q]
        for p in range(4):
            for q in range(4):
                r[f"NormalizedMoment_{p}_{q}"][i] = nu[p, q]
        for k in range(7):
            r[f"HuMoment_{k}"][i] = hu[k]
        r["InertiaTensor_0_0"][i] = T[0, 0]
        r["InertiaTensor_0_1"][i] = T[0, 1]
        r["InertiaTensor_1_0"][i] = T[1, 0]
        r["InertiaTensor_1_1"][i] = T[1, 1]
        r["InertiaTensorEigenvalues_0"][i] = l1
        r["InertiaTensorEigenvalues_1"][i] = l2
    return res


def get_feret(masks, pixels=None):
    labels = np.asarray(masks)
    idx = _indices(labels)
    mn, mx = np.full(len(idx), np.nan), np.full(len(idx), np.nan)
    for i, sl in enumerate(ndi.find_objects(labels.astype(np.int32), max_label=len(idx))):
        if sl is None:
            continue
        mn[i], mx[i] = feret_diameters(labels[sl] == (i + 1))
    return {"MinFeretDiameter": mn, "MaxFeretDiameter": mx}


def get_core_measurements():
    """cp_measure.bulk.get_core_measurements() restated: name -> f(masks, pixels, **kw)."""
    from oracle.granularity_restated import get_granularity
    from oracle.radial_restated import get_radial_distribution
    from oracle.texture_restated import get_texture
    from oracle.zernike_restated import get_radial_zernikes, get_zernike

    return {
        "radial_distribution": get_radial_distribution,
        "radial_zernikes": get_radial_zernikes,
        "intensity": get_intensity,
        "sizeshape": get_sizeshape,
        "zernike": get_zernike,
        "feret": get_feret,
        "texture": get_texture,
        "granularity": get_granularity,
    }


# --------------------------------------------------------------------------------------
# colocalisation  (CellProfiler MeasureColocalization, per-object branch)
# call convention of wrap_cp_corr_features (loaders.py:153-167): fun(pixels1, pixels2, mask)
# --------------------------------------------------------------------------------------


def _masked(pixels_1, pixels_2, masks):
    labels = np.asarray(masks).astype(np.int32)
    m = labels > 0
    n = int(labels.max()) if labels.size else 0
    lrange = np.arange(n, dtype=np.int32) + 1
    # float64 throughout: CellProfiler images are floats; integer inputs must not wrap
    return labels[m], np.asarray(pixels_1)[m].astype(np.float64), np.asarray(pixels_2)[m].astype(np.float64), lrange


def get_correlation_pearson(pixels_1, pixels_2, masks):
    labels, fi, si, lrange = _masked(pixels_1, pixels_2, masks)
    n = len(lrange)
    corr, slope = np.full(n, np.nan), np.full(n, np.nan)
    if n and len(labels):
        mean1 = _fix(ndi.mean(fi, labels, lrange))
        mean2 = _fix(ndi.mean(si, labels, lrange))
        x = fi - mean1[labels - 1]
        y = si - mean2[labels - 1]
        sxx = _fix(ndi.sum(x * x, labels, lrange))
        syy = _fix(ndi.sum(y * y, labels, lrange))
        sxy = _fix(ndi.sum(x * y, labels, lrange))
        with np.errstate(invalid="ignore", divide="ignore"):
            corr = sxy / (np.sqrt(sxx) * np.sqrt(syy))
            slope = sxy / sxx  # least-squares A of  A*i1 + B = i2
        cnt = _fix(ndi.sum(np.ones_like(fi), labels, lrange))
        corr[cnt == 0] = np.nan
        slope[cnt == 0] = np.nan
    return {"Correlation_Pearson": corr, "Correlation_Slope": slope}


def _threshold_sums(labels, fi, si, lrange, thr):
    tff = (thr / 100) * _fix(ndi.maximum(fi, labels, lrange))
    tss = (thr / 100) * _fix(ndi.maximum(si, labels, lrange))
    a1 = fi >= tff[labels - 1]
    a2 = si >= tss[labels - 1]
    combined = a1 & a2
    tot_fi = _fix(ndi.sum(fi[a1], labels[a1], lrange))
    tot_si = _fix(ndi.sum(si[a2], labels[a2], lrange))
    return combined, tot_fi, tot_si


def get_correlation_manders_fold(pixels_1, pixels_2, masks, thr=15):
    labels, fi, si, lrange = _masked(pixels_1, pixels_2, masks)
    n = len(lrange)
    M1, M2 = np.zeros(n), np.zeros(n)
    if n and len(labels):
        combined, tot_fi, tot_si = _threshold_sums(labels, fi, si, lrange, thr)
        if combined.any():
            with np.errstate(invalid="ignore", divide="ignore"):
                M1 = _fix(ndi.sum(fi[combined], labels[combined], lrange)) / tot_fi
                M2 = _fix(ndi.sum(si[combined], labels[combined], lrange)) / tot_si
    return {"Correlation_Manders_1": M1, "Correlation_Manders_2": M2}


def get_correlation_rwc(pixels_1, pixels_2, masks, thr=15):
    labels, fi, si, lrange = _masked(pixels_1, pixels_2, masks)
    n = len(lrange)
    R1, R2 = np.zeros(n), np.zeros(n)
    if n and len(labels):
        combined, tot_fi, tot_si = _threshold_sums(labels, fi, si, lrange, thr)
        rank1 = np.lexsort((labels, fi))
        rank2 = np.lexsort((labels, si))
        u1 = np.hstack([[False], fi[rank1[:-1]] != fi[rank1[1:]]])
        u2 = np.hstack([[False], si[rank2[:-1]] != si[rank2[1:]]])
        s1, s2 = np.cumsum(u1), np.cumsum(u2)
        im1 = np.zeros(fi.shape, dtype=int)
        im2 = np.zeros(si.shape, dtype=int)
        im1[rank1] = s1
        im2[rank2] = s2
        R = max(im1.max(), im2.max()) + 1
        weight = (R - np.abs(im1 - im2)) * 1.0 / R
        if combined.any():
            w = weight[combined]
            with np.errstate(invalid="ignore", divide="ignore"):
                R1 = _fix(ndi.sum(fi[combined] * w, labels[combined], lrange)) / tot_fi
                R2 = _fix(ndi.sum(si[combined] * w, labels[combined], lrange)) / tot_si
    return {"Correlation_RWC_1": R1, "Correlation_RWC_2": R2}


def _pearsonr(x, y):
    """scipy.stats.pearsonr's statistic; nan for constant input (no ValueError)."""
    xm, ym = x - x.mean(), y - y.mean()
    nx, ny = np.sqrt((xm * xm).sum()), np.sqrt((ym * ym).sum())
    if nx == 0 or ny == 0:
        return np.nan
    return max(min(float((xm / nx) @ (ym / ny)), 1.0), -1.0)


def bisection_costes(fi, si, scale_max=255):
    """CellProfiler MeasureColocalization.bisection_costes ("Faster" mode)."""
    non_zero = (fi > 0) | (si > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        xvar = np.var(fi[non_zero], ddof=1)
        yvar = np.var(si[non_zero], ddof=1)
        xmean = np.mean(fi[non_zero])
        ymean = np.mean(si[non_zero])
        zvar = np.var(fi[non_zero] + si[non_zero], ddof=1)
        covar = 0.5 * (zvar - (xvar + yvar))
        denom = 2 * covar
        num = (yvar - xvar) + np.sqrt((yvar - xvar) * (yvar - xvar) + 4 * (covar * covar))
        a = num / denom
        b = ymean - a * xmean
    left, right = 1, scale_max
    mid = ((right - left) // (6 / 5)) + left
    lastmid = 0
    valid = 1
    while lastmid != mid:
        thr_fi_c = mid / scale_max
        thr_si_c = (a * thr_fi_c) + b
        combt = (fi < thr_fi_c) | (si < thr_si_c)
        if np.count_nonzero(combt) <= 2:
            left = mid - 1
        else:
            cost = _pearsonr(fi[combt], si[combt])
            if cost < 0:
                left = mid - 1
            elif cost >= 0:
                right = mid + 1
                valid = mid
        lastmid = mid
        if right - left > 6:
            mid = ((right - left) // (6 / 5)) + left
        else:
            mid = ((right - left) // 2) + left
    thr_fi_c = (valid - 1) / scale_max
    thr_si_c = (a * thr_fi_c) + b
    return thr_fi_c, thr_si_c


def get_correlation_costes(pixels_1, pixels_2, masks, scale_max=255):
    """Costes' automatic threshold (bisection) then Manders-style fractions above it.
    As called by the reference there is one object per mask, so the "whole image" threshold of
    CellProfiler is that object's own threshold; for multi-label masks it is evaluated per label."""
    labels, fi, si, lrange = _masked(pixels_1, pixels_2, masks)
    n = len(lrange)
    C1, C2 = np.zeros(n), np.zeros(n)
    for i, lab in enumerate(lrange):
        sel = labels == lab
        if not sel.any():
            continue
        f, s = fi[sel], si[sel]
        t1, t2 = bisection_costes(f, s, scale_max)
        a1, a2 = f > t1, s > t2
        comb = a1 & a2
        tot1 = f[f >= t1].sum() if a1.any() else 0.0
        tot2 = s[s >= t2].sum() if a2.any() else 0.0
        if comb.any():
            with np.errstate(invalid="ignore", divide="ignore"):
                C1[i] = np.float64(f[comb].sum()) / np.float64(tot1)
                C2[i] = np.float64(s[comb].sum()) / np.float64(tot2)
    return {"Correlation_Costes_1": C1, "Correlation_Costes_2": C2}


def get_correlation_measurements():
    """cp_measure.bulk.get_correlation_measurements() restated: name -> f(pixels_1, pixels_2, masks, **kw)."""
    return {
        "pearson": get_correlation_pearson,
        "manders_fold": get_correlation_manders_fold,
        "rwc": get_correlation_rwc,
        "costes": get_correlation_costes,
    }
