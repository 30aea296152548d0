"""
TEST INFRASTRUCTURE — CPU restatement of what `cellpose.models.CellposeModel.eval` does around the
network for the reference's call (src/aliby/segment/dispatch.py:208-215: 2-D, do_3D=False,
stitch_threshold=0.0, normalize=True, z_axis=None) plus the reference's own pre/post steps
(dispatch.py:192-206 channel select / Z max-projection, 216-234 max over axis 0, relabel_sequential,
overflow check, uint16 cast).

cellpose 4.0.6 (uv.lock:130-131) is NOT in /root/reference nor installed and its weights cannot be
fetched: PARITY UNPINNED.  The steps follow the published Cellpose algorithm:
  normalize99         per-image 1st/99th percentile rescale to float32 (0 if the range is < 1e-3);
  make_tiles /        224-px tiles (v3 U-Net family) with >= 10 % overlap, sigmoid taper blending;
  average_tiles
  follow_flows        200 Euler steps of p += bilinear(dP/5)(p) with torch grid_sample's
                      align_corners=False index mapping and zero padding, positions clamped;
  get_masks           histogram of end points (padded by 20), 5x5 max-pool seeds with > 10 points,
                      5 rounds of 3x3 dilation limited to bins with > 2 points inside an 11x11 window,
                      label = seed owning the end-point bin, masks > 40 % of the image removed,
                      labels renumbered in order of first appearance (fastremap.renumber);
  remove_bad_flow_masks  flows re-derived from the masks by heat diffusion from each mask's centre,
                      masks whose mean squared flow error exceeds 0.4 are removed;
  fill_holes_and_remove_small_masks  min_size = 15.
Where the original leaves an order to an unstable sort (seed overlap), the restatement fixes it:
seed priority = (points in the bin, then raster position).  Every float32 operation is written out in
the order the HIP kernels use, so labels are compared bit-for-bit.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

f32 = np.float32


# ------------------------------------------------------------------------------------------
# reference-side pre/post (dispatch.py)
# ------------------------------------------------------------------------------------------
def select_and_project(pixels_fczyx, channel):
    px = pixels_fczyx[:, channel]  # FZYX
    if px.shape[1] > 1:
        return px.max(axis=1)
    return px[:, 0]


def relabel_sequential(labels):
    uniq = np.unique(labels)
    uniq = uniq[uniq != 0]
    fwd = np.zeros(int(labels.max()) + 1 if labels.size else 1, dtype=np.int64)
    fwd[uniq] = np.arange(1, len(uniq) + 1)
    return fwd[labels]


def finish_labels(labels):
    """dispatch.py:216-234."""
    if labels.ndim == 3:
        labels = relabel_sequential(labels.max(axis=0))
    elif not 1 < labels.ndim < 4:
        raise Exception(f"Segmentation yielded {labels.ndim} dimensions instead of 3")
    if labels.size and labels.max() >= np.iinfo(np.uint16).max:
        raise OverflowError(f"Segmentation produced {labels.max()} labels; uint16 cast unsafe.")
    return labels.astype(np.uint16, copy=False)


# ------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------
def percentile_linear(sorted_vals, q):
    """numpy.percentile(method='linear') on an already sorted 1-D array, float64 arithmetic."""
    n = len(sorted_vals)
    pos = (n - 1) * (q / 100.0)
    lo = int(np.floor(pos))
    hi = min(lo + 1, n - 1)
    t = pos - lo
    a, b = float(sorted_vals[lo]), float(sorted_vals[hi])
    return a + (b - a) * t


def normalize99(img, lower=1.0, upper=99.0):
    """cellpose.transforms.normalize99 on one 2-D image -> float32."""
    s = np.sort(img.ravel())
    x01 = percentile_linear(s, lower)
    x99 = percentile_linear(s, upper)
    if x99 - x01 > 1e-3:
        return ((img.astype(np.float64) - x01) / (x99 - x01)).astype(f32)
    return np.zeros(img.shape, f32)


# ------------------------------------------------------------------------------------------
# tiling / blending
# ------------------------------------------------------------------------------------------
def tile_starts(L, bsize=224, tile_overlap=0.1):
    tile_overlap = min(0.5, max(0.05, tile_overlap))
    b = min(bsize, L)
    n = 1 if L <= bsize else int(np.ceil((1.0 + 2 * tile_overlap) * L / bsize))
    return np.linspace(0, L - b, n).astype(int), b


def pad_to_16(Ly, Lx, div=16, extra=1):
    """cellpose.transforms.pad_image_ND amounts: returns (ypad1, ypad2, xpad1, xpad2)."""
    Lpad = int(div * np.ceil(Ly / div) - Ly)
    ypad1 = extra * div // 2 + Lpad // 2
    ypad2 = extra * div // 2 + Lpad - Lpad // 2
    Lpad = int(div * np.ceil(Lx / div) - Lx)
    xpad1 = extra * div // 2 + Lpad // 2
    xpad2 = extra * div // 2 + Lpad - Lpad // 2
    return ypad1, ypad2, xpad1, xpad2


def taper_mask(ly=224, lx=None, sig=7.5):
    """cellpose transforms._taper_mask(ly, lx, sig): a square window of side max(224, ly, lx) — the product of two 1-D sigmoids
    falling off 20 px before the edge — cropped about its centre to ly x lx.  (Tiles smaller than 224, i.e. small images, take the
    CENTRE of the 224 window, not a narrower window.)"""
    lx = ly if lx is None else lx
    side = max(224, ly, lx)
    d = np.abs(np.arange(side) - (side - 1) / 2.0)
    w = 1.0 / (1.0 + np.exp((d - (side / 2 - 20)) / sig))
    full = w * w[:, np.newaxis]
    c = side // 2
    return full[c - ly // 2 : c + ly // 2 + ly % 2, c - lx // 2 : c + lx // 2 + lx % 2].astype(f32)


def make_tiles(img_chw, bsize=224, tile_overlap=0.1):
    """img [nchan, Ly, Lx] (already padded) -> (tiles [ny*nx, nchan, b, b], ystarts, xstarts)."""
    _, Ly, Lx = img_chw.shape
    ys, by = tile_starts(Ly, bsize, tile_overlap)
    xs, bx = tile_starts(Lx, bsize, tile_overlap)
    tiles = np.stack([img_chw[:, y : y + by, x : x + bx] for y in ys for x in xs])
    return tiles, ys, xs


def average_tiles(y_tiles, ys, xs, Ly, Lx):
    """Weighted average of overlapping tile outputs [ntiles, nout, b, b] -> [nout, Ly, Lx] (float32)."""
    by, bx = y_tiles.shape[-2:]
    mask = taper_mask(by, bx)
    out = np.zeros((y_tiles.shape[1], Ly, Lx), f32)
    navg = np.zeros((Ly, Lx), f32)
    k = 0
    for y in ys:
        for x in xs:
            out[:, y : y + by, x : x + bx] += y_tiles[k] * mask
            navg[y : y + by, x : x + bx] += mask
            k += 1
    return out / navg


# ------------------------------------------------------------------------------------------
# dynamics
# ------------------------------------------------------------------------------------------
def follow_flows(dP, inds, niter=200):
    """dP float32 [2,Y,X] already divided by 5 and masked; inds = (ys, xs) of foreground pixels.
    Returns final positions float32 [2, n] as (y, x) in pixel units.  All arithmetic float32."""
    H, W = dP.shape[1:]
    ys, xs = inds
    sx, sy = f32(W - 1), f32(H - 1)
    # normalised flow field and positions (torch grid_sample convention: x first)
    imx = dP[1] * (f32(2.0) / sx)
    imy = dP[0] * (f32(2.0) / sy)
    px = xs.astype(f32) / sx * f32(2.0) - f32(1.0)
    py = ys.astype(f32) / sy * f32(2.0) - f32(1.0)
    Wf, Hf = f32(W), f32(H)
    for _ in range(niter):
        ix = ((px + f32(1.0)) * Wf - f32(1.0)) / f32(2.0)
        iy = ((py + f32(1.0)) * Hf - f32(1.0)) / f32(2.0)
        x0 = np.floor(ix)
        y0 = np.floor(iy)
        x1 = x0 + f32(1.0)
        y1 = y0 + f32(1.0)
        wnw = (x1 - ix) * (y1 - iy)
        wne = (ix - x0) * (y1 - iy)
        wsw = (x1 - ix) * (iy - y0)
        wse = (ix - x0) * (iy - y0)
        x0i, y0i, x1i, y1i = x0.astype(np.int64), y0.astype(np.int64), x1.astype(np.int64), y1.astype(np.int64)

        def tap(field, yy, xx):
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.zeros(len(yy), f32)
            v[ok] = field[yy[ok], xx[ok]]
            return v

        dx = f32(0.0) + tap(imx, y0i, x0i) * wnw
        dx = dx + tap(imx, y0i, x1i) * wne
        dx = dx + tap(imx, y1i, x0i) * wsw
        dx = dx + tap(imx, y1i, x1i) * wse
        dy = f32(0.0) + tap(imy, y0i, x0i) * wnw
        dy = dy + tap(imy, y0i, x1i) * wne
        dy = dy + tap(imy, y1i, x0i) * wsw
        dy = dy + tap(imy, y1i, x1i) * wse
        px = np.minimum(np.maximum(px + dx, f32(-1.0)), f32(1.0))
        py = np.minimum(np.maximum(py + dy, f32(-1.0)), f32(1.0))
    fx = (px + f32(1.0)) * f32(0.5) * sx
    fy = (py + f32(1.0)) * f32(0.5) * sy
    return np.stack([fy, fx]).astype(f32)


def get_masks(p_final, inds, shape0, rpad=20, max_size_fraction=0.4):
    """End points -> labels (uint32 [Y,X]), renumbered by first raster appearance."""
    Y, X = shape0
    ys, xs = inds
    shape = (Y + 2 * rpad, X + 2 * rpad)
    # cellpose keeps float positions and lets torch cast them to long: truncation towards zero
    pty = np.clip((p_final[0] + f32(rpad)), 0, None)
    ptx = np.clip((p_final[1] + f32(rpad)), 0, None)
    pty = np.minimum(pty, f32(Y + rpad - 1)).astype(np.int64)
    ptx = np.minimum(ptx, f32(X + rpad - 1)).astype(np.int64)
    h1 = np.zeros(shape, np.int64)
    np.add.at(h1, (pty, ptx), 1)
    hmax = ndi.maximum_filter(h1, size=5, mode="constant", cval=-1)
    seeds = np.nonzero((h1 - hmax > -1e-6) & (h1 > 10))
    M0 = np.zeros(shape0, np.uint32)
    if len(seeds[0]) == 0:
        return M0
    sy, sx = seeds
    npts = h1[sy, sx]
    flat = sy * shape[1] + sx
    prio = npts.astype(np.int64) * (shape[0] * shape[1]) + flat  # ascending: later (larger) overwrite earlier
    order = np.argsort(prio, kind="stable")
    M1 = np.zeros(shape, np.int64)  # holds (priority rank + 1)
    st3 = np.ones((3, 3), bool)
    for rank, k in enumerate(order):
        cy, cx = sy[k], sx[k]
        win = h1[cy - 5 : cy + 6, cx - 5 : cx + 6]
        sm = np.zeros((11, 11), bool)
        sm[5, 5] = True
        for _ in range(5):
            sm = ndi.binary_dilation(sm, st3) & (win > 2)
        wy, wx = np.nonzero(sm)
        M1[wy + cy - 5, wx + cx - 5] = rank + 1
    lab = M1[pty, ptx]
    M0[ys, xs] = lab
    # remove big masks
    uniq, counts = np.unique(M0, return_counts=True)
    big = Y * X * max_size_fraction
    bigc = uniq[counts > big]
    if len(bigc) > 0 and (len(bigc) > 1 or bigc[0] != 0):
        M0[np.isin(M0, bigc[bigc != 0])] = 0
    return renumber_first_appearance(M0)


def renumber_first_appearance(M):
    """fastremap.renumber: 1..n in the order labels are first met in a raster scan; 0 stays 0."""
    flat = M.ravel()
    uniq, first = np.unique(flat, return_index=True)
    keep = uniq != 0
    uniq, first = uniq[keep], first[keep]
    order = np.argsort(first, kind="stable")
    fwd = np.zeros(int(flat.max()) + 1 if flat.size else 1, np.uint32)
    fwd[uniq[order]] = np.arange(1, len(uniq) + 1, dtype=np.uint32)
    return fwd[M]


def mask_centres(masks):
    """Per mask: the pixel of the mask closest to its centre of mass (first one on ties), and extents."""
    slices = ndi.find_objects(masks.astype(np.int64))
    centres, ext = {}, []
    for i, si in enumerate(slices):
        if si is None:
            continue
        sr, sc = si
        yi, xi = np.nonzero(masks[sr, sc] == (i + 1))
        ymed, xmed = yi.mean(), xi.mean()
        imin = ((xi - xmed) ** 2 + (yi - ymed) ** 2).argmin()
        centres[i + 1] = (yi[imin] + sr.start, xi[imin] + sc.start)
        ext.append((sr.stop - sr.start + 1) + (sc.stop - sc.start + 1))
    return centres, ext


def masks_to_flows(masks):
    """Heat-diffusion flows of a label image (cellpose.dynamics.masks_to_flows_gpu): float64."""
    Y, X = masks.shape
    mp = np.pad(masks.astype(np.int64), 1)
    centres, ext = mask_centres(masks)
    mu0 = np.zeros((2, Y, X))
    if not centres:
        return mu0
    n_iter = 2 * max(ext)
    yy, xx = np.nonzero(mp)
    offs = [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (-1, 1), (1, -1), (1, 1)]
    same = [mp[yy + dy, xx + dx] == mp[yy, xx] for dy, dx in offs]
    cy = np.array([c[0] + 1 for c in centres.values()])
    cx = np.array([c[1] + 1 for c in centres.values()])
    T = np.zeros(mp.shape)
    for _ in range(n_iter):
        T[cy, cx] += 1
        acc = np.zeros(len(yy))
        for (dy, dx), ok in zip(offs, same):
            acc = acc + T[yy + dy, xx + dx] * ok
        T[yy, xx] = acc / 9.0
    dy = T[yy + 1, xx] - T[yy - 1, xx]
    dx = T[yy, xx + 1] - T[yy, xx - 1]
    nrm = 1e-60 + np.sqrt(dy * dy + dx * dx)
    mu0[0, yy - 1, xx - 1] = dy / nrm
    mu0[1, yy - 1, xx - 1] = dx / nrm
    return mu0


def flow_errors(masks, dP_net):
    n = int(masks.max())
    if n == 0:
        return np.zeros(0)
    mu = masks_to_flows(masks)
    idx = np.arange(1, n + 1)
    err = np.zeros(n)
    for i in range(2):
        err += np.atleast_1d(ndi.mean((mu[i] - dP_net[i].astype(np.float64) / 5.0) ** 2, masks, index=idx))
    return err


def remove_bad_flow_masks(masks, dP_net, threshold=0.4):
    err = flow_errors(masks, dP_net)
    bad = 1 + np.nonzero(err > threshold)[0]
    out = masks.copy()
    out[np.isin(out, bad)] = 0
    return out


def fill_holes_and_remove_small_masks(masks, min_size=15):
    """Restated with order-independent hole ownership: a pixel inside the holes of several masks goes
    to the mask with the highest label (the original resolves this by processing order)."""
    n = int(masks.max())
    out = np.zeros_like(masks)
    slices = ndi.find_objects(masks.astype(np.int64))
    j = 0
    for i, slc in enumerate(slices):
        if slc is None:
            continue
        msk = masks[slc] == (i + 1)
        npix = msk.sum()
        if min_size > 0 and npix < min_size:
            continue
        filled = ndi.binary_fill_holes(msk)
        j += 1
        sub = out[slc]
        sub[filled] = j
    del n
    return out


def compute_masks(dP, cellprob, niter=200, cellprob_threshold=0.0, flow_threshold=0.4, min_size=15,
                  max_size_fraction=0.4):
    """cellpose.dynamics.compute_masks for one 2-D image.  dP float32 [2,Y,X], cellprob float32 [Y,X]."""
    cp_mask = cellprob > cellprob_threshold
    shape0 = cellprob.shape
    if not cp_mask.any():
        return np.zeros(shape0, np.uint16)
    inds = np.nonzero(cp_mask)
    dPs = (dP * cp_mask) / f32(5.0)
    p_final = follow_flows(dPs.astype(f32), inds, niter=niter)
    mask = get_masks(p_final, inds, shape0, max_size_fraction=max_size_fraction)
    if mask.max() > 0 and flow_threshold is not None and flow_threshold > 0:
        mask = remove_bad_flow_masks(mask, dP, threshold=flow_threshold)
    mask = fill_holes_and_remove_small_masks(mask, min_size=min_size)
    return mask.astype(np.uint16 if mask.max() < 2**16 else np.uint32)
