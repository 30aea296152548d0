"""
Deterministic synthetic TCZYX stacks + ground-truth label images.

Nothing here comes from the reference: it is the build's own generator, shared
by tests, the oracle's CPU baseline and bench.py so that every leg sees the
same bytes.

Recipe (per field of view, `seed = 20260821 + 1000*config + fov`, PCG64):
  * nuclei  = non-overlapping ellipses on a jittered grid, semi-axes U[8,16] px;
  * cells   = nucleus dilated by U[6,14] px, clipped by nearest-seed Voronoi;
  * pixels  = uint16: background 400 + N(0,30^2); per-object signal
              A*(1-r^2)^0.5 with A~U[2000,20000] per channel, plus correlated
              texture (Gaussian-filtered noise sigma=2, amplitude 0.15 A) and
              Poisson shot noise; clipped to [0, 65535].
  * analytic flows for the segmentation leg: unit vectors towards each
    object's centre, cellprob = +6 inside / -6 outside.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage

BASE_SEED = 20260821


def fov_seed(config: int, fov: int) -> int:
    return BASE_SEED + 1000 * config + fov


def _grid_centres(rng, shape, n_target, margin):
    """Jittered-grid seeds; pitch chosen so that ~n_target cells fit."""
    Y, X = shape
    pitch = max(int(np.sqrt(Y * X / max(n_target, 1))), 2 * margin + 2)
    ys = np.arange(pitch // 2, Y - margin, pitch)
    xs = np.arange(pitch // 2, X - margin, pitch)
    cy, cx = np.meshgrid(ys, xs, indexing="ij")
    cy = cy.ravel().astype(np.float64)
    cx = cx.ravel().astype(np.float64)
    jit = max((pitch - 2 * margin) / 2.0 - 1.0, 0.0)
    cy += rng.uniform(-jit, jit, cy.size)
    cx += rng.uniform(-jit, jit, cx.size)
    keep = (cy > margin) & (cy < Y - margin) & (cx > margin) & (cx < X - margin)
    return cy[keep], cx[keep], pitch


def make_labels(rng, shape, n_target, with_cells=True):
    """Return (nuclei u16 [Y,X], cells u16 [Y,X], params dict)."""
    Y, X = shape
    cy, cx, pitch = _grid_centres(rng, shape, n_target, margin=17)
    n = cy.size
    # keep ellipses inside their grid cell so they never overlap
    rmax = min(16.0, pitch / 2.0 - 1.5)
    rmin = min(8.0, rmax)
    a = rng.uniform(rmin, rmax, n)
    b = rng.uniform(rmin, rmax, n)
    th = rng.uniform(0, np.pi, n)
    grow = rng.uniform(6.0, 14.0, n)
    nuclei = np.zeros(shape, np.uint16)
    yy, xx = np.mgrid[0:Y, 0:X]
    for i in range(n):
        r = int(np.ceil(max(a[i], b[i]))) + 1
        y0, y1 = max(int(cy[i]) - r, 0), min(int(cy[i]) + r + 2, Y)
        x0, x1 = max(int(cx[i]) - r, 0), min(int(cx[i]) + r + 2, X)
        dy = yy[y0:y1, x0:x1] - cy[i]
        dx = xx[y0:y1, x0:x1] - cx[i]
        u = (dx * np.cos(th[i]) + dy * np.sin(th[i])) / a[i]
        v = (-dx * np.sin(th[i]) + dy * np.cos(th[i])) / b[i]
        m = (u * u + v * v) <= 1.0
        sub = nuclei[y0:y1, x0:x1]
        sub[m & (sub == 0)] = i + 1
    cells = None
    if with_cells:
        # distance to every nucleus + index of the nearest one (Voronoi clip)
        dist, (iy, ix) = ndimage.distance_transform_edt(nuclei == 0, return_indices=True)
        nearest = nuclei[iy, ix]
        lim = np.zeros(n + 1)
        lim[1:] = grow
        cells = np.where(dist <= lim[nearest], nearest, 0).astype(np.uint16)
    params = dict(cy=cy, cx=cx, a=a, b=b, theta=th, grow=grow)
    return nuclei, cells, params


def make_pixels(rng, labels, params, n_channels, n_z=1):
    """uint16 [C,Z,Y,X] rendered from `labels` (the larger object set)."""
    Y, X = labels.shape
    n = int(labels.max())
    out = np.empty((n_channels, n_z, Y, X), np.uint16)
    yy, xx = np.mgrid[0:Y, 0:X]
    cy = np.concatenate([[0.0], params["cy"]])
    cx = np.concatenate([[0.0], params["cx"]])
    # normalised radius inside each object (0 at centre, ~1 at the rim)
    dist_in = ndimage.distance_transform_edt(labels > 0)
    rmax = np.maximum(ndimage.maximum(dist_in, labels, np.arange(n + 1)), 1.0)
    r2 = np.clip(1.0 - dist_in / rmax[labels], 0.0, 1.0) ** 2
    del cy, cx, yy, xx
    inside = labels > 0
    for c in range(n_channels):
        amp = np.zeros(n + 1)
        amp[1:] = rng.uniform(2000.0, 20000.0, n)
        A = amp[labels]
        tex = ndimage.gaussian_filter(rng.standard_normal((Y, X)), 2.0)
        tex /= max(tex.std(), 1e-9)
        for z in range(n_z):
            zfall = 1.0 - 0.15 * abs(z - (n_z - 1) / 2.0)
            sig = A * zfall * (np.sqrt(np.clip(1.0 - r2, 0.0, 1.0)) + 0.15 * tex) * inside
            sig = np.clip(sig, 0.0, None)
            img = 400.0 + rng.normal(0.0, 30.0, (Y, X)) + rng.poisson(sig)
            out[c, z] = np.clip(np.rint(img), 0, 65535).astype(np.uint16)
    return out


def analytic_flows(labels):
    """Network-scale outputs for a label image: dP f32 [2,Y,X] (dy,dx) = 5 x unit vectors pointing at each
    object's centre of mass (Cellpose trains on 5 x the normalised diffusion gradient and divides by 5
    before following), cellprob f32 [Y,X] = +6 inside / -6 outside."""
    n = int(labels.max())
    Y, X = labels.shape
    dP = np.zeros((2, Y, X), np.float32)
    prob = np.full((Y, X), -6.0, np.float32)
    if n == 0:
        return dP, prob
    idx = np.arange(1, n + 1)
    com = np.array(ndimage.center_of_mass(labels > 0, labels, idx)).reshape(n, 2)
    cy = np.concatenate([[0.0], com[:, 0]])
    cx = np.concatenate([[0.0], com[:, 1]])
    yy, xx = np.mgrid[0:Y, 0:X]
    inside = labels > 0
    dy = (cy[labels] - yy) * inside
    dx = (cx[labels] - xx) * inside
    nrm = np.sqrt(dy * dy + dx * dx)
    nrm[nrm < 1e-6] = 1.0
    # cellpose flows have magnitude <~1 far from the centre and vanish at the centre
    mag = np.clip(np.sqrt(dy * dy + dx * dx) / 3.0, 0.0, 1.0)
    dP[0] = (5.0 * dy / nrm * mag).astype(np.float32)
    dP[1] = (5.0 * dx / nrm * mag).astype(np.float32)
    prob[inside] = 6.0
    return dP, prob


CONFIGS = {
    # id: (n_fov, C, Z, Y, X, n_target, segment_channel)
    1: dict(n_fov=1, C=2, Z=1, Y=512, X=512, n_target=60, seg_channel=1),
    2: dict(n_fov=256, C=5, Z=1, Y=1024, X=1024, n_target=250, seg_channel=0),
    3: dict(n_fov=2048, C=5, Z=1, Y=1024, X=1024, n_target=250, seg_channel=0),
    4: dict(n_fov=1, C=1, Z=5, Y=512, X=512, n_target=60, seg_channel=0, T=200),
    5: dict(n_fov=1, C=2, Z=32, Y=512, X=512, n_target=100, seg_channel=0),
}


def make_fov(config: int, fov: int = 0, shape=None, n_channels=None, n_z=None, n_target=None):
    """One synthetic field of view.

    Returns dict(pixels u16 [C,Z,Y,X], nuclei u16 [Y,X], cells u16 [Y,X], params).
    Keyword overrides shrink the case for unit tests while keeping the seed rule.
    """
    cfg = CONFIGS[config]
    shape = shape or (cfg["Y"], cfg["X"])
    C = n_channels or cfg["C"]
    Z = n_z or cfg["Z"]
    nt = n_target or cfg["n_target"]
    rng = np.random.default_rng(fov_seed(config, fov))
    nuclei, cells, params = make_labels(rng, shape, nt)
    pixels = make_pixels(rng, cells, params, C, Z)
    return dict(pixels=pixels, nuclei=nuclei, cells=cells, params=params)


def ellipsoid_planes(labels, n_z: int, seed: int = 0):
    """Per-plane label images [Z,Y,X] of ellipsoids whose mid-plane cross-sections are the objects of `labels` [Y,X]: object k
    sits at depth zc_k with half-height az_k and its section shrinks towards its poles (normalised in-plane radius <=
    sqrt(1 - ((z - zc) / az)^2)).  Planes keep the object ids of `labels` (an object is simply absent above and below its
    poles), which is what a 3-D ground truth looks like; each plane relabelled on its own is what a per-plane segmenter sees."""
    rng = np.random.default_rng(seed)
    n = int(labels.max())
    dist_in = ndimage.distance_transform_edt(labels > 0)
    rmax = np.maximum(ndimage.maximum(dist_in, labels, np.arange(n + 1)), 1.0)
    r = np.where(labels > 0, 1.0 - dist_in / rmax[labels], 2.0)  # 0 at the centre, ~1 at the rim
    zc = np.concatenate([[0.0], rng.uniform(0.25 * n_z, 0.75 * n_z, n)])
    az = np.concatenate([[1.0], rng.uniform(max(1.5, 0.1 * n_z), max(2.5, 0.3 * n_z), n)])
    out = np.zeros((n_z, *labels.shape), labels.dtype)
    for z in range(n_z):
        t = 1.0 - ((z - zc) / az) ** 2
        lim = np.sqrt(np.clip(t, 0.0, None))
        keep = (labels > 0) & (t[labels] > 0) & (r <= lim[labels])
        out[z] = np.where(keep, labels, 0)
    return out


def write_tiff(path, plane: np.ndarray, compression: str | None = None, rows_per_strip: int = 64) -> None:
    """Minimal little-endian baseline TIFF writer (one 2-D grayscale page; no compression or Deflate) so that
    synthetic stacks can be laid out on disk the way a microscope leaves them, one file per (t, c, z) plane."""
    import struct
    import zlib

    plane = np.ascontiguousarray(plane)
    assert plane.ndim == 2 and plane.dtype.kind in "uif" and plane.dtype.itemsize in (1, 2, 4)
    h, w = plane.shape
    strips = [plane[r : r + rows_per_strip].tobytes() for r in range(0, h, rows_per_strip)]
    if compression == "deflate":
        strips = [zlib.compress(s, 1) for s in strips]
    elif compression is not None:
        raise ValueError("compression must be None or 'deflate'")
    n = len(strips)
    data_at = 8
    offsets, at = [], data_at
    for s in strips:
        offsets.append(at)
        at += len(s) + (len(s) & 1)
    arrays_at = at
    offs_at, counts_at = (arrays_at, arrays_at + 4 * n) if n > 1 else (0, 0)
    ifd_at = arrays_at + (8 * n if n > 1 else 0)
    fmt = {"u": 1, "i": 2, "f": 3}[plane.dtype.kind]
    tags = [
        (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 8 * plane.dtype.itemsize), (259, 3, 1, 8 if compression else 1),
        (262, 3, 1, 1), (273, 4, n, offs_at if n > 1 else offsets[0]), (277, 3, 1, 1), (278, 4, 1, rows_per_strip),
        (279, 4, n, counts_at if n > 1 else len(strips[0])), (339, 3, 1, fmt),
    ]
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_at))
        for s in strips:
            f.write(s)
            if len(s) & 1:
                f.write(b"\0")
        if n > 1:
            f.write(struct.pack(f"<{n}I", *offsets))
            f.write(struct.pack(f"<{n}I", *(len(s) for s in strips)))
        f.write(struct.pack("<H", len(tags)))
        for tag, typ, count, value in tags:
            f.write(struct.pack("<HHI", tag, typ, count))
            f.write(struct.pack("<HH", value, 0) if typ == 3 and count == 1 else struct.pack("<I", value))
        f.write(struct.pack("<I", 0))


def trap_image(seed: int = 11, shape=(512, 512), spacing: int = 128, first: int = 96, jitter: int = 6):
    """Brightfield-like frame with a jittered grid of identical 'traps' (a textured ellipse with a bright bar on one
    side, so that rotations of the template differ) on a flat, weakly noisy background — the input of trap detection
    (config 4).  Returns (uint16 image, list of (y, x) centres)."""
    rng = np.random.default_rng(seed)
    img = 3000 + rng.normal(0, 30, shape)
    yy, xx = np.mgrid[0 : shape[0], 0 : shape[1]]
    centres = []
    for cy in range(first, shape[0], spacing):
        for cx in range(first, shape[1], spacing):
            y = cy + int(rng.integers(-jitter, jitter + 1))
            x = cx + int(rng.integers(-jitter, jitter + 1))
            centres.append((y, x))
            inside = ((yy - y) / 26.0) ** 2 + ((xx - x) / 20.0) ** 2 <= 1
            img[inside] += rng.normal(0, 900, int(inside.sum())) + 1500
            bar = (np.abs(yy - (y - 14)) <= 3) & (np.abs(xx - x) <= 16)
            img[bar] += 4000
    return np.clip(img, 0, 65535).astype(np.uint16), centres


def make_timelapse(T: int = 10, seed: int = 11, n_z: int = 5, max_step: int = 2, shape=(512, 512)):
    """Config-4-shaped position: a trap grid (`trap_image`) with 1 -> 3 yeast-like cells per trap appearing and growing
    over time, the whole sample drifting by an integer random walk of at most `max_step` px per timepoint, Z planes of one
    channel with a mild focus fall-off.  Returns dict(pixels u16 [T,1,Z,Y,X], labels u16 [T,Y,X] (ground truth, full
    frame, labels = 3*trap + cell + 1 renumbered densely per frame), shifts int [T,2] cumulative (dy, dx), centres)."""
    base, centres = trap_image(seed=seed, shape=shape)
    rng = np.random.default_rng(fov_seed(4, seed))
    Y, X = shape
    yy, xx = np.mgrid[0:Y, 0:X]
    steps = rng.integers(-max_step, max_step + 1, size=(T, 2))
    steps[0] = 0
    shifts = np.cumsum(steps, axis=0)
    n_traps = len(centres)
    # per trap: three cell slots at 120 degrees around a point just below the trap, born at t = 0, T/3, 2T/3
    ang0 = rng.uniform(0, 2 * np.pi, n_traps)
    r0 = rng.uniform(6.0, 8.0, (n_traps, 3))
    amp = rng.uniform(500.0, 900.0, (n_traps, 3))
    births = np.array([0, max(1, T // 3), max(2, (2 * T) // 3)])
    pixels = np.zeros((T, 1, n_z, Y, X), np.uint16)
    labels = np.zeros((T, Y, X), np.uint16)
    for t in range(T):
        frame = base.astype(np.float64)
        lab = np.zeros(shape, np.uint16)
        nxt = 0
        for k, (cy, cx) in enumerate(centres):
            for j in range(3):
                if t < births[j]:
                    continue
                r = r0[k, j] + 0.15 * (t - births[j])
                a = ang0[k] + j * 2 * np.pi / 3
                y0, x0 = cy + 6 + 13.0 * np.sin(a), cx + 13.0 * np.cos(a)
                d2 = (yy - y0) ** 2 + (xx - x0) ** 2
                m = (d2 <= r * r) & (lab == 0)
                if not m.any():
                    continue
                nxt += 1
                lab[m] = nxt
                frame[m] += amp[k, j] * np.sqrt(np.clip(1.0 - d2[m] / (r * r), 0, 1))
        frame = np.roll(frame, tuple(shifts[t]), axis=(0, 1))
        labels[t] = np.roll(lab, tuple(shifts[t]), axis=(0, 1))
        for z in range(n_z):
            plane = frame * (1.0 - 0.04 * abs(z - n_z // 2)) + rng.normal(0, 10.0, shape)
            pixels[t, 0, z] = np.clip(plane, 0, 65535).astype(np.uint16)
    return dict(pixels=pixels, labels=labels, shifts=shifts, centres=centres)
