"""
TEST INFRASTRUCTURE — restatement of the tile stager's host arithmetic:
`if_out_of_bounds_pad` (src/aliby/tile/tiler.py:601-650) and the per-channel assembly of
`Tiler.get_fczyx` / `get_tp_channel` (tiler.py:309-366).  np.pad(mode="median") is NumPy's own.
PINNED indirectly: Tile.as_range is checked against the reference in tests/test_oracle_golden.py.
"""

from __future__ import annotations

import numpy as np


def if_out_of_bounds_pad(pixels, slices, max_padding=0.25):
    max_yx = pixels.shape[-2:]
    y, x = [slice(max(0, s.start), min(ub, s.stop)) for s, ub in zip(slices, max_yx)]
    padding = np.array([(-min(0, s.start), -min(0, ub - s.stop)) for s, ub in zip(slices, max_yx)])
    tile = pixels[:, y, x]
    if padding.any():
        tile_shape = [s.stop - s.start for s in slices]
        if (padding / 0.25 > tile_shape).any():
            tile = np.full((pixels.shape[0], *tile_shape), np.nan)
        else:
            tile = np.pad(tile, [[0, 0]] + padding.tolist(), "median")
    return tile


def get_fczyx(stack_czyx, ranges):
    """stack [C,Z,Y,X], ranges = [(yslice, xslice), ...] -> [F,C,Z,h,w]."""
    channels = []
    for c in range(stack_czyx.shape[0]):
        channels.append(np.stack([if_out_of_bounds_pad(stack_czyx[c], r) for r in ranges]))
    return np.swapaxes(np.array(channels), 0, 1)


def relabel_sequential(labels):
    """skimage.segmentation.relabel_sequential(labels)[0] with offset=1 (dispatch.py:223)."""
    uniq = np.unique(labels)
    uniq = uniq[uniq != 0]
    fwd = np.zeros(int(labels.max()) + 1 if labels.size else 1, dtype=labels.dtype)
    fwd[uniq] = np.arange(1, len(uniq) + 1)
    return fwd[labels]
