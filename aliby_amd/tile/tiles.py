"""
Tile bookkeeping: centres, sizes and drift-corrected ranges.

Behaviour follows src/aliby/tile/tiles.py (TileLocations 8-90, Tile 93-166): a tile is a centre plus
a (rows, cols) size; its window at time tp is the centre minus the cumulative drift up to and including
tp (truncated towards zero), minus half the size.
"""

from __future__ import annotations

import numpy as np


def _pair(v):
    return (v, v) if isinstance(v, int) else v


class Tile:
    """One tile; `parent_class.drifts` is shared with the owning TileLocations."""

    def __init__(self, centre, parent, size, max_size):
        self.centre, self.parent_class = centre, parent
        self.size, self.max_size = size, max_size
        self.half_size = [extent // 2 for extent in size]

    def centre_at_time(self, tp: int):
        drift = np.sum(self.parent_class.drifts[: tp + 1], axis=0)
        return list((self.centre - drift).astype(int))

    def as_tile(self, tp: int):
        """(row0, col0, n_rows, n_cols) at `tp`."""
        cy, cx = self.centre_at_time(tp)
        return (int(cy - self.half_size[0]), int(cx - self.half_size[1]), *self.size)

    def as_range(self, tp: int):
        y0, x0, ny, nx = self.as_tile(tp)
        return slice(y0, y0 + ny), slice(x0, x0 + nx)


class TileLocations:
    def __init__(self, initial_location, tile_size=None, max_size=1200, drifts=None):
        self.tile_size, self.max_size = _pair(tile_size), _pair(max_size)
        self.initial_location = initial_location
        self.drifts = drifts if drifts is not None else []
        extent = self.tile_size or self.max_size
        self.tiles = [Tile(centre, self, extent, self.max_size) for centre in initial_location]

    @classmethod
    def from_tiler_init(cls, initial_location, tile_size=None, max_size=1200):
        return cls(initial_location, tile_size, max_size, drifts=[])

    def __len__(self):
        return len(self.tiles)

    def __iter__(self):
        return iter(self.tiles)

    @property
    def shape(self):
        """(number of tiles, number of timepoints with a drift)."""
        return len(self.tiles), len(self.drifts)

    def centres_at_time(self, tp: int):
        return np.array([tile.centre_at_time(tp) for tile in self.tiles])

    def to_dict(self, tp: int):
        out = {"drifts": np.expand_dims(self.drifts[tp], axis=0)}
        if tp == 0:
            out = {"trap_locations": self.initial_location, "attrs/tile_size": self.tile_size,
                   "attrs/max_size": self.max_size, **out}
        return out
