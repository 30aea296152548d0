"""
Tile bookkeeping: centres, sizes and drift-corrected ranges.

Mirrors the behaviour of src/aliby/tile/tiles.py (TileLocations 8-90, Tile 93-166): a tile is a
centre plus a (rows, cols) size; its range at time tp is centre - cumulative drift (truncated
towards zero) minus half the size.
"""

from __future__ import annotations

import numpy as np


class Tile:
    def __init__(self, centre, parent, size, max_size):
        self.centre = centre
        self.parent_class = parent
        self.size = size
        self.half_size = [s // 2 for s in size]
        self.max_size = max_size

    def centre_at_time(self, tp: int):
        shift = np.sum(self.parent_class.drifts[: tp + 1], axis=0)
        return list((self.centre - shift).astype(int))

    def as_tile(self, tp: int):
        a, b = self.centre_at_time(tp)
        return (int(a - self.half_size[0]), int(b - self.half_size[1]), *self.size)

    def as_range(self, tp: int):
        a, b, da, db = self.as_tile(tp)
        return slice(a, a + da), slice(b, b + db)


class TileLocations:
    def __init__(self, initial_location, tile_size=None, max_size=1200, drifts=None):
        if isinstance(tile_size, int):
            tile_size = (tile_size, tile_size)
        if isinstance(max_size, int):
            max_size = (max_size, max_size)
        self.tile_size = tile_size
        self.max_size = max_size
        self.initial_location = initial_location
        self.tiles = [Tile(c, self, tile_size or max_size, max_size) for c in initial_location]
        self.drifts = [] if drifts is None else drifts

    def __len__(self):
        return len(self.tiles)

    def __iter__(self):
        yield from self.tiles

    @property
    def shape(self):
        return len(self.tiles), len(self.drifts)

    def to_dict(self, tp: int):
        res = {}
        if tp == 0:
            res["trap_locations"] = self.initial_location
            res["attrs/tile_size"] = self.tile_size
            res["attrs/max_size"] = self.max_size
        res["drifts"] = np.expand_dims(self.drifts[tp], axis=0)
        return res

    def centres_at_time(self, tp: int):
        return np.array([t.centre_at_time(tp) for t in self.tiles])

    @classmethod
    def from_tiler_init(cls, initial_location, tile_size=None, max_size=1200):
        return cls(initial_location, tile_size, max_size, drifts=[])
