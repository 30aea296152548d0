"""
Dataset discovery: which files make up which position.

Same public surface as src/aliby/io/dataset.py — `dispatch_dataset` (21-48), `DatasetZarr` (108-124), `DatasetDir`
(127-161), `sort_groups_by_regex` (164-216) — and the same answers: a position is one combination of the regex captures
that are not T, C or Z (well, field of view, ...); inside a position the files come T-major, then C, then Z, which is the
order `ImageList` fills its (T, C, Z) grid in; positions themselves are ordered by the LAST site capture first (the
reference sorts key by key with a stable sort, so the last key sorted dominates).
"""

from __future__ import annotations

import os
import re
import shutil
import time
from pathlib import Path

IMAGE_AXES = "TCZYX"


def walk_files(root) -> list[str]:
    """Every file below `root` as '<dir>/<name>', in os.walk order."""
    found = []
    for folder, _, names in os.walk(root):
        found.extend(f"{folder}/{name}" for name in names)
    return [f for f in found if not f.startswith(".")]


def sort_groups_by_regex(datasets_path, regex: str, capture_order: str, out_dimorder: str = IMAGE_AXES):
    """[{"key": "<site captures joined by __>", "path": [files of that position, ordered]}, ...]"""
    pattern = re.compile(regex)
    rows = [(m.groups(), path) for path in walk_files(datasets_path) if (m := pattern.match(path))]
    site = [i for i, axis in enumerate(capture_order) if axis not in out_dimorder]
    image = [capture_order.index(axis) for axis in out_dimorder if axis in capture_order]
    # least significant key first; every sort is stable, so the result is ordered by site (last capture first), then T, C, Z
    for index in [*reversed(image), *site]:
        rows.sort(key=lambda row: row[0][index])
    positions, current = [], None
    for captures, path in rows:
        label = "__".join(captures[i] for i in site)
        if current is None or current["key"] != label:
            current = {"key": label, "path": []}
            positions.append(current)
        current["path"].append(str(Path(datasets_path) / path))
    assert positions, "No files were found."
    return positions


class _LocalDataset:
    """What both kinds of local data set share: the root path and the acquisition logs lying around in it."""

    _valid_suffixes = ("tiff", "png", "zarr", "tif")
    _valid_meta_suffixes = ("txt", "log")

    def __init__(self, dpath, *args, **kwargs):
        self.path = Path(dpath)
        self._logs = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    dataset = property(lambda self: self.path)
    name = property(lambda self: self.path.name)
    unique_name = property(lambda self: self.path.name)

    @property
    def files(self):
        if self._logs is None:
            self._logs = {f: f for f in self.path.rglob("*") if f.suffix.lstrip(".") in self._valid_meta_suffixes}
        return self._logs

    def cache_logs(self, root_dir):
        for log in self.files:
            shutil.copy(log, Path(root_dir) / log.name)
        return True

    @property
    def date(self):
        return time.strftime("%Y%m%d", time.localtime(os.path.getmtime(self.path)))


DatasetLocalABC = _LocalDataset  # the reference's name for the base class


class DatasetZarr(_LocalDataset):
    """Every directory at the root of a zarr store is a position: {"path": store, "key": group name}."""

    def get_position_ids(self):
        return [{"path": self.path, "key": entry.name} for entry in os.scandir(self.path) if entry.is_dir()]


class DatasetDir(_LocalDataset):
    """Loose image files, nested or not.  `regex` captures, in `capture_order`, C(hannel), W(ell), T(ime point),
    F(ield of view) and Z(-section) from each path."""

    def __init__(self, dpath, regex: str, capture_order: str):
        super().__init__(dpath)
        self.regex, self.capture_order = regex, capture_order

    def get_position_ids(self, regex: str = None, capture_order: str = None):
        return sort_groups_by_regex(self.path, regex or self.regex, capture_order or self.capture_order)


def dispatch_dataset(expt_id, is_zarr: bool = False, **kwargs):
    """A path to a zarr store (`is_zarr=True`) or to a folder of image files (with `regex` and `capture_order`)."""
    if not isinstance(expt_id, (str, Path)):
        raise Exception("Invalid experiment id, it must be a Path")
    root = Path(expt_id)
    assert root.exists(), f"Experiment path does not exist: {root}"
    if is_zarr is True:
        kwargs.pop("is_monozarr", None)
        return DatasetZarr(root, **kwargs)
    return DatasetDir(root, **kwargs)
