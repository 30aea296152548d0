"""
Dataset discovery: which files make up which position.

Mirrors src/aliby/io/dataset.py — `dispatch_dataset` (21-48), `DatasetZarr` (108-124), `DatasetDir` (127-161),
`sort_groups_by_regex` (164-216): positions are the groups of the non-TCZ regex captures (well, field of view), files
inside a position are ordered T-major, then C, then Z, which is the order `ImageList` fills its (T, C, Z) grid in.
"""

from __future__ import annotations

import os
import re
import shutil
import time
from itertools import groupby
from pathlib import Path


def dispatch_dataset(expt_id, is_zarr: bool = False, **kwargs):
    if isinstance(expt_id, (str, Path)):
        expt_path = Path(expt_id)
        assert expt_path.exists(), f"Experiment path does not exist: {expt_path}"
        if is_zarr is True:
            kwargs.pop("is_monozarr", None)
            return DatasetZarr(expt_path, **kwargs)
        return DatasetDir(expt_path, **kwargs)
    raise Exception("Invalid experiment id, it must be a Path")


class DatasetLocalABC:
    _valid_suffixes = ("tiff", "png", "zarr", "tif")
    _valid_meta_suffixes = ("txt", "log")

    def __init__(self, dpath, *args, **kwargs):
        self.path = Path(dpath)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    @property
    def dataset(self):
        return self.path

    @property
    def name(self):
        return self.path.name

    @property
    def unique_name(self):
        return self.path.name

    @property
    def files(self):
        if not hasattr(self, "_files"):
            self._files = {f: f for f in self.path.rglob("*") if str(f).endswith(self._valid_meta_suffixes)}
        return self._files

    def cache_logs(self, root_dir):
        for name, annotation in self.files.items():
            shutil.copy(annotation, Path(root_dir) / name.name)
        return True

    @property
    def date(self):
        return time.strftime("%Y%m%d", time.strptime(time.ctime(os.path.getmtime(self.path))))


class DatasetZarr(DatasetLocalABC):
    """Positions are the groups at the root of a zarr directory store."""

    def get_position_ids(self):
        with os.scandir(self.path) as it:
            return [{"path": self.path, "key": entry.name} for entry in it if entry.is_dir()]


class DatasetDir(DatasetLocalABC):
    """Individual files, possibly nested; `regex` captures the dimensions named, in order, by `capture_order`
    (C channel, W well, T time point, F field of view, Z z-section)."""

    def __init__(self, dpath, regex: str, capture_order: str):
        super().__init__(dpath)
        self.regex = regex
        self.capture_order = capture_order

    def get_position_ids(self, regex: str = None, capture_order: str = None):
        return sort_groups_by_regex(self.path, regex or self.regex, capture_order or self.capture_order)


def scan_directory(path) -> list[str]:
    return [f"{root}/{fname}" for root, _, files in os.walk(path) for fname in files if not f"{root}/{fname}".startswith(".")]


def sort_groups_by_regex(datasets_path, regex: str, capture_order: str, out_dimorder: str = "TCZYX"):
    pattern = re.compile(regex)
    found = []
    for pth in scan_directory(datasets_path):
        m = pattern.match(pth)
        if m:
            found.append((*m.groups(), pth))
    site_keys = [capture_order.index(x) for x in capture_order if x not in out_dimorder]
    dim_keys = [capture_order.index(x) for x in out_dimorder if x in capture_order]
    # stable sorts from the least significant key: Z, C, T, then the site keys in capture order — the last sort wins
    for key in [*dim_keys[::-1], *site_keys]:
        found.sort(key=lambda row, k=key: row[k])
    position_ids = []
    for key, group in groupby(found, key=lambda row: [row[i] for i in site_keys]):
        position_ids.append({"key": "__".join(key), "path": [str(Path(datasets_path) / row[-1]) for row in group]})
    assert len(position_ids), "No files were found."
    return position_ids
