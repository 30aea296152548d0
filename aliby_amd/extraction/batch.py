"""
Batched device-resident extraction: many tiles (FOVs) per launch.

The reference's API is already batched over tiles — `masks` is a list with one label image per tile and
`pixels` is [F,C,Z,Y,X] (extract.py:256-259) — but evaluates them one object at a time.  This module is
the device-side core both `process_tree_masks` and bench.py go through: labels [F,Y,X] and pixels
[F,C,Z,Y,X] stay in HBM, one object table covers every tile, each metric family is one launch.
"""

from __future__ import annotations

from aliby_amd.extraction import families
from aliby_amd.extraction.extract import flatten, kv


def instructions_of(tree) -> list:
    return kv(flatten(tree))


def extract_batch(eng, labels_dev, planes, tree, cp_measure_kwargs=None, multi=False, table=None):
    """Returns (matrix_dev [n_obj, n_cols] float64, column_names, table).

    planes = (device tensor [F,C,Z,Y,X], dtype code) or None when the tree needs no pixels."""
    if table is None:
        table = eng.object_table(labels_dev)
    instructions = instructions_of(tree)
    matrix, blocks = families.evaluate(eng, labels_dev, table, planes, instructions, cp_measure_kwargs or {}, multi=multi)
    names = []
    for inst, (start, keys) in zip(instructions, blocks):
        branch = "/".join(str(x) for x in inst)
        names.extend([f"{branch}/{inst[-1]}"] if keys is None else [f"{branch}/{k}" for k in keys])
    return matrix, names, table
