"""
Device-residency side table.

The reference's step protocol hands NumPy arrays from step to step through
state["data"] (src/aliby/pipe_core.py:188-205,230).  To stay a drop-in while keeping
pixels and labels resident in HBM between tile -> segment -> extract, each step that
produces a NumPy result also registers the device tensor it came from; consumers look
the NumPy object up by identity and skip the host->device upload.  Entries vanish with
the NumPy array (weakref), so the reference's end-of-tp `del entry["pixels"]`
(pipe_core.py:238-242) also frees the device copy.
"""

from __future__ import annotations

import weakref

_table: dict[int, tuple] = {}


def attach(host_array, device_tensor, **meta):
    key = id(host_array)

    def _drop(k=key):
        _table.pop(k, None)

    try:
        ref = weakref.ref(host_array)
        weakref.finalize(host_array, _drop)
    except TypeError:  # object does not support weak references
        return host_array
    _table[key] = (ref, device_tensor, meta)
    return host_array


def lookup(host_array):
    """Return (device_tensor, meta) if `host_array` is the very object that was registered."""
    hit = _table.get(id(host_array))
    if hit is None:
        return None
    ref, dev, meta = hit
    if ref() is not host_array:
        _table.pop(id(host_array), None)
        return None
    return dev, meta
