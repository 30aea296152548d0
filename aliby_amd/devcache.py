"""
Device-residency side table.

The reference's step protocol hands NumPy arrays from step to step through
state["data"] (src/aliby/pipe_core.py:188-205,230).  To stay a drop-in while keeping
pixels and labels resident in HBM between tile -> segment -> extract, each step that
produces a NumPy result also registers the device tensor it came from; consumers look
the NumPy object up by identity and skip the host->device upload.  Entries vanish with
the NumPy array (weakref), so the reference's end-of-tp `del entry["pixels"]`
(pipe_core.py:238-242) also frees the device copy.

Dirty check: a registered array is handed out READ-ONLY (`flags.writeable = False`).  An in-place edit between
steps therefore raises NumPy's "assignment destination is read-only" instead of silently leaving a stale device
copy behind; a caller who wants to edit either works on `arr.copy()` (a new object: never a cache hit) or flips
`arr.flags.writeable = True` — and `lookup` treats a registered array found writeable as dirty, drops the entry
and lets the consumer upload the host data again.
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
    try:
        host_array.flags.writeable = False
    except (AttributeError, ValueError):
        pass
    _table[key] = (ref, device_tensor, meta)
    return host_array


def lookup(host_array):
    """Return (device_tensor, meta) if `host_array` is the very object that was registered."""
    hit = _table.get(id(host_array))
    if hit is None:
        return None
    ref, dev, meta = hit
    if ref() is not host_array or getattr(getattr(host_array, "flags", None), "writeable", False):
        _table.pop(id(host_array), None)  # another object at a recycled id, or the caller unlocked it to edit in place
        return None
    return dev, meta


def wait_ready(host_array) -> None:
    """Block until the bytes of a registered host array are valid: producers that download asynchronously register the
    array with `ready=<event>` (the position-batched segmenter does); everything else returns at once."""
    hit = _table.get(id(host_array))
    if hit is not None and hit[0]() is host_array:
        event = hit[2].get("ready")
        if event is not None:
            event.synchronize()
