"""
Host wrapper of the HIP Cellpose dynamics (aliby_amd/csrc/dynamics.hip): network outputs -> labels.

Plays the role of cellpose.dynamics.compute_masks inside `model.eval` (reference call site
src/aliby/segment/dispatch.py:208-215).  torch only provides the device buffers.
"""

from __future__ import annotations

import numpy as np
import torch

from aliby_amd import _lib
from aliby_amd.extraction.engine import _ptr, _stream_ptr

_workspaces: dict = {}


def _mark(label):
    from aliby_amd import trace  # (diagnostic marks of the launch thread: ALIBY_RUNNER_TRACE)

    trace.mark(label)


def _workspace(lib, F, Y, X, device):
    need = int(lib.aliby_masks_workspace_bytes(F, Y, X))
    key = (str(device),)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws, need


def masks_from_flows(eng, dP, cellprob, niter=200, cellprob_threshold=0.0, flow_threshold=0.4, min_size=15,
                     max_size_fraction=0.4, return_endpoints=False):
    """dP float32 [F,2,Y,X], cellprob float32 [F,Y,X] (device) -> (labels uint16 [F,Y,X] device, counts[F])."""
    assert dP.dtype == torch.float32 and cellprob.dtype == torch.float32
    dP = dP.contiguous()
    cellprob = cellprob.contiguous()
    F, two, Y, X = dP.shape
    assert two == 2 and tuple(cellprob.shape) == (F, Y, X)
    labels = torch.empty((F, Y, X), dtype=torch.uint16, device=dP.device)  # (cleared by the library)
    n = np.zeros(max(F, 1), np.int32)
    ws, need = _workspace(eng.lib, F, Y, X, dP.device)
    pf = torch.zeros((F, 2, Y, X), dtype=torch.float32, device=dP.device) if return_endpoints else None
    _mark("dynamics:call")
    from aliby_amd import trace

    trace.about_to_block()  # (the call below waits for everything queued so far: ~100 ms for a 64-position batch)
    with eng.timed("dynamics"):
        _lib.check(
            eng.lib.aliby_masks_from_flows(
                eng.ctx.handle, _ptr(dP), _ptr(cellprob), F, Y, X, int(niter), float(cellprob_threshold),
                float(flow_threshold if flow_threshold is not None else 0.0), int(min_size), float(max_size_fraction),
                _ptr(ws), need, _ptr(labels), _ptr(n), _ptr(pf) if pf is not None else 0, _stream_ptr(),
            )
        )
    _mark("dynamics:returned")
    if return_endpoints:
        return labels, n[:F], pf
    return labels, n[:F]
