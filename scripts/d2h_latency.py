"""Round-trip latency of small device->host copies: pageable vs pinned destination (decides how the per-step scalars travel)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from aliby_amd import _lib
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr
eng = FeatureEngine()
d = torch.arange(64, dtype=torch.int32, device="cuda")
big = torch.zeros(8192 * 8, dtype=torch.int32, device="cuda")
for name, src in (("256 B", d), ("256 KB", big)):
    host = np.empty(src.numel(), np.int32)
    pin = torch.empty(src.numel(), dtype=torch.int32, pin_memory=True)
    for label, fn in (
        ("C ABI aliby_memcpy_d2h -> pageable numpy", lambda: _lib.check(eng.lib.aliby_memcpy_d2h(eng.ctx.handle, host.ctypes.data, _ptr(src), src.numel() * 4, _stream_ptr()))),
        ("torch copy_ -> pinned + synchronize", lambda: (pin.copy_(src, non_blocking=True), torch.cuda.current_stream().synchronize())),
        ("torch .cpu()", lambda: src.cpu()),
    ):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(300): fn()
        print(f"{name:7s} {label:45s} {(time.perf_counter() - t) / 300 * 1e6:8.1f} us per copy")
