"""Does a chain of launches over a SUB-BATCH whose tensors fit the 256 MiB Infinity Cache run faster than the same chain over
the full batch?  Ping-pong (out of launch k is the input of launch k+1) of the 32->32 @224 conv unit and of a plain device
copy, per-tile time against the number of tiles per launch (run on the GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
H, cin, cout = 224, 32, 32
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
_lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
scale = torch.ones(cin, device="cuda")
bias = torch.zeros(cout, device="cuda")
TOTAL = 288
for N in (8, 16, 24, 32, 48, 72, 96, 144, 288):
    bufs = [torch.randn(N, H, H, cin, device="cuda").bfloat16() for _ in range(2)]
    shift = torch.zeros(N, cin, device="cuda")
    reps = TOTAL // N * 4

    def conv(k):
        a, b = bufs[k & 1], bufs[(k + 1) & 1]
        _lib.check(eng.lib.aliby_nn_conv3x3_bf16(eng.ctx.handle, _ptr(a), _ptr(wpk), _ptr(b), _ptr(scale), _ptr(shift), 1, _ptr(bias), 0, 0,
                                                 N, H, H, cin, cout, 0, 0, 0, 0, 0, 0, _stream_ptr()))

    def copy(k):
        bufs[(k + 1) & 1].copy_(bufs[k & 1])

    for name, fn in (("conv32", conv), ("copy", copy)):
        for k in range(4):
            fn(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byts = 2 * bufs[0].numel() * 2
        print(f"{name:7s} N={N:4d} ({bufs[0].numel() * 2 / 2**20:6.1f} MiB per tensor): {ms * 1e3:8.1f} us per launch, {ms * 1e3 / N:6.2f} us per tile, "
              f"{byts / ms / 1e9:6.2f} TB/s", flush=True)
    del bufs
