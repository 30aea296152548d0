"""Microbenchmark of the MFMA convolution unit at the U-Net's level-0/1 shapes (run on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
PLAIN = len(sys.argv) > 2 and sys.argv[2] == "plain"  # the two register-staged shapes only, no MIOpen (phase builds)
SHAPES = [(32, 32, 0, 224), (64, 32, 1, 224), (64, 64, 0, 112), (32, 64, 0, 112), (64, 128, 0, 56), (64, 128, 0, 28)]
for cin, cout, up, H in ([SHAPES[0], SHAPES[2]] if PLAIN else SHAPES):
    ih = H // 2 if up else H
    x = torch.randn(N, ih, ih, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    scale = torch.ones(cin, device="cuda")
    shift = torch.zeros(N, cin, device="cuda")
    bias = torch.zeros(cout, device="cuda")
    res = torch.randn(N, H, H, cout, device="cuda").bfloat16()
    out = torch.empty(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    for with_res in (0, 1):
        def run():
            _lib.check(eng.lib.aliby_nn_conv3x3_bf16(eng.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1,
                                                     _ptr(bias), _ptr(res) if with_res else 0, 0, N, H, H, cin, cout, up, 0, 0, 0, 0, 0, _stream_ptr()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        px = N * H * H
        byts = x.numel() * 2 + out.numel() * 2 + (res.numel() * 2 if with_res else 0)
        fl = 2.0 * 9 * cin * cout * px
        print(f"conv {cin}->{cout} up={up} H={H} N={N} res={with_res}: {ms*1e3:8.1f} us  {byts/ms/1e9:7.2f} TB/s... {byts/ms/1e6/1e3:.2f} GB/ms  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
    if PLAIN:
        continue
    # MIOpen for comparison (bf16 channels_last, no fusion)
    xa = torch.randn(N, cin, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    wa = w.bfloat16().contiguous(memory_format=torch.channels_last)
    torch.backends.cudnn.benchmark = True
    for _ in range(3):
        torch.nn.functional.conv2d(xa, wa, None, padding=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.nn.functional.conv2d(xa, wa, None, padding=1)
    e1.record()
    torch.cuda.synchronize()
    print(f"   MIOpen conv {cin}->{cout} at H={H}: {e0.elapsed_time(e1)/20*1e3:8.1f} us", flush=True)
