"""Microbenchmark of the deep-level K-loop convolution (csrc/nn_conv_deep.hip) at the network's shapes (run on the GPU box).
usage: python scripts/bench_conv_deep.py [N=288] [m16]      (m16: the v_mfma_f32_16x16x32_bf16 form)"""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 288
M16 = len(sys.argv) > 2 and sys.argv[2] == "m16"
tot_ms = tot_fl = 0.0
# (cin, cout, upsampled input, H, how many times the network runs this shape per forward)
for cin, cout, up, H, count in [(64, 128, 0, 56, 1), (128, 128, 0, 56, 6), (256, 128, 1, 56, 1), (128, 256, 0, 28, 1), (256, 256, 0, 28, 7)]:
    ih = H // 2 if up else H
    x = torch.randn(N, ih, ih, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    if M16:
        _lib.check(eng.lib.aliby_nn_pack_conv3x3_deep16_bf16(eng.ctx.handle, _ptr(w), cout, cin, _ptr(wpk), _stream_ptr()))
    else:
        _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    scale = torch.ones(cin, device="cuda")
    shift = torch.zeros(N, cin, device="cuda")
    bias = torch.zeros(cout, device="cuda")
    res = torch.randn(N, H, H, cout, device="cuda").bfloat16()
    out = torch.empty(N, H, H, cout, device="cuda", dtype=torch.bfloat16)

    def run():
        _lib.check((eng.lib.aliby_nn_conv3x3_deep16_bf16 if M16 else eng.lib.aliby_nn_conv3x3_deep_bf16)(eng.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1, _ptr(bias),
                                                      _ptr(res), 0, N, H, H, cin, cout, up, _stream_ptr()))

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * 9 * cin * cout * N * H * H
    byts = 2 * (x.numel() + out.numel() + res.numel())
    tiles = -(-(N * (H + 1) * (H + 2)) // 224) * (cout // 128)
    print(f"deep conv {cin:3d}->{cout:3d} up={up} @{H}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e6:7.1f} GB/s alg  "
          f"{tiles} tiles = {tiles / 512:.2f} rounds of 512 workgroups", flush=True)
    tot_ms += ms * count
    tot_fl += fl * count
print(f"per forward of {N} tiles: {tot_ms:.3f} ms in the deep convolutions = {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
