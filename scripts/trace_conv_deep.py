"""Phase stamps of the deep-level convolution kernel (wave 0 of workgroup 0), in shader cycles per phase."""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = 288
for cin, cout, up, H in [(64, 128, 0, 56), (256, 256, 0, 28)]:
    ih = H // 2 if up else H
    x = torch.randn(N, ih, ih, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    scale, shift, bias = torch.ones(cin, device="cuda"), torch.zeros(N, cin, device="cuda"), torch.zeros(cout, device="cuda")
    res = torch.randn(N, H, H, cout, device="cuda").bfloat16()
    out = torch.empty(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    stamps = torch.zeros((8, 32), dtype=torch.int64, device="cuda")

    def run():
        _lib.check(eng.lib.aliby_nn_conv3x3_deep_bf16(eng.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1, _ptr(bias),
                                                      _ptr(res), 0, N, H, H, cin, cout, up, _stream_ptr()))

    for _ in range(3):
        run()
    _lib.check(eng.lib.aliby_debug_conv_deep_trace(eng.ctx.handle, _ptr(stamps)))
    run()
    torch.cuda.synchronize()
    _lib.check(eng.lib.aliby_debug_conv_deep_trace(eng.ctx.handle, 0))
    st = stamps.cpu().numpy()
    S = cin // 64
    print(f"{cin}->{cout} @{H}: cycles per phase, tiles 1..3 of workgroup 0 (s_memtime ticks = 100 MHz reference? see guide: shader cycles)")
    for t in range(1, 4):
        row = st[t]
        d = lambda a, b: int(row[b] - row[a])
        parts = [f"setup {d(0, 1)}"]
        prev = 1
        for s in range(S):
            parts.append(f"| s{s}: wait1 {d(prev, 2 + 4 * s)} act {d(2 + 4 * s, 3 + 4 * s)} wait2 {d(3 + 4 * s, 4 + 4 * s)} mfma {d(4 + 4 * s, 5 + 4 * s)}")
            prev = 5 + 4 * s
        parts.append(f"| epilogue {d(prev, 2 + 4 * S)} | tile {d(0, 2 + 4 * S)}  next-tile gap {int(st[t + 1][0] - row[2 + 4 * S])}")
        print("  ", " ".join(parts))
