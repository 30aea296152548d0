"""Phase timing of the conv unit's workgroup 0 (shader-clock stamps), for the latency analysis in DESIGN.md."""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = 96
for cin, cout, up, H in [(32, 32, 0, 224)]:
    x = torch.randn(N, H, H, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    scale = torch.ones(cin, device="cuda"); shift = torch.zeros(N, cin, device="cuda"); bias = torch.zeros(cout, device="cuda")
    res = torch.randn(N, H, H, cout, device="cuda").bfloat16()
    out = torch.empty(N, H, H, cout, device="cuda", dtype=torch.bfloat16)
    for with_res in (0, 1):
        stamps = torch.zeros(16, 8, dtype=torch.int64, device="cuda")
        for rep in range(3):
            _lib.check(eng.lib.aliby_debug_conv_trace(eng.ctx.handle, _ptr(stamps) if rep == 2 else 0))
            _lib.check(eng.lib.aliby_nn_conv3x3_bf16(eng.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1,
                                                     _ptr(bias), _ptr(res) if with_res else 0, 0, N, H, H, cin, cout, up, 0, 0, 0, 0, 0, _stream_ptr()))
        torch.cuda.synchronize()
        _lib.check(eng.lib.aliby_debug_conv_trace(eng.ctx.handle, 0))
        s = stamps.cpu().numpy()
        d = s[:, 1:7] - s[:, 0:6]
        print(f"conv {cin}->{cout} res={with_res}: phase ticks (s_memtime, 100 MHz?) per tile: loads-issue, barrier1, prologue(incl load wait), barrier2, mfma, epilogue")
        for t in range(1, 12):
            print("   tile", t, d[t].tolist(), " tile total", int(s[t, 6] - s[t, 0]), " gap to next", int(s[t + 1, 0] - s[t, 6]))
