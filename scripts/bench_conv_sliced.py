"""64 -> 128 at the 28-pixel level as the network launches it: a 64-channel slice of a 256-channel input, a 128-channel slice
of a 256-channel output accumulated in place (RES = OUT), against the same launch on dense tensors."""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 288
for H in (28, 56):
    ctot = 256 if H == 28 else 128
    cin, cout = 64, 128
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), cout, cin, cin, _ptr(wpk), _stream_ptr()))
    scale, shift, bias = torch.ones(cin, device="cuda"), torch.zeros(N, cin, device="cuda"), torch.zeros(cout, device="cuda")
    for label, in_c, out_c in (("dense", cin, cout), ("sliced", ctot, max(ctot, cout))):
        x = torch.randn(N, H, H, in_c, device="cuda").bfloat16()
        out = torch.zeros(N, H, H, out_c, device="cuda", dtype=torch.bfloat16)
        for inplace in (0, 1):
            res = out if inplace else torch.randn(N, H, H, out_c, device="cuda").bfloat16()

            def run(k=0):
                _lib.check(eng.lib.aliby_nn_conv3x3_bf16(
                    eng.ctx.handle, _ptr(x), _ptr(wpk), _ptr(out), _ptr(scale), _ptr(shift), 1, _ptr(bias), _ptr(res), 0, N, H, H, cin, cout, 0,
                    in_c, (k % (in_c // cin)) * cin, out_c, 0, 0, _stream_ptr()))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(40):
                run(k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            fl = 2.0 * 9 * cin * cout * N * H * H
            print(f"H={H} {label:6s} res={'OUT (in place)' if inplace else 'separate'}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
