"""conv2 + conv3 of a 32-channel block as two launches vs the wave-specialised fused pair: equality (bit for bit) and time."""
import sys

import torch

sys.path.insert(0, ".")
from aliby_amd import _lib  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr  # noqa: E402

eng = FeatureEngine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 288
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 224
g = torch.Generator().manual_seed(3)
x = torch.randn(N, H, W, 32, generator=g).to(torch.bfloat16).cuda()
res = torch.randn(N, H, W, 32, generator=g).to(torch.bfloat16).cuda()
pk = []
for k in range(2):
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).float().cuda()
    p = torch.empty(32 * 32 * 9, dtype=torch.bfloat16, device="cuda")
    _lib.check(eng.lib.aliby_nn_pack_conv3x3_bf16(eng.ctx.handle, _ptr(w), 32, 32, 32, _ptr(p), _stream_ptr()))
    pk.append(p)
sc = [(torch.rand(32, generator=g) + 0.5).float().cuda() for _ in range(2)]
sh = [(torch.randn(N, 32, generator=g) * 0.2).float().cuda() for _ in range(2)]
bi = [(torch.randn(32, generator=g) * 0.1).float().cuda() for _ in range(2)]
mid = torch.empty_like(x)
out2, out1 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
pool2 = torch.full((N, H // 2, W // 2, 32), float("nan"), dtype=torch.bfloat16, device="cuda")
pool1 = torch.full_like(pool2, float("nan"))


def two(pool):
    _lib.check(eng.lib.aliby_nn_conv3x3_bf16(eng.ctx.handle, _ptr(x), _ptr(pk[0]), _ptr(mid), _ptr(sc[0]), _ptr(sh[0]), 32, _ptr(bi[0]), 0, 0,
                                             N, H, W, 32, 32, 0, 0, 0, 0, 0, 0, _stream_ptr()))
    _lib.check(eng.lib.aliby_nn_conv3x3_bf16(eng.ctx.handle, _ptr(mid), _ptr(pk[1]), _ptr(out2), _ptr(sc[1]), _ptr(sh[1]), 32, _ptr(bi[1]), _ptr(res), 0,
                                             N, H, W, 32, 32, 0, 0, 0, 0, 0, _ptr(pool2) if pool else 0, _stream_ptr()))


def one(pool):
    _lib.check(eng.lib.aliby_nn_conv3x3_pair_bf16(eng.ctx.handle, _ptr(x), _ptr(pk[0]), _ptr(pk[1]), _ptr(out1), _ptr(sc[0]), _ptr(sh[0]), 32,
                                                  _ptr(bi[0]), _ptr(sc[1]), _ptr(sh[1]), 32, _ptr(bi[1]), _ptr(res), N, H, W,
                                                  _ptr(pool1) if pool else 0, 0, 0, 0, 0, 0, 0, _stream_ptr()))


for pool in (False, True):
    two(pool)
    one(pool)
    torch.cuda.synchronize()
    print(f"pool={pool}: outputs equal: {torch.equal(out1, out2)}" + (f", pooled equal: {torch.equal(pool1, pool2)}" if pool else ""), flush=True)
    for name, fn in (("two launches", two), ("fused pair", one)):
        for _ in range(3):
            fn(pool)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(pool)
        e1.record()
        torch.cuda.synchronize()
        print(f"  {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
