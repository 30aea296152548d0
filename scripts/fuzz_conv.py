"""Randomised shape sweep of the conv unit against the PyTorch reference on exact integer data (run on the GPU box):
every (cin, cout, upsample) instantiation, image sizes around the packed-launch conditions (W in {28, 56, 112}, N a multiple of
224 / W or not, H a multiple of the tile height or not), residual / pooled output / shared or per-image shift at random."""
import random
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_gpu_conv as T  # noqa: E402
from aliby_amd.extraction.engine import FeatureEngine  # noqa: E402

eng = FeatureEngine()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for case in range(n_cases):
    cin, cout, up = rnd.choice(T.COMBOS)
    W = rnd.choice([28, 56, 112, 112, 56, 28, 30, 60, 224, 34, 2, 66])
    H = rnd.choice([28, 56, 112, 14, 42, 8, 4, 30, 58, 2, 224 if W <= 56 else 16])
    if up:
        H, W = H + (H & 1), W + (W & 1)
    n = rnd.choice([1, 2, 3, 4, 6, 8, 9, 16])
    if n * H * W > 300000:
        n = max(1, 300000 // (H * W))
    g = torch.Generator().manual_seed(case)
    ih, iw = (H // 2, W // 2) if up else (H, W)
    x = torch.randint(-1, 3, (n, ih, iw, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randint(-1, 2, (cout, cin, 3, 3), generator=g) * (torch.rand(cout, cin, 3, 3, generator=g) < 0.15)).float().cuda()
    scale = torch.randint(1, 3, (cin,), generator=g).float().cuda()
    shift = torch.randint(-1, 2, (n, cin), generator=g).float().cuda()
    if rnd.random() < 0.4:
        shift = shift[0].contiguous()
    bias = torch.randint(-2, 3, (cout,), generator=g).float().cuda() if rnd.random() < 0.8 else None
    res_mode = rnd.choice(["none", "same", "up"]) if H % 2 == 0 and W % 2 == 0 else rnd.choice(["none", "same"])
    res = None
    if res_mode == "same":
        res = torch.randint(-3, 4, (n, H, W, cout), generator=g).to(torch.bfloat16).cuda()
    elif res_mode == "up":
        res = torch.randint(-3, 4, (n, H // 2, W // 2, cout), generator=g).to(torch.bfloat16).cuda()
    want_pool = (not up) and H % 2 == 0 and W % 2 == 0 and rnd.random() < 0.4
    pool = torch.full((n, H // 2, W // 2, cout), float("nan"), dtype=torch.bfloat16, device="cuda") if want_pool else None
    out = T._run(eng, x, w, scale, shift, bias, res, res_mode == "up", up, H, W, pool=pool)
    ref = T._reference(x, w, scale, shift, bias, res, res_mode == "up", up)
    ok = torch.equal(out.float(), ref)
    if ok and pool is not None:
        ok = torch.equal(pool.float(), torch.nn.functional.max_pool2d(ref.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))
    if not ok:
        bad += 1
        print("MISMATCH", dict(cin=cin, cout=cout, up=up, n=n, H=H, W=W, res=res_mode, pool=want_pool, shift=tuple(shift.shape)), flush=True)
print(f"{n_cases - bad} of {n_cases} cases exact")
sys.exit(1 if bad else 0)
