"""Fused network at very small tile sizes against the fp32 module (a probe for shape assumptions of the launch forms)."""
import sys
import torch
sys.path.insert(0, ".")
from aliby_amd.extraction.engine import FeatureEngine
from aliby_amd.segment.fused_unet import FusedUNet
from aliby_amd.segment.unet import build_network

eng = FeatureEngine()
net = build_network(seed=6, device="cuda")
g = torch.Generator(device="cpu").manual_seed(2)
fused = FusedUNet(net, eng)
for n, h, w in [(2, 64, 64), (3, 48, 80), (2, 32, 32), (1, 32, 48), (5, 48, 32), (2, 16, 16), (1, 16, 32), (7, 24, 40)]:
    x = torch.randn(n, 2, h, w, generator=g).cuda().contiguous()
    with torch.no_grad():
        y_ref, _ = net(x)
    y, _ = fused(x)
    torch.cuda.synchronize()
    print(n, h, w, "rel l2", float((y - y_ref).norm() / y_ref.norm()), flush=True)
