import sys, time, torch
sys.path.insert(0, "/root/repo")
import warnings; warnings.simplefilter("ignore")
from aliby_amd.segment.cellpose_hip import CellposeModel
torch.backends.cudnn.benchmark = (sys.argv[1] == "1")
model = CellposeModel(net_dtype="bfloat16", flows_override=None, batch_size=96)
x = torch.randn(96, 2, 224, 224, device="cuda")
for _ in range(3): model.fused(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(6): model.fused(x)
b.record(); torch.cuda.synchronize()
print("benchmark", sys.argv[1], "ms per 96-tile batch", a.elapsed_time(b)/6)
