import time, sys
sys.path.insert(0, "/root/repo")
t=time.perf_counter()
import torch; torch.cuda.init(); print("torch+init", round(time.perf_counter()-t,2)); t=time.perf_counter()
from aliby_amd import _lib
lib=_lib.load(); print("lib load", round(time.perf_counter()-t,2)); t=time.perf_counter()
from aliby_amd.extraction.engine import FeatureEngine
eng=FeatureEngine(); h=eng.ctx.handle; print("engine+ctx", round(time.perf_counter()-t,2)); t=time.perf_counter()
x=torch.zeros(4,64,64,dtype=torch.uint16,device="cuda"); tb=eng.object_table(x); torch.cuda.synchronize(); print("first kernel", round(time.perf_counter()-t,2)); t=time.perf_counter()
from aliby_amd.segment.cellpose_hip import CellposeModel
m=CellposeModel(); torch.cuda.synchronize(); print("CellposeModel()", round(time.perf_counter()-t,2)); t=time.perf_counter()
img=torch.randint(0,4000,(2,512,512),dtype=torch.int32).to(torch.uint16).cuda()
r=m.eval(img); torch.cuda.synchronize(); print("first eval", round(time.perf_counter()-t,2)); t=time.perf_counter()
r=m.eval(img); torch.cuda.synchronize(); print("second eval", round(time.perf_counter()-t,2))
