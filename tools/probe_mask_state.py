import os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
os.environ.setdefault("GNNX_SPARSE_RESIDENT","0")
import torch, bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob
wl = bench.Workload("ba100k", 1024)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
hy = Hyper(num_iters=300, edge_results_only=True)
dn = engine.khop_device(graph, wl.targets, 3)
job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"])
job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + wl.targets, pin=True))
torch.cuda.synchronize()
def t(): 
    job.time_kernel(hy, 0, 10); return round(job.time_kernel(hy, 0, 20)[0]*1e3,1)
print("fresh state: k_mask", t(), t())
job.set_masks_raw_resident(); job.launch(hy); torch.cuda.synchronize()
ws = job.ws
print("after a full run: k_mask", t(), t())
M = job.M
tiny = torch.finfo(torch.float32).tiny
print("M: denormal frac", float(((M.abs() < tiny) & (M != 0)).float().mean()), "min/max", float(M.min()), float(M.max()))
job.set_masks_raw_resident(); torch.cuda.synchronize()
print("masks reloaded (moments / gradients of the run stay): k_mask", t(), t())
