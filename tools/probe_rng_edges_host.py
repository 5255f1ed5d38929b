#!/usr/bin/env python
"""Measurement tool (HOST ONLY - runs in the build container too): the seeded draw of a BA-House x100k target set standalone - the full stream
(gnnx_host_draw_masks_sliced) against the values on the edges only (gnnx_host_draw_edge_masks, block-granular form), k-hop sets and edge lists
from the host index, bit-compared on a sample of targets.  python tools/probe_rng_edges_host.py [targets = 2048]"""
import sys, time, numpy as np, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from gnn_model_explainer_amd import engine
NT=int(sys.argv[1]) if len(sys.argv)>1 else 2048
t0=time.time(); wl=bench.Workload("ba100k", NT); print('workload', round(time.time()-t0,1),'s', flush=True)
t0=time.time(); nbs=wl.idx.neighbors_batch(wl.targets); print('khop host', round(time.time()-t0,1),'s', flush=True)
sizes=np.asarray([len(nb) for nb in nbs],np.int32)
csr=wl.idx.csr
rcs=[];eoff=[0]
t0=time.time()
for nb in nbs:
    sub=csr[nb][:,nb].tocoo()
    m=sub.row<sub.col
    r=sub.row[m].astype(np.int32); c=sub.col[m].astype(np.int32)
    o=np.lexsort((c,r)); rcs.append(np.stack([r[o],c[o]],1)); eoff.append(eoff[-1]+len(r))
rc=np.concatenate(rcs); eoff=np.asarray(eoff,np.int64)
print('edges', round(time.time()-t0,1),'s', len(sizes),'targets', '%.3g'%float((sizes.astype(np.int64)**2).sum()),'normals', len(rc),'edges, max n', sizes.max(), flush=True)
total=int((sizes.astype(np.int64)**2).sum())
seeds=1000+wl.targets
full=torch.empty(total,dtype=torch.float32)
out=torch.empty(len(rc),2,dtype=torch.float32)
for th in (8,16):
    ts=[]
    for _ in range(3):
        t0=time.perf_counter(); engine.init_edge_masks_raw(sizes,seeds=seeds,threads=th,out=full); ts.append(time.perf_counter()-t0)
    print('full stream threads %d: %.1f ms'%(th,min(ts)*1e3), flush=True)
    for sl in (1<<15,1<<17):
        ts=[]
        for _ in range(3):
            t0=time.perf_counter(); engine.init_edge_masks_on_edges(sizes,seeds,eoff,rc,threads=th,out=out,slice_values=sl); ts.append(time.perf_counter()-t0)
        print('edges only  threads %d slice %d: %.1f ms'%(th,sl,min(ts)*1e3), flush=True)
off=np.concatenate([[0],np.cumsum(sizes.astype(np.int64)**2)])
# bit check on a sample of targets
idx=np.random.default_rng(0).choice(len(sizes),64,replace=False)
for k in idx:
    a,b=eoff[k],eoff[k+1]; r=rc[a:b,0].astype(np.int64); c=rc[a:b,1].astype(np.int64); n=int(sizes[k])
    assert torch.equal(out[a:b,0], full[off[k]+r*n+c]) and torch.equal(out[a:b,1], full[off[k]+c*n+r]), k
k=int(np.argmax(sizes)); a,b=eoff[k],eoff[k+1]; r=rc[a:b,0].astype(np.int64); c=rc[a:b,1].astype(np.int64); n=int(sizes[k])
assert torch.equal(out[a:b,0], full[off[k]+r*n+c]) and torch.equal(out[a:b,1], full[off[k]+c*n+r])
print('bit-identical to the full stream on 65 targets incl. the largest (n = %d)'%n)
