#!/usr/bin/env python
"""bench.py — explained nodes/sec of the MI355X-native GNNExplainer engine.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): syn1 BA-House, explain ALL 400 house-motif nodes (300..699) as ONE
batched job, 300 mask-optimisation iterations, 3-hop sub-graphs, Adam lr 0.1, fp32.  The graph and the
trained GCN come from tests/golden/syn1_ckpt.npz (minted by the reference's own train.py); initial masks
follow the seed protocol torch.manual_seed(1000 + node).
A "step" = one pass of the hot path over the whole batch: reset the edge masks to M0 (device copy) and run
the 300 iterations.  Inputs are resident in HBM before the timed region.
N > 1: weak scaling — every rank runs its own copy of the batch (targets are independent; no collective on the
data path), value = N * 400 * K / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12       # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12


class Workload:
    """Targets of one rank: the full graph (CSR), the frozen encoder and the k-hop neighbour list of every target."""

    def __init__(self, name, rank=0, num_targets=4096):
        import helpers
        from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
        ck = helpers.load_ckpt("syn4" if name == "syn4" else "syn5" if name == "syn5" else "syn1")
        self.ck = ck
        if name in ("syn4", "syn5"):
            # BASELINE.json configs[2] for the record: Tree-Cycle / Tree-Grid, all motif nodes (ids >= 511)
            self.idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
            self.feat, self.label, self.pred = ck["feat"], ck["label"], ck["pred"]
            targets = range(511, ck["num_nodes"])
            self.desc = f"{name}: all {ck['num_nodes'] - 511} motif nodes (511..{ck['num_nodes'] - 1}) as one batch per GPU"
        elif name == "syn1":
            self.idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
            self.feat, self.label, self.pred = ck["feat"], ck["label"], ck["pred"]
            targets = range(300, 700)
            self.desc = "syn1: all 400 house-motif nodes (300..699) as one batch per GPU"
        elif name == "ba100k":
            # BASELINE.json configs[4]: BA-House scaled to 100k nodes (42857 BA + 11428 houses, 1 % random edges),
            # encoder = the syn1 checkpoint (same D/H/C), targets = a fixed random sample of motif nodes per rank
            from gnn_model_explainer_amd.utils import synthetic
            n, edges, self.label = synthetic.ba_house(42857, 11428, seed=0)
            csr = synthetic.csr_from_edges(n, edges)
            self.feat = np.ones((n, 10), np.float32)
            self.pred = synthetic.sparse_gcn_predict(csr, self.feat, ck["sd"])
            self.idx = KHopIndex(csr, 3)
            rng = np.random.default_rng(1234 + rank)
            targets = np.sort(rng.choice(np.arange(42857, n), num_targets, replace=False))
            self.desc = f"BA-House x100k (99997 nodes): {num_targets} sampled motif nodes per GPU (seed 1234+rank)"
        else:
            raise SystemExit("unknown workload " + name)
        self.targets = [int(t) for t in targets]

    def prepare(self):
        """Per-batch host work: k-hop neighbour lists (sparse products) and the seeded initial masks."""
        import helpers
        self.nbs = self.idx.neighbors_batch(self.targets)
        self.rows = [int(np.searchsorted(nb, t)) for t, nb in zip(self.targets, self.nbs)]
        self.masks = [helpers.seeded_mask0(t, len(nb)).numpy() for t, nb in zip(self.targets, self.nbs)]

    def dense_subgraph(self, k):
        from gnn_model_explainer_amd.engine import Subgraph
        nb, t = self.nbs[k], self.targets[k]
        return Subgraph(self.idx.sub_adjacency(nb), self.feat[nb], int(self.label[t]), self.rows[k],
                        np.argmax(self.pred[nb], 1), self.masks[k])


def cpu_baseline(wl, iters, budget_s=20.0):
    """Oracle ("port": torch-autograd restatement, bit-identical to the reference on CPU) timed on this
    host's cores over a bounded, size-stratified sample of the same targets.  Wall-clock bounded: the epoch
    loop is stepped in chunks and the last target may be counted fractionally."""
    from oracle import reference_restatement as rr
    ck = wl.ck
    order = np.argsort([len(nb) for nb in wl.nbs])
    sample = [wl.dense_subgraph(int(k)) for k in order[np.linspace(0, len(order) - 1, 24).astype(int)]]
    subs = wl.targets
    sd = {k: torch.tensor(v) for k, v in ck["sd"].items()}
    # the reference is dispatch-bound (~700 tiny aten ops / epoch): more than a few threads only adds OpenMP
    # fork/join cost, and on a many-core GPU host os.cpu_count() threads is pathologically slow
    cores = min(8, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    done, t0, ns = 0.0, time.time(), []
    for s in sample:
        o = rr.MaskOptimOracle(torch.tensor(s.adj), torch.tensor(s.feat), sd, s.gt_label, s.pred_label, s.target_row,
                               mask0=torch.tensor(s.mask0))
        ep = 0
        while ep < iters and time.time() - t0 < budget_s:
            o.run(min(25, iters - ep))
            ep += min(25, iters - ep)
        done += ep / iters
        ns.append(s.adj.shape[0])
        if time.time() - t0 >= budget_s:
            break
    dt = time.time() - t0
    return {"value": done / dt, "unit": "explained nodes/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{done:.2f} of {len(subs)} targets (size-stratified, n={ns}), {iters} iters each, "
                      f"oracle/reference_restatement.py (bit-identical to the reference) on torch {torch.__version__} CPU, "
                      f"{cores} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--workload", default="syn1", choices=["syn1", "ba100k", "syn4", "syn5"])
    ap.add_argument("--targets", type=int, default=4096, help="ba100k: sampled motif targets per GPU")
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="streaming kernels for every target (no on-chip-resident path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    from gnn_model_explainer_amd.engine import device_graph
    wl = Workload(args.workload, rank, args.targets)
    ck, desc, subs = wl.ck, wl.desc, wl.targets
    graph = device_graph(wl.idx.csr, wl.feat, wl.pred)          # the input graph lives in HBM (uploaded once)
    torch.cuda.synchronize()
    t_prep = time.perf_counter()
    wl.prepare()                                                # host: k-hop lists + seeded masks
    t_prep = time.perf_counter() - t_prep
    log(f"workload built: {len(subs)} targets")
    t_pack = time.perf_counter()
    job = MaskOptimJob.from_csr(graph, wl.nbs, wl.rows, wl.label[np.asarray(wl.targets)], ck["sd"])
    hy = Hyper(num_iters=args.iters, use_graph=not args.no_graph, use_resident=not args.no_resident)
    job.set_masks(wl.masks)
    torch.cuda.synchronize()
    t_pack = time.perf_counter() - t_pack          # neighbour lists H2D + device-side packing + M0 H2D
    M0 = job.M.clone()

    def step():
        job.M.copy_(M0)
        job.launch(hy)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    log(f"timed region done: {dt:.3f} s")
    n_targets = len(subs) * world
    value = n_targets * args.steps / dt
    t_fetch = time.perf_counter()
    job.fetch(hy)                                   # D2H of Abar, M, feature masks + unpack to per-target arrays
    t_fetch = time.perf_counter() - t_fetch

    out = None
    if rank == 0:
        # roofline of the dominant kernel, measured live with HIP events on the launch stream.
        # Which kernel that is depends on the routing: the resident kernels run ALL iterations in one launch (their
        # targets never touch HBM inside the loop, SURVEY.md §8d -> MFMA bound on the algorithmic flops); the streaming
        # remainder runs 5 launches per iteration (HBM bound on the algorithmic bytes).
        sum_n2 = job.sum_n2
        kagg = job.D + 2 * job.H
        route = job.route() if not args.no_resident else np.zeros(len(subs), np.int32)
        n2 = job.n.astype(np.float64) ** 2
        n_stream = int((route == 0).sum())
        launches = {}
        if n_stream:
            reps = 50 if n2[route == 0].sum() < 1e8 else 5
            names = ["k_mask<true,true>", "k_conv<FWD1>", "k_conv<FWD2>", "k_node_head", "k_conv<BWD1>"]
            per = [job.time_kernel(hy, k, reps) for k in (0, 1, 2, 3, 4)]
            launches["streaming"] = {"targets": n_stream, "ms_per_iter": sum(x[0] for x in per),
                                     "ms_total": sum(x[0] for x in per) * args.iters,
                                     "avg_launch_us": {nm: x[0] * 1e3 for nm, x in zip(names, per)}}
        else:
            per = []
        # resident launches: timed IN SITU (HIP events on the side streams they run on) during five more full steps, so
        # the durations are those of the timed region's launches (the rocprofv3 kernel trace of this command shows the same)
        rts = []
        for _ in range(5):
            step()
            torch.cuda.synchronize()
            rts.append(job.resident_times())
        rt = [float(np.mean(x)) for x in zip(*rts)]
        sel = lambda m: float(n2[m].sum())
        # the longest resident launch: route 1 = k_resident<1>, 4 / 5 / 6 = k_sparse_resident with 1024 / 256 / 64 threads,
        # 7 = k_sparse_large
        res_names = {1: "k_resident<1>", 4: "k_sparse_resident<.., 1024>", 5: "k_sparse_resident<.., 256>", 6: "k_sparse_resident<.., 64>",
                     7: "k_sparse_large", 8: "k_sparse_resident<.., 512>"}
        res_ms = {1: rt[0], 4: rt[3], 5: rt[4], 6: rt[5], 7: rt[6], 8: rt[7]}
        mixed = bool((route == 6).any() and (route == 8).any() and not rt[5] and rt[7])   # one launch for both groups
        if mixed:
            res_names[8] = "k_sparse_resident_mixed (512-thread targets + six single-tile targets per workgroup)"
        for rv, ms_v in res_ms.items():
            if ms_v:
                cnt = int((route == rv).sum()) + (int((route == 6).sum()) if (mixed and rv == 8) else 0)
                launches[res_names[rv]] = {"targets": cnt, "ms_total": ms_v}
        # concurrent launches of (nearly) the same length: report the one that carries the most algorithmic work
        longest = max(res_ms.values())
        top = max((rv for rv in res_ms if res_ms[rv] >= 0.8 * longest and res_ms[rv] > 0), key=lambda rv: sel(route == rv),
                  default=max(res_ms, key=res_ms.get))
        ms_sp = res_ms[top]
        ms_r1 = 0.0
        top_sel = ((route == 8) | (route == 6)) if (mixed and top == 8) else (route == top)
        by_sp, fl_sp = 28.0 * sel(top_sel) * args.iters, 6.0 * sel(top_sel) * kagg * args.iters
        by_r1 = fl_r1 = 0.0
        stream_total = launches.get("streaming", {}).get("ms_total", 0.0)
        if stream_total >= max(ms_sp, ms_r1):
            k = int(np.argmax([x[0] for x in per]))
            roof = {"kernel": names[k] + (" (fused sigmoid-mask + regulariser + Adam + G-tile MFMA)" if k == 0 else " (masked-adjacency contraction)"),
                    "bound": "hbm", "achieved": per[k][1] / (per[k][0] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": per[k][1] / (per[k][0] * 1e-3) / HBM_PEAK, "traffic": None, "avg_launch_us": per[k][0] * 1e3}
        else:
            sparse = top >= 4
            ms, by, fl = ms_sp, by_sp, fl_sp
            roof = {"kernel": (res_names[top] + " (edge-sparse on-chip-resident optimisation, one workgroup per target, "
                               "all iterations in one launch)") if sparse else "k_resident<1> (dense single-tile on-chip-resident optimisation)",
                    "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": fl / (ms * 1e-3) / MFMA_F32_PEAK, "traffic": None, "avg_launch_us": ms * 1e3,
                    "hbm_equivalent": {"achieved": by / (ms * 1e-3) / 1e9, "unit": "GB/s", "frac": by / (ms * 1e-3) / HBM_PEAK},
                    "note": "algorithmic work of the reference's dense formulation for the launch's targets (SURVEY.md §8d: "
                            "6 n^2 (D+2H) flop and 28 n^2 B per iteration); the kernel keeps all state on chip (HBM is touched only "
                            "before and after the loop)" + (" and skips the mask entries off the edges, which never reach an output" if sparse else "")}
        roof["launches"] = launches
        roof["whole_job"] = {
            "alg_flops_per_step": 6.0 * sum_n2 * kagg * args.iters,
            "mfma_f32_frac": 6.0 * sum_n2 * kagg * args.iters * args.steps / dt / MFMA_F32_PEAK,
            "alg_bytes_per_step": 28.0 * sum_n2 * args.iters,
            "hbm_frac": 28.0 * sum_n2 * args.iters * args.steps / dt / HBM_PEAK,
            "wall_ms_per_iter": dt / args.steps / args.iters * 1e3}
        out = {"metric": "explained nodes/sec (300 mask-opt iters, k-hop subgraph)", "value": value,
               "unit": "explained nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic (BA-House graphs; GCN trained by the reference train.py on syn1, fixture tests/golden/syn1_ckpt.npz)",
               "config": {"workload": desc + f", 3-hop sub-graphs, {args.iters} iters, Adam lr 0.1",
                          "targets_per_gpu": len(subs), "sum_n2": sum_n2,
                          "launch": "plain" if args.no_graph else "hipGraph", "resident_path": not args.no_resident,
                          "routing": {"streaming": int((route == 0).sum()), "dense_resident": int(((route >= 1) & (route <= 3)).sum()),
                                      "sparse_resident": int((route >= 4).sum())},
                          "parallelism": f"target-sharded x{world}"},
               "roofline": roof}
        log("kernel timings done")
        step_s = dt / args.steps
        out["pcie_inclusive"] = {"value": len(subs) / (t_prep + t_pack + step_s + t_fetch), "unit": "explained nodes/s",
                                 "host_khop_and_mask_init_ms": t_prep * 1e3, "plan_pack_h2d_ms": t_pack * 1e3,
                                 "gpu_ms": step_s * 1e3, "d2h_unpack_ms": t_fetch * 1e3,
                                 "note": "one batch end to end on rank 0: host k-hop lists + seeded mask init, plan + "
                                         "neighbour-list H2D + device-side packing (gnnx_pack_csr) + M0 H2D, optimisation, "
                                         "D2H + unpack; the graph itself is resident; never used as `value`"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.iters)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
