#!/usr/bin/env python
"""bench.py — explained nodes/sec of the MI355X-native GNNExplainer engine.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N = 1 (BASELINE.json configs[1], the configuration the metric is quoted on): syn1 BA-House, ALL 400 house-motif nodes
(300..699) as ONE batched job, 300 mask-optimisation iterations, 3-hop sub-graphs, Adam lr 0.1, fp32.  The graph and the
trained GCN come from tests/golden/syn1_ckpt.npz (minted by the reference's own train.py); initial masks follow the seed
protocol torch.manual_seed(1000 + node).  A "step" = one batch through the WHOLE hot path with only the graph resident in HBM
(SURVEY.md section 8(d)): device k-hop, plan, device-side packing, routing, seeded host RNG, H2D + scatter, the 300 iterations,
gather + D2H of the masks; consecutive batches overlap their stages (pipeline.BatchPipeline) and `value` = targets x K / wall time
from the first submission to the last result.  The optimisation alone on resident inputs (what rounds 1-2 reported) is the
secondary `loop_only`; `--workload config4` times BASELINE configs[3] (graph mode, 4337 molecule-like graphs) the same way.

N > 1 (BASELINE.json configs[4], the north-star scaling curve): BA-House scaled to 100k nodes, ONE fixed set of 16384
motif targets (seed-fixed) split over the ranks by longest-processing-time-first on parallel.target_cost(n); every
rank optimises its shard as one batched job and the masks are gathered as edge entries through RCCL INSIDE the timed
region.  Strong scaling: the total work is fixed, value = 16384 * K / max-over-ranks time.

Every run checks parity in the same process: the GPU masks of all targets - the ones the TIMED end-to-end batches produced -
against the reference's own outputs (tests/golden/*_full_explain.npz, produced by running /root/reference) and - at N = 1 - against
the CPU oracle on the CPU-baseline sample.  The rule is helpers.explained_outcome (round 6: no percentage): every CALM target (conditioning over
the horizon <= 2e-6, measured on the CPU alone) lies within 1e-5 of the reference's output or is on the decision suite's committed list
(tests/golden/<name>_ties.json: the first differing decision is a tie of the reference, or the drift is inside the accumulated round-off bound)
and within 5e-3 (the largest branch jump seen on the CPU); the run FAILS otherwise.  The targets that are not calm (CPU-vs-CPU up to 0.97 on
Tree-Grid) cannot be gated at the full horizon by any implementation; what pins them - every target, iterations 0..300 - is the decision suite
against the reference's own optimiser state (tests/test_decision_parity.py), not this gate.
"""
import argparse
import gc
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnn_model_explainer_amd as _pkg

# BEFORE torch is imported and before the first HIP call: eight hardware queues, small copies as blit kernels, torch's CPU pool sized to the
# container's quota, the process confined to one NUMA node - an explicit call since round 6 (importing the package changes nothing any more).
_pkg.tune_process()

import numpy as np
import torch


HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12     # flop/s, dense f32 MFMA
LDS_PEAK_PER_CU = 128 * 2.4e9   # B/s: ds_read_b32 = 128 B/clk/CU at ~2.4 GHz (MI355X_MICROARCH.md, LDS table)
NUM_CUS = 256
PARITY_TOL = 1e-5
def cpu_quota_cores():
    """cores the cgroup of this process may use (cpu.max = quota period), or None: the GPU box's container gets 16 of the host's 256 CPUs"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        return None


def rng_threads_for(total_values, world=1):
    """host threads drawing the seeded initial masks (targets are independent under the seed protocol): the pipeline's own setting for
    small and large batches alike (beyond ~32 threads the hand-off costs more than it saves: tools/probe_rng.py, tools/probe_rng_big.py)"""
    from gnn_model_explainer_amd import engine
    # (more threads do not help the big batches either: 1e9 normals take 94-140 ms on 32 threads of the GPU box's 256-CPU host and 160-230 ms on
    #  96-128 - the draw is bound by the memory system, tools/probe_rng_big.py; N ranks on one host share its cores)
    # (... of the cores the process may USE: the GPU box's container has a quota of 16 of its host's 256 CPUs, and eight ranks of sixteen threads
    #  each on sixteen cores only take turns)
    cores = (os.cpu_count() or 4) // 2
    quota = engine.cpu_quota_cores()
    if quota:
        cores = min(cores, int(2 * quota))
    return max(2, min(engine.default_rng_threads(big=total_values > 2e7), cores // max(1, world)))


WELL = 2e-6                  # CPU-vs-CPU deviation (reference vs closed-form oracle) up to which a target is well conditioned


class Workload:
    """The full graph (CSR), the frozen encoder and the target ids of the job."""

    def __init__(self, name, num_targets=16384):
        import helpers
        from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
        self.name = name
        self.golden = None
        ck = helpers.load_ckpt("syn4" if name == "syn4" else "syn5" if name == "syn5" else "syn1")
        self.ck = ck
        if name in ("syn1", "syn4", "syn5"):
            # configs[1] (syn1) / configs[2] for the record (syn4 Tree-Cycle, syn5 Tree-Grid): all motif nodes
            self.idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
            self.feat, self.label, self.pred = ck["feat"], ck["label"], ck["pred"]
            first = 300 if name == "syn1" else 511
            targets = range(first, ck["num_nodes"])
            self.desc = f"{name}: all {ck['num_nodes'] - first} motif nodes ({first}..{ck['num_nodes'] - 1}) as one batch"
            gp = os.path.join(helpers.GOLDEN, name + "_full_explain.npz")
            if os.path.exists(gp):
                self.golden = np.load(gp)
        elif name == "ba100k":
            # configs[4]: BA-House scaled to 100k nodes (42857 BA + 11428 houses, 1 % random edges), encoder = the syn1
            # checkpoint (same D/H/C), targets = ONE fixed random sample of motif nodes (the same on every rank)
            from gnn_model_explainer_amd.utils import synthetic
            n, edges, self.label = synthetic.ba_house(42857, 11428, seed=0)
            csr = synthetic.csr_from_edges(n, edges)
            self.feat = np.ones((n, 10), np.float32)
            self.pred = synthetic.sparse_gcn_predict(csr, self.feat, ck["sd"])
            self.idx = KHopIndex(csr, 3)
            rng = np.random.default_rng(1234)
            targets = np.sort(rng.choice(np.arange(42857, n), num_targets, replace=False))
            self.desc = f"BA-House x100k (99997 nodes): {num_targets} sampled motif nodes (seed 1234), one fixed set"
        else:
            raise SystemExit("unknown workload " + name)
        self.targets = np.asarray([int(t) for t in targets], np.int64)

    def dense_subgraph(self, t, nb, row, mask0):
        from gnn_model_explainer_amd.engine import Subgraph
        return Subgraph(self.idx.sub_adjacency(nb), self.feat[nb], int(self.label[t]), int(row), np.argmax(self.pred[nb], 1), mask0)


def _oracle_worker(args):
    """One single-thread CPU process: the oracle ("port": torch-autograd restatement, bit-identical to the reference on
    CPU) on its share of the sample.  -> [(index, seconds, masked_adj, sigmoid(feat_mask))]"""
    subs, sd, iters = args[:3]
    graph_mode = len(args) > 3 and args[3]
    import torch as th
    th.set_num_threads(1)
    sys.path.insert(0, ROOT)
    from oracle import reference_restatement as rr
    sdt = {k: th.tensor(v) for k, v in sd.items()}
    out = []
    for k, s in subs:
        t0 = time.perf_counter()
        o = rr.MaskOptimOracle(th.tensor(s.adj), th.tensor(s.feat), sdt, s.gt_label, s.pred_label, s.target_row, graph_mode=graph_mode,
                               mask0=th.tensor(s.mask0))
        ma = o.run(iters)
        out.append((k, time.perf_counter() - t0, np.asarray(ma), th.sigmoid(o.feat_mask.detach()).numpy()))
    return out


def cpu_baselines(wl, sample, iters):
    """The oracle timed on this host's cores over a size-stratified sample of the same targets:
      * process-parallel: one single-thread worker per sample target (the reference is dispatch-bound - ~700 tiny aten
        ops per epoch - so one thread per target is its best configuration; 8 intra-op threads are slower);
      * one thread, one process, on a sub-sample.
    Returns (cpu_baseline dict, cpu_baseline_1thread dict, {sample index: (masked_adj, feat_sig)})."""
    import multiprocessing as mp
    procs = max(1, min(len(sample), os.cpu_count() or 1))
    jobs = [([(k, s) for k, s in sample[p::procs]], wl.ck["sd"], iters) for p in range(procs)]
    ns = [s.adj.shape[0] for _, s in sample]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        pool.map(_noop, range(procs))            # start-up (interpreter + torch import) is not the reference's work
        t0 = time.perf_counter()
        parts = pool.map(_oracle_worker, jobs)
        dt = time.perf_counter() - t0
    res = {k: (ma, fs) for part in parts for k, _, ma, fs in part}
    cpu_s = sum(s for part in parts for _, s, _, _ in part)
    base = {"value": len(sample) / dt, "unit": "explained nodes/s", "cores": procs, "host_cpus": os.cpu_count(), "cgroup_cpu_quota_cores": cpu_quota_cores(), "kind": "port",
            "sample": f"{len(sample)} of {len(wl.targets)} targets (size-stratified, n={ns}), {iters} iters each, "
                      f"oracle/reference_restatement.py (bit-identical to the reference) on torch {torch.__version__} CPU, "
                      f"{procs} single-thread processes in parallel, wall {dt:.1f} s, {cpu_s:.1f} CPU-seconds; the LOOP only (explain.py:137-146 on "
                      f"pre-extracted sub-graphs) - the reference's Explainer.explain additionally runs its dense neighbourhood extraction, and "
                      f"/root/reference itself does not exist on this box (the port is pinned bit-identical to it by tests/test_oracle_golden.py)"}
    sub = sample[::4]
    t0 = time.perf_counter()
    _oracle_worker((sub, wl.ck["sd"], iters))
    dt1 = time.perf_counter() - t0
    one = {"value": len(sub) / dt1, "unit": "explained nodes/s", "cores": 1, "kind": "port",
           "sample": f"{len(sub)} of the sample above (n={[s.adj.shape[0] for _, s in sub]}), one process, one thread, {dt1:.1f} s"}
    return base, one, res


def bench_config4(args, dev, log):
    """BASELINE.json configs[3]: graph-level explanation (GcnEncoderGraph, models.py:269-316), per-graph edge masks batched across
    all molecules on one GPU.  The Mutagenicity files are not available offline: 4337 synthetic molecule-like graphs
    (utils/synthetic.molecule_like_graphs: random trees + ring closures, 10..100 atoms, 14 one-hot atom types, padded to 100 x 100 like
    the reference's GraphSampler) and the GcnEncoderGraph weights of the fixture tests/golden/config4_windows.npz.  The dataset (packed
    adjacencies + features) is resident in HBM like the node-mode graph; a step = the whole batch end to end: seeded host RNG of the
    4337 initial masks (C++ threads, drawn while the previous step optimises), one H2D copy + scatter, 300 iterations, gather + D2H
    of the masks on the edges and of the feature masks.  512 size-stratified graphs are checked against the LIVE reference's outputs."""
    import helpers
    from gnn_model_explainer_amd import engine
    from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
    from gnn_model_explainer_amd.utils import synthetic
    z = np.load(os.path.join(helpers.GOLDEN, "config4_windows.npz"))
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    G = int(z["total_graphs"])
    A, X, nn, y = synthetic.molecule_like_graphs(G, seed=0)
    n = A.shape[1]
    subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, None) for g in range(G)]
    t0 = time.perf_counter()
    job = MaskOptimJob(subs, sd, graph_mode=True)
    job._edge_layout()
    E = int(job._eoff[-1])
    rc_host = job._rc[:E].cpu().numpy()
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3
    log(f"config4: {G} graphs packed and routed in {setup_ms:.0f} ms, {E} undirected edges")
    hy = Hyper(num_iters=args.iters, edge_results_only=True)
    sizes, seeds = np.full(G, n, np.int32), 1000 + np.arange(G, dtype=np.int64)
    threads = engine.default_rng_threads()
    pins = [torch.empty(G * n * n, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    out_pin = torch.empty(E, dtype=torch.float32, pin_memory=True)
    fm_pin = torch.empty(G, engine.FEAT_STRIDE, dtype=torch.float32, pin_memory=True)
    rng_ms = []

    def draw(slot):
        t_r = time.perf_counter()
        engine.init_edge_masks_raw(sizes, seeds=seeds, threads=threads, out=pins[slot])
        rng_ms.append((time.perf_counter() - t_r) * 1e3)

    def run(k_steps):
        draw(0)
        for k in range(k_steps):
            th = None
            if k + 1 < k_steps:                  # the next batch's masks are drawn while this one optimises
                th = threading.Thread(target=draw, args=((k + 1) & 1,))
                th.start()
            job.set_masks_raw(pins[k & 1])
            job.launch(hy)
            vals = job.gather_edges_device()
            out_pin.copy_(vals[:E], non_blocking=True)
            fm_pin.copy_(job.fmask, non_blocking=True)
            torch.cuda.synchronize()             # the result of batch k is on the host
            if th is not None:
                th.join()
    run(max(1, args.warmup))
    rng_ms.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = G * args.steps / dt
    rts = job.resident_times()
    route = job.route()
    # loop only (resident masks), as a secondary
    job.launch(hy); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        job.set_masks_raw_resident()
        job.launch(hy)
    torch.cuda.synchronize()
    loop_ms = (time.perf_counter() - t1) / 3 * 1e3
    # ---- parity: the 512 fixture graphs against the live reference's outputs ----
    gids = z["graphs"]
    eoff_all = job._eoff
    vals = np.concatenate([out_pin.numpy()[eoff_all[g]:eoff_all[g + 1]] for g in gids])
    feat_sig = 1.0 / (1.0 + np.exp(-fm_pin.numpy()[gids, :job.D].astype(np.float64)))
    parity = None
    if args.iters == int(z["full_epochs"]):
        assert np.array_equal(np.concatenate([[0], np.cumsum([eoff_all[g + 1] - eoff_all[g] for g in gids])]), z["eoff"])
        d = np.abs(vals.astype(np.float64) - z["vals"].astype(np.float64))
        err = np.asarray([d[a:b].max() if b > a else 0.0 for a, b in zip(z["eoff"][:-1], z["eoff"][1:])])
        ferr = np.abs(feat_sig - z["feat_sig"]).max(1)
        well = (z["cond_mask"] <= WELL) & (z["cond_feat"] <= WELL)
        # No percentage rule (VERDICT r4 #4b).  The full-horizon GATE of graph mode is decision-based and lives in tests/test_decision_parity.py
        # (every decision of every epoch against the live reference's); this in-run check requires that every miss on a graph the closed form
        # agrees on is EXPLAINED by a window in which, on the CPU alone, a max-pool margin / ReLU gate comes within fp32 round-off of its
        # boundary or the two CPU implementations / a 1-ulp perturbation already diverge.
        W = helpers.Windows("config4")
        e = np.maximum(err, ferr)
        miss = np.nonzero(well & (e > PARITY_TOL))[0]
        explained = [int(gids[k]) for k in miss if W.flagged[k].any()]
        parity = {"reference": "outputs of /root/reference itself on 512 size-stratified graphs (tests/golden/config4_windows.npz)", "graphs_checked": int(len(gids)),
                  "coverage": f"{len(gids)} of the job's {G} graphs are compared with the reference (size-stratified sample; the reference needs ~1 CPU-minute per graph)",
                  "non_chaotic": int(well.sum()), "within_tolerance": int((well & (e <= PARITY_TOL)).sum()), "worst_non_chaotic": float(e[well].max()),
                  "rule": "every miss on a non-chaotic graph must have a CPU-flagged window; the decision-by-decision gate is tests/test_decision_parity.py",
                  "misses_with_a_flagged_window": len(explained), "misses_unexplained": [int(gids[k]) for k in miss if not W.flagged[k].any()],
                  "tolerance": PARITY_TOL}
        if parity["misses_unexplained"] and not args.no_parity_gate:
            raise SystemExit("PARITY FAILURE: " + json.dumps(parity))
    kagg = job.D + 2 * job.H
    top = int(np.argmax(rts))
    ms = rts[top]
    names = ["k_resident<1>", "k_resident<2>", "k_resident<3>", "k_sparse_resident<7,10,graph,1024>", "k_sparse_resident<7,10,graph,256>",
             "k_sparse_resident<7,10,graph,64>", "k_sparse_large", "k_sparse_resident<.., 512>"]
    rcode = {3: 4, 4: 5, 5: 6, 6: 7, 7: 8}.get(top, top + 1)
    sel = route == rcode
    # the executed-work model of the node-mode lines (gnn_model_explainer_amd/utils/work_model.py), graph mode: three full layers on all rows, max-pool
    # head, three row-local backwards; SURVEY.md section 8(d)'s dense figure (6 n^2 (D+2H) flop with the reference's padded n = 100) as `dense_equivalent`
    from gnn_model_explainer_amd.utils import work_model as wm
    lat = wm.load_latency_table(ROOT)
    idx = np.nonzero(sel)[0]
    eoff_np = np.asarray(job._eoff)
    S = wm.target_structure(np.full(G, n), eoff_np, rc_host, None, graph_mode=True)
    Ssel = {kk: v[idx] for kk, v in S.items()}
    fl = wm.executed_flops_per_iter(Ssel, job.D, job.H, job.H, job.C, 0, graph_mode=True)
    by = wm.executed_lds_bytes_per_iter(Ssel, job.D, job.H, job.H, job.C, 0, graph_mode=True)
    ops = wm.chain_ops_per_iter(Ssel, job.D, job.H, job.H, job.C, 0, graph_mode=True)
    ch = wm.chain_ns_per_iter(ops, lat)
    wpc = {4: 1, 5: 2, 6: 6}.get(rcode, 1)
    bnd = wm.launch_bounds(fl, by, ch, args.iters, np.arange(len(idx)), wpc)
    tb = {"flops": bnd["flops_s"], "lds": bnd["lds_s"], "chain": bnd["chain_s"]}
    bname = max(tb, key=tb.get)
    f_alg = 6.0 * float(sel.sum()) * n * n * kagg * args.iters        # the reference optimises the padded 100 x 100 graphs
    roof = {"kernel": names[top] + " (edge-sparse on-chip-resident optimisation, graph mode)", "bound": bname, "achieved": args.iters / (ms * 1e-3),
            "peak": args.iters / tb[bname], "unit": "iterations/s of the launch", "frac": tb[bname] / (ms * 1e-3), "traffic": None, "avg_launch_us": ms * 1e3,
            "definition": "executed-work model (work_model.py): the largest of the lower bounds executed flops / executed LDS bytes on the busy CUs / critical chain "
                          "of dependent operations at unloaded measured latencies (saturated regime: sum of the chains / (CUs x workgroups per CU)), over the measured "
                          "launch time (HIP events on its lane stream, in situ)",
            "model": {"workgroups": bnd["workgroups"], "workgroups_per_cu": wpc, "lower_bounds_ms": {kk: v * 1e3 for kk, v in tb.items()},
                      "frac_by_bound": {kk: v / (ms * 1e-3) for kk, v in tb.items()},
                      "chain_ns_per_iteration_mean": bnd["chain_ns_per_iter_mean"], "latency_source": lat["source"],
                      "executed_tflops": bnd["executed_flops"] / (ms * 1e-3) / 1e12},
            "dense_equivalent": {"mfma_frac": f_alg / (ms * 1e-3) / MFMA_F32_PEAK,
                                 "note": "6 n^2 (D+2H) flop per graph and iteration with the reference's padded n = 100: not what the edge formulation executes"},
            "launches": {names[i]: {"targets": int((route == {3: 4, 4: 5, 5: 6, 6: 7, 7: 8}.get(i, i + 1)).sum()), "ms_total": rts[i]} for i in range(8) if rts[i]}}
    out = {"metric": "explained graphs/sec (300 mask-opt iters, graph-level explanation)", "value": value, "unit": "explained graphs/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic (4337 molecule-like graphs, 14 one-hot atom types; GcnEncoderGraph weights of the fixture)",
           "config": {"workload": f"config4: Mutagenicity-like graph mode, {G} graphs x {n} (padded) as one batch, {args.iters} iters, Adam lr 0.1",
                      "targets_total": G, "routing": {int(k): int(v) for k, v in zip(*np.unique(route, return_counts=True))}, "parallelism": "single GPU"},
           "value_definition": "graphs / wall time of the whole batch with the dataset resident: seeded host RNG (overlapped with the previous batch), H2D + "
                               "scatter of the initial masks, the 300 iterations, gather + D2H of the masks",
           "loop_only": {"value": G / (loop_ms * 1e-3), "unit": "explained graphs/s", "ms_per_step": loop_ms},
           "end_to_end_stage_ms": {"host_rng_ms": float(np.mean(rng_ms)) if rng_ms else None, "rng_threads": threads, "masks_h2d_MB": G * n * n * 4 / 1e6,
                                   "setup_once_ms": setup_ms},
           "roofline": roof}
    if parity is not None:
        out["parity"] = parity
    if not args.no_cpu_baseline:
        order = np.argsort(nn, kind="stable")
        pick = order[np.linspace(0, G - 1, 32).astype(int)]
        sample = [(int(g), Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(int(g), n).numpy())) for g in pick]
        import multiprocessing as mp
        procs = max(1, min(len(sample), os.cpu_count() or 1))
        jobs = [([(k, sg) for k, sg in sample[p_::procs]], sd, args.iters, True) for p_ in range(procs)]
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(_noop, range(procs))
            t0 = time.perf_counter()
            parts = pool.map(_oracle_worker, jobs)
            dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample) / dtc, "unit": "explained graphs/s", "cores": procs, "host_cpus": os.cpu_count(), "cgroup_cpu_quota_cores": cpu_quota_cores(), "kind": "port",
                               "sample": f"{len(sample)} of {G} graphs (size-stratified), {args.iters} iters each, oracle/reference_restatement.py (bit-identical to "
                                         f"the reference) on torch {torch.__version__} CPU, {procs} single-thread processes, wall {dtc:.1f} s; the loop only - "
                                         "the reference's Explainer.explain additionally slices the graph out of the dataset"}
        errs = []
        for part in parts:
            for g, _, ma, fs in part:
                a, b = eoff_all[g], eoff_all[g + 1]
                r, c = rc_host[a:b, 0], rc_host[a:b, 1]
                errs.append(float(np.abs(ma[r, c] - out_pin.numpy()[a:b]).max()) if b > a else 0.0)
        out.setdefault("parity", {})["vs_cpu_oracle"] = {"graphs": len(errs), "within_1e-5": int(sum(e <= PARITY_TOL for e in errs)), "max_abs_err": max(errs)}
    print(json.dumps(out))


ALL_BINS = (0, 32, 128, 512, 2048, 8192, 16383, 24000, 1 << 30)          # strata of the all-node workload: sub-graph nodes in (lo, hi]
ALL_TAKE = (2048, 2048, 2048, 1024, 512, 256, 256, 256)                   # sampled targets per stratum: every batch fills the chip (one workgroup per CU in
                                                                          # the strata beyond 512 nodes), 512 of them beyond 16 383 nodes


def all_node_sample(sizes, seed=2026, take=ALL_TAKE, scale=1.0):
    """The seed-fixed stratified sample of ALL nodes of the graph: per size stratum (ALL_BINS) up to take[s] x scale nodes, evenly spaced over the
    stratum's nodes sorted by sub-graph size (so that every stratum's tail is in the sample), ties broken by a seeded permutation.
    -> list of (lo, hi, population count, sampled node ids ascending)"""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(len(sizes))
    out = []
    for s, (lo, hi) in enumerate(zip(ALL_BINS[:-1], ALL_BINS[1:])):
        ids = perm[(sizes[perm] > lo) & (sizes[perm] <= hi)]
        ids = ids[np.argsort(sizes[ids], kind="stable")]
        k = min(len(ids), max(1, int(round(take[s] * scale)))) if len(ids) else 0
        pick = ids[np.unique(np.linspace(0, len(ids) - 1, k).astype(np.int64))] if k else ids[:0]
        out.append((lo, hi, int(len(ids)), np.sort(pick).astype(np.int64)))
    return out


def bench_ba100k_all(args, dev, log, dist, world, rank):
    """BASELINE.json configs[4] AS NAMED - "explain every node" of the 100k-node BA-House graph (the reference's loop explain.py:296-299 over
    explainer_main.py:309-313's node list, each through utils/graph_utils.py:147-158) - on a seed-fixed sample stratified over ALL 99 997 nodes, BA nodes
    and motif nodes alike (rounds 2-5 benchmarked motif nodes only - the easy 57 %).

    Sub-graph sizes of every node come from the device k-hop pass; the nodes are cut into size strata (ALL_BINS) and ALL_TAKE of each are sampled.
    A step = the whole sample through pipeline.BatchPipeline, one batch per stratum, every stage inside (k-hop lists, packing / sub-graph CSRs,
    routing, seeded masks, the 300 iterations, edge lists back on the host); N > 1: the sample is sharded over the ranks by modelled cost (LPT over
    per-stratum costs measured by rank 0) and every rank's masks are all-gathered as edge entries over RCCL inside the timed region.
    `value` = sampled targets x K / wall time.  The line also carries the route histogram, per-stratum milliseconds (one batch alone: end to end and
    the optimisation launch) and `every_node`: the single-GPU time for ALL nodes extrapolated stratum by stratum (population / sample x measured time)."""
    import helpers
    from gnn_model_explainer_amd import engine, parallel
    from gnn_model_explainer_amd.engine import Hyper
    from gnn_model_explainer_amd.pipeline import BatchPipeline
    from gnn_model_explainer_amd.utils import synthetic
    ck = helpers.load_ckpt("syn1")
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, ck["sd"])
    graph = engine.device_graph(csr, feat, pred)
    hy = Hyper(num_iters=args.iters, edge_results_only=True)
    lib = engine.get_library()

    # ---- sub-graph sizes of EVERY node (device k-hop size pass) ----
    engine.khop_device(graph, np.arange(8, dtype=np.int64), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sizes = np.zeros(N, np.int64)
    CH = 8192
    for a in range(0, N, CH):
        tg = np.arange(a, min(N, a + CH), dtype=np.int64)
        tg_d = engine._h2d(tg.astype(np.int32), dev)
        sr = torch.empty(2, len(tg), dtype=torch.int32, device=dev)
        sb = int(lib.gnnx_khop_scratch_bytes(graph.num_nodes, len(tg)))
        scratch = torch.empty(max(sb, 1), dtype=torch.uint8, device=dev)
        import ctypes
        engine._check(lib, lib.gnnx_khop(graph.indptr.data_ptr(), graph.indices.data_ptr(), graph.num_nodes, 3, tg_d.data_ptr(), len(tg), sr[0].data_ptr(),
                                         None, None, sr[1].data_ptr(), scratch.data_ptr(), sb, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        sizes[a:a + len(tg)] = sr[0].cpu().numpy()
    sizes_ms = (time.perf_counter() - t0) * 1e3
    sum_n2_all = float((sizes.astype(np.float64) ** 2).sum())
    strata = all_node_sample(sizes, scale=args.sample_scale)
    sample = np.sort(np.concatenate([s[3] for s in strata]))
    if os.environ.get("GNNX_WRITE_SAMPLE") and rank == 0:      # tests/golden/ba100k_all_sample.npy: (node id, sub-graph size) of the sample - the CPU suite
        np.save(os.environ["GNNX_WRITE_SAMPLE"], np.stack([sample, sizes[sample]]).astype(np.int64))      # checks the shards cut from it (tests/test_distributed.py)
    log(f"sizes of all {N} nodes: {sizes_ms:.0f} ms; sum n^2 = {sum_n2_all:.3g}; sample {len(sample)} targets in {len(strata)} strata: " +
        ", ".join(f"({lo},{hi}]: {len(pick)}/{pop}" for lo, hi, pop, pick in strata))

    # ---- N = 1 (and rank 0 of N > 1): every stratum's batch alone - end to end and the launch - for the extrapolation and the cost model ----
    per_stratum = []
    if rank == 0:
        pipe1 = BatchPipeline(graph, ck["sd"], label, hy, prepare_workers=int(os.environ.get("GNNX_PIPE_WORKERS", "4")))
        for lo, hi, pop, pick in strata:
            if not len(pick):
                per_stratum.append(dict(n_lo=lo, n_hi=hi, population=pop, sampled=0))
                continue
            list(pipe1.run([pick]))                                   # warm (allocator, code objects)
            pipe1.stats.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            em = list(pipe1.run([pick]))[0]
            alone_ms = (time.perf_counter() - t0) * 1e3
            st = dict(pipe1.stats[-1])
            reps = 6
            piped_all = []
            for _rep in range(3):      # median of three 6-batch regions (single regions scatter by 2-4 x on a loaded host: profiles/r06_bench_ba100k_all_*.json)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in pipe1.run([pick] * reps):
                    pass
                piped_all.append((time.perf_counter() - t0) * 1e3 / reps)
            piped_ms = float(np.median(piped_all))
            # the optimisation launch(es) of the batch alone, on resident inputs
            dn = engine.khop_device(graph, pick, 3)
            big = dn.sizes > pipe1.xl_min_n
            loop_ms, routes = 0.0, {}
            for sel, xl in ((~big, False), (big, True)):
                if not sel.any():
                    continue
                sub, dns = pick[sel], dn.subset(np.nonzero(sel)[0])
                if xl:
                    job = engine.XLJob(graph, dns, None, label[sub], ck["sd"])
                    job.set_masks_seeded(1000 + sub)
                    routes["xl"] = int(sel.sum())
                else:
                    job = engine.MaskOptimJob.from_csr(graph, dns, None, label[sub], ck["sd"])
                    job.set_masks_raw(engine.init_edge_masks_raw(dns.sizes, seeds=1000 + sub, threads=engine.default_rng_threads(big=True)))
                    r = job.route()
                    for k in np.unique(r):
                        routes[str(int(k))] = int((r == k).sum())
                job.launch(hy)
                torch.cuda.synchronize()
                lm = []
                for _rep in range(3):      # median of three launches on resident inputs
                    job.reset_masks()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(job.stream):
                        e0.record(job.stream)
                        job.launch(hy)
                        e1.record(job.stream)
                    torch.cuda.synchronize()
                    lm.append(e0.elapsed_time(e1))
                loop_ms += float(np.median(lm))
                job.close()
                del job
            per_stratum.append(dict(n_lo=lo, n_hi=hi, population=pop, sampled=int(len(pick)), n_mean=float(sizes[pick].mean()), n_max=int(sizes[pick].max()),
                                    sum_n2_population=float((sizes[(sizes > lo) & (sizes <= hi)].astype(np.float64) ** 2).sum()), routes=routes,
                                    one_batch_alone_ms=alone_ms, pipelined_ms_per_batch=piped_ms, pipelined_ms_per_batch_repetitions=[round(x, 2) for x in piped_all], loop_ms=loop_ms,
                                    stage_ms={k: round(float(v), 3) for k, v in st.items() if k.endswith("_ms")}, edges=int(em.eoff[-1])))
            log(f"stratum ({lo},{hi}]: {len(pick)} of {pop}: alone {alone_ms:.1f} ms, pipelined {piped_ms:.1f} ms per batch, loop {loop_ms:.1f} ms, routes {routes}")
        del pipe1
        gc.collect()
    w = np.asarray([(s["population"] / s["sampled"]) if s.get("sampled") else 0.0 for s in per_stratum]) if rank == 0 else None

    # ---- shards (N > 1): LPT over per-target costs = the stratum's measured loop time / its sample size ----
    batches = [pick for _, _, _, pick in strata if len(pick)]
    shard_info = None
    if world > 1:
        cost_s = torch.zeros(len(strata), dtype=torch.float64, device=dev)
        if rank == 0:
            cost_s[:] = torch.tensor([(s["loop_ms"] / s["sampled"]) if s.get("sampled") else 0.0 for s in per_stratum], dtype=torch.float64)
        dist.broadcast(cost_s, 0)
        cost_s = cost_s.cpu().numpy()
        # inside a stratum the cost follows the edge count; n is its proxy here (the sizes are known on every rank, the edge counts are not yet)
        tcost = np.zeros(len(sample))
        which = np.searchsorted(np.asarray(ALL_BINS[1:]), sizes[sample], side="left")
        for s in range(len(strata)):
            m = which == s
            if m.any():
                tcost[m] = cost_s[s] * sizes[sample][m] / max(1.0, sizes[sample][m].mean())
        shard = np.asarray(parallel.lpt_shards(tcost, world)[rank], np.int64)
        mine = sample[shard]
        batches = [mine[which[shard] == s] for s in range(len(strata)) if (which[shard] == s).any()]
        loads = [float(tcost[np.asarray(sh, np.int64)].sum()) for sh in parallel.lpt_shards(tcost, world)]
        shard_info = {"modelled_cost_ms_per_rank": [round(x, 2) for x in loads], "imbalance_pct": 100.0 * (max(loads) - min(loads)) / max(loads),
                      "targets_per_rank": [len(sh) for sh in parallel.lpt_shards(tcost, world)],
                      "sum_n2_per_rank": [float((sizes[sample[np.asarray(sh, np.int64)]].astype(np.float64) ** 2).sum()) for sh in parallel.lpt_shards(tcost, world)]}
        log(f"rank shards: {shard_info['targets_per_rank']} targets, modelled {shard_info['modelled_cost_ms_per_rank']} ms")

    gather = {}
    hook = None
    if dist is not None:
        # the masks of every rank, as edge entries, on every rank: one padded all-gather per batch on the fetch stream (RCCL over xGMI)
        def hook(vals_d, job_k):
            cnt = torch.tensor([vals_d.numel()], device=dev)
            cnts = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(cnts, cnt)
            emax = max(int(c.item()) for c in cnts)
            key = ("buf", emax)
            if key not in gather:
                gather[key] = (torch.zeros(emax, dtype=torch.float32, device=dev), [torch.zeros(emax, dtype=torch.float32, device=dev) for _ in range(world)])
            mine_b, all_b = gather[key]
            mine_b[:vals_d.numel()].copy_(vals_d)
            dist.all_gather(all_b, mine_b)
    # four prepare workers: a batch of large sub-graphs spends most of its preparation waiting for the device (engine walk, CSR build), and the walks of
    # several batches overlap on the chip (profiles/r06_ab_xl_threshold.txt: every node 11.2 s with two workers, 9.1 s with four)
    pipe = BatchPipeline(graph, ck["sd"], label, hy, device_hook=hook, prepare_workers=int(os.environ.get("GNNX_PIPE_WORKERS", "4")))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # with ranks whose shard lacks a stratum the number of all-gathers per step differs from rank to rank: every rank runs the SAME number of batches
    if dist is not None:
        nb = torch.tensor([len(batches)], device=dev)
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
        while len(batches) < int(nb.item()):
            batches.append(batches[-1][:1])
    for _ in range(max(1, args.warmup)):
        for _ in pipe.run(batches):
            pass
    reps = []
    last = None
    for _ in range(args.reps if args.reps > 0 else 3):
        pipe.stats.clear()
        gc.collect()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = list(pipe.run(batches))
        barrier()
        dt_r = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt_r], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_r = float(tt.item())
        reps.append(dt_r)
    dt = float(np.median(reps))
    value = len(sample) * args.steps / dt
    log(f"timed region: median {dt:.3f} s for {args.steps} steps ({len(reps)} repetitions)")
    if rank != 0:
        return
    nanf = float(np.mean([np.isnan(em.masked_adj).mean() for em in last]))
    route_hist = {}
    for s in per_stratum:
        for k, v in s.get("routes", {}).items():
            route_hist[k] = route_hist.get(k, 0) + int(round(v * s["population"] / s["sampled"]))
    every = {"nodes": int(N), "sum_n2": sum_n2_all,
             "loop_s": float(sum(wk * s["loop_ms"] for wk, s in zip(w, per_stratum) if s.get("sampled")) * 1e-3),
             "end_to_end_pipelined_s": float(sum(wk * s["pipelined_ms_per_batch"] for wk, s in zip(w, per_stratum) if s.get("sampled")) * 1e-3) + sizes_ms * 1e-3,
             "end_to_end_batches_alone_s": float(sum(wk * s["one_batch_alone_ms"] for wk, s in zip(w, per_stratum) if s.get("sampled")) * 1e-3) + sizes_ms * 1e-3,
             "sizes_of_all_nodes_ms": sizes_ms, "route_histogram_all_nodes": route_hist,
             "streaming_targets": int(route_hist.get("0", 0)),
             "dense_equivalent_s_at_8TBps": 28.0 * sum_n2_all * args.iters / HBM_PEAK,
             "note": "single-GPU time for ALL nodes, extrapolated stratum by stratum: population / sample x the stratum batch's measured time (loop: the optimisation "
                     "launches on resident inputs; end to end: batches of that composition through the pipeline, stages overlapped / one batch alone) + the size pass"}
    every["nodes_per_s_end_to_end"] = N / every["end_to_end_pipelined_s"]
    # roofline of the dominant launch: the XL launch of the largest stratum (one workgroup per target, state L2-resident)
    dom = max((s for s in per_stratum if s.get("sampled")), key=lambda s: wk_loop(s, per_stratum, w))
    alg_bytes = 28.0 * dom["n_mean"] ** 2 * dom["sampled"] * args.iters
    exe_bytes = dom["edges"] * (2 * 8 * 3 + 60.0) * args.iters                # per directed entry 3 passes x (Abar + column), per edge 60 B of planes: what the kernel moves (L2)
    roof = {"kernel": "k_sparse_large<5, 10, false, true, XL> (stratum n in (%d, %d])" % (dom["n_lo"], dom["n_hi"]), "bound": "hbm",
            "achieved": exe_bytes / (dom["loop_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": exe_bytes / (dom["loop_ms"] * 1e-3) / HBM_PEAK,
            "traffic": None, "avg_launch_us": dom["loop_ms"] * 1e3,
            "dense_equivalent": {"achieved_GBps": alg_bytes / (dom["loop_ms"] * 1e-3) / 1e9, "frac_of_8TBps": alg_bytes / (dom["loop_ms"] * 1e-3) / HBM_PEAK,
                                 "note": "SURVEY 8(d)'s 28 n^2 bytes per target and iteration over the launch time: the speed-up over a dense implementation at peak, not a utilisation"},
            "note": "edge formulation: the state of a target is O(edges) and L2-resident; one workgroup per target walks ~8 dependent phases per iteration - "
                    "latency-bound on ONE compute unit per target, the chip is filled by the targets of a batch.  `achieved` prices EVERY edge of the stratum's "
                    "sub-graphs at 108 bytes per iteration (3 passes x 16 B per directed entry + 60 B of per-edge planes): an UPPER bound on what the kernel moves "
                    "(it iterates the entries within two hops of the target; the far edges run a closed recursion once), and those bytes come from L2 / LDS, not "
                    "from HBM - `frac` is that bound over 8 TB/s, not an HBM utilisation; `traffic` (PMC) is not collected for this workload"}
    out = {"metric": "explained nodes/sec (300 mask-opt iters, k-hop subgraph) on a stratified sample of ALL nodes of BA-House x100k",
           "value": value, "unit": "explained nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BA-House x100k (99997 nodes, 297 k edges), every-node job: %d targets sampled over ALL nodes in %d size strata (seed 2026), %d iters" %
                                  (len(sample), len([s for s in strata if len(s[3])]), args.iters), "parallelism": f"targets sharded over {world} GPU(s) by modelled cost"},
           "repetitions": {"count": len(reps), "seconds": reps}, "nan_fraction": nanf, "strata": per_stratum, "every_node": every, "roofline": roof,
           "shards": shard_info, "rccl_world_size": world, "xl_min_n": pipe.xl_min_n}
    if not args.no_cpu_baseline and world == 1:
        # CPU baseline: the bit-pinned port on a bounded sample of the SAME workload - small strata only (a dense 4000 x 4000 autograd loop takes minutes)
        wl = Workload.__new__(Workload)
        from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
        wl.ck, wl.idx, wl.feat, wl.label, wl.pred, wl.targets = ck, KHopIndex(csr, 3), feat, label, pred, sample
        small = [t for t in sample if sizes[t] <= 700]
        pick = [small[i] for i in np.linspace(0, len(small) - 1, 16).astype(int)]
        subs = []
        for t in pick:
            nb = wl.idx.neighbors(int(t))
            subs.append((int(t), wl.dense_subgraph(int(t), nb, int(np.searchsorted(nb, t)), helpers.seeded_mask0(int(t), len(nb)).numpy())))
        base, one, res = cpu_baselines(wl, subs, args.iters)
        out["cpu_baseline"] = base
        out["cpu_baseline_1thread"] = one
        # the same targets through the engine: parity with the port (the in-run checker)
        tg = np.asarray(pick, np.int64)
        em = list(BatchPipeline(graph, ck["sd"], label, hy).run([tg]))[0]
        errs = []
        for k, t in enumerate(pick):
            ma = res[int(t)][0]
            a, b = int(em.eoff[k]), int(em.eoff[k + 1])
            r, c = em.rc[a:b, 0], em.rc[a:b, 1]
            errs.append(float(np.abs(ma[r, c] - em.masked_adj[a:b]).max()) if b > a else 0.0)
        out["parity"] = {"vs_cpu_oracle": {"targets": len(errs), "within_1e-5": int(sum(e <= PARITY_TOL for e in errs)), "max_abs_err": max(errs)},
                         "note": "the XL kernel is pinned to the live reference's optimiser state on three sub-graphs beyond 16 383 nodes by tests/test_xl_reference_windows.py "
                                 "and bit for bit to route 7 by tests/test_xl_route.py"}
    print(json.dumps(out))


def wk_loop(s, per_stratum, w):
    return w[per_stratum.index(s)] * s["loop_ms"]


def _noop(_):
    import torch as th   # noqa: F401  (pays the import inside the pool start-up, outside the timed region)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="batches per timed region (20 / 5 = the command the round driver runs; --steps 300 --warmup 10: the steady state)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--workload", default=None, choices=["syn1", "ba100k", "ba100k-all", "syn4", "syn5", "config4"],
                    help="default: syn1 at 1 GPU (the metric's configuration), ba100k-all (the every-node job of configs[4] on a stratified all-node sample: "
                         "the scaling curve) at N > 1; ba100k: the motif-only 16 384-target set of rounds 2-5")
    ap.add_argument("--sample-scale", type=float, default=1.0, help="ba100k-all: scale of the per-stratum sample sizes (ALL_TAKE)")
    ap.add_argument("--targets", type=int, default=16384, help="ba100k: size of the fixed target set")
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of hipGraph replay (streaming kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="streaming kernels for every target (no on-chip-resident path)")
    ap.add_argument("--no-parity-gate", action="store_true", help="measurement sessions: report parity but do not fail the run")
    ap.add_argument("--no-single-gpu-leg", action="store_true", help="N > 1: skip the same-workload single-GPU run on rank 0")
    ap.add_argument("--no-calibration", action="store_true", help="N > 1: shard by the default cost table instead of measuring it on rank 0")
    ap.add_argument("--reps", type=int, default=0, help="end-to-end line: the K-step timed region is repeated this many times (each repetition: exactly "
                                                       "K batches between two barriers) and the MEDIAN repetition is reported - one 20-batch region lasts "
                                                       "50 ms and ten consecutive runs of it spanned 134.8-166.6 k nodes/s (profiles/r03_bench_syn1_ten_runs.txt)")
    ap.add_argument("--loop-only", action="store_true", help="N = 1: report the resident-input loop rate as `value` (rounds 1-2) instead of the end-to-end rate")
    args = ap.parse_args()
    if os.environ.get("GNNX_SWITCH_INTERVAL"):      # (measurement knob: CPython's GIL hand-over interval, default 5 ms)
        sys.setswitchinterval(float(os.environ["GNNX_SWITCH_INTERVAL"]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # HIP's host-side waits SPIN by default (one core at 100 % per waiting thread).  The ranks of a node share its cores - the GPU box's container
    # has a quota of 16 - so a sharded run asks for sleeping waits before the first HIP call (hipDeviceScheduleBlockingSync = 4); GNNX_BLOCKING_SYNC=0/1 overrides.
    if int(os.environ.get("GNNX_BLOCKING_SYNC", "1" if world > 1 else "0")):
        try:
            import ctypes
            ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(ctypes.c_uint(4))
        except OSError:
            pass
    # one rank per GPU; GNNX_DIST_BACKEND=gloo lets several ranks share one GPU (a smoke test of the sharded path on a 1-GPU box)
    backend = os.environ.get("GNNX_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    name = args.workload or ("syn1" if world == 1 else "ba100k-all")

    from gnn_model_explainer_amd import engine, parallel
    from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    if name == "config4":
        if world > 1:
            raise SystemExit("--workload config4 is a single-GPU line")
        return bench_config4(args, dev, log)
    if name == "ba100k-all":
        if args.steps == 20 and args.warmup == 5 and world == 1 and args.workload:      # (the syn1 defaults: a step here is the whole sample)
            args.steps, args.warmup = 3, 1
        return bench_ba100k_all(args, dev, log, dist, world, rank)
    wl = Workload(name, args.targets)
    graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)          # the input graph lives in HBM (uploaded once)
    hy = Hyper(num_iters=args.iters, use_graph=not args.no_graph, use_resident=not args.no_resident,
               edge_results_only=True)       # every result in this file leaves as an edge list (fetch_edges / gather_edges_device)
    engine.khop_device(graph, wl.targets[:1], 3)                         # load the code objects before anything is timed
    torch.cuda.synchronize()

    # ---------------------------------------------------------------- one batch end to end (this rank's shard) ----------------
    def build(targets, timings=None):
        """k-hop sets + packing + routing on the device, seeded masks on the host, one H2D copy."""
        tm = {}
        t0 = time.perf_counter()
        dn = engine.khop_device(graph, targets, 3)
        tm["khop_device_ms"] = (time.perf_counter() - t0) * 1e3
        # the host draws the seeded initial masks (it only needs the sizes) while the device builds the plan and packs
        rng_threads = rng_threads_for(float((dn.sizes.astype(np.float64) ** 2).sum()), world)
        box = {}

        def draw():
            t_r = time.perf_counter()
            box["raw"] = engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets, pin=True, threads=rng_threads)
            box["ms"] = (time.perf_counter() - t_r) * 1e3
        # large batches on the edge-sparse kernels (what pipeline.BatchPipeline does too): the engine of the seeded draw walks on the device, the
        # host transforms the word pairs of the edge entries only (gnnx_mt_edge_words + gnnx_host_transform_edge_words)
        edges_path = (float((dn.sizes.astype(np.float64) ** 2).sum()) > 2e7 and not args.no_resident and engine.pair_staging_ok()
                      and os.environ.get("GNNX_PIPE_EDGE_DRAW", "1") != "0" and os.environ.get("GNNX_PIPE_DEVICE_WALK", "1") != "0")
        t0 = time.perf_counter()
        th = threading.Thread(target=draw)
        if not edges_path:
            th.start()
        job = MaskOptimJob.from_csr(graph, dn, None, wl.label[targets], wl.ck["sd"])
        torch.cuda.synchronize()
        tm["plan_pack_analyze_ms"] = (time.perf_counter() - t0) * 1e3
        if edges_path and not np.isin(job.route(), (4, 5, 6, 7, 8)).all():
            edges_path = False
            th.start()
        tm["host_rng_threads"] = rng_threads
        if edges_path:
            t1 = time.perf_counter()
            words = job.draw_edge_words_device(1000 + targets)
            E_ = int(job._eoff[-1])
            words_h = words.cpu()
            tm["device_walk_ms"] = (time.perf_counter() - t1) * 1e3
            t1 = time.perf_counter()
            vals = engine.transform_edge_words(dn.sizes, 1000 + targets, job._eoff, job._rc[:E_].cpu().numpy(), words_h, threads=rng_threads)
            tm["host_rng_ms"] = tm["host_transform_ms"] = (time.perf_counter() - t1) * 1e3
            tm["plan_and_rng_overlapped_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            job.set_masks_on_edges(vals)
        else:
            th.join()
            raw = box["raw"]
            tm["host_rng_ms"] = box["ms"]
            tm["plan_and_rng_overlapped_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            job.set_masks_raw(raw)
        torch.cuda.synchronize()
        tm["mask_h2d_scatter_ms"] = (time.perf_counter() - t0) * 1e3
        if timings is not None:
            timings.update(tm)
        return dn, job

    cost_table = parallel.DEFAULT_COST_TABLE.copy()
    if world > 1:
        sizes = engine.khop_device(graph, wl.targets, 3).sizes.astype(np.float64)
        if not args.no_calibration:
            # the per-class constants of the cost model, measured on THIS workload and machine by rank 0 (one saturated batch per kernel
            # class, outside the timed region) and broadcast, so that every rank cuts the same shards
            if rank == 0:
                def run_batch(idx):
                    t_sub = wl.targets[np.asarray(idx, np.int64)]
                    dn_c = engine.khop_device(graph, t_sub, 3)
                    job_c = MaskOptimJob.from_csr(graph, dn_c, None, wl.label[t_sub], wl.ck["sd"])
                    job_c.set_masks_raw(engine.init_edge_masks_raw(dn_c.sizes, seeds=1000 + t_sub, threads=engine.default_rng_threads()))
                    job_c.launch(hy)
                    torch.cuda.synchronize()
                    t_c = time.perf_counter()
                    job_c.set_masks_raw_resident()
                    job_c.launch(hy)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t_c) * 1e3
                    job_c.close()
                    return ms
                cost_table = parallel.calibrate_cost_table(sizes, run_batch)
            tt = torch.tensor(cost_table, device=dev, dtype=torch.float64)
            dist.broadcast(tt, 0)
            cost_table = tt.cpu().numpy()
            log(f"cost table (us per target, classes n <= 32 / 128 / 512 / 16383): {np.round(cost_table, 2).tolist()}")
        shard = parallel.lpt_shards(parallel.target_cost(sizes, cost_table), world)[rank]
        my_targets = wl.targets[np.asarray(shard, np.int64)]
    else:
        shard, my_targets = list(range(len(wl.targets))), wl.targets
    e2e = {}
    t_e2e = time.perf_counter()
    dn, job = build(my_targets, e2e)
    log(f"workload built: {len(my_targets)} targets on rank 0, sum n^2 = {job.sum_n2:.3g}")

    gather_bufs = {}
    if dist is not None:
        em0 = job.fetch_edges()                                       # edge structure + counts (fixed for the batch)
        cnt = torch.tensor([int(em0.eoff[-1])], device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        emax = max(int(c.item()) for c in cnts)
        gather_bufs["mine"] = torch.zeros(emax, dtype=torch.float32, device=dev)
        gather_bufs["all"] = [torch.zeros(emax, dtype=torch.float32, device=dev) for _ in range(world)]
        gather_bufs["counts"] = [int(c.item()) for c in cnts]

    jobs = [job]     # the warm batch below replaces the job the timed steps run on

    def step():
        job = jobs[0]
        job.reset_masks()                     # device op: re-spread the resident RNG stream over the padded masks / the resident edge values over M
        job.launch(hy)
        if dist is not None:                  # the masks of every rank, as edge entries, on every rank (RCCL over xGMI)
            vals = job.gather_edges_device()
            gather_bufs["mine"][:vals.numel()].copy_(vals)
            dist.all_gather(gather_bufs["all"], gather_bufs["mine"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    e2e["first_run_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    em = job.fetch_edges()
    e2e["edges_d2h_ms"] = (time.perf_counter() - t0) * 1e3
    e2e["total_ms"] = (time.perf_counter() - t_e2e) * 1e3
    if world == 1:
        # the same batch once more, end to end, now that code objects and the allocators are warm - a long-lived process
        # recycles the previous batch's device buffers (a fresh 28 GB allocation for the 16 384-target set costs 0.7 s, far
        # more than the batch itself): the steady-state cost of a batch.  The timed steps below run on this job.
        job.close()
        jobs[0] = None
        del job, dn
        gc.collect()                 # the job's tensors go back to the caching allocator now, not whenever the cycle collector runs
        torch.cuda.synchronize()
        warm = {}
        t_w = time.perf_counter()
        dn, job = build(my_targets, warm)
        job.launch(hy)
        job.fetch_edges()
        warm["total_ms"] = (time.perf_counter() - t_w) * 1e3
        jobs[0] = job
        e2e["warm"] = warm
    for _ in range(max(0, args.warmup - 1)):
        step()
    barrier()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    log(f"timed region done: {dt:.3f} s")
    n_targets = len(wl.targets)
    value = n_targets * args.steps / dt
    loop_only = {"value": value, "unit": "explained nodes/s", "ms_per_step": dt / args.steps * 1e3,
                 "note": "the optimisation alone on a batch whose inputs are already packed in HBM: gnnx_scatter_masks (re-spread of the resident "
                         "RNG stream) + gnnx_run, K steps back to back; rounds 1-2 reported this as `value`"}
    e2e_stats = None
    e2e_em = None
    if not args.loop_only:
        # ---- the metric of SURVEY.md section 8(d): targets / wall time of the WHOLE batched job - k-hop extraction, plan, packing, routing,
        # seeded host RNG, H2D, the 300 iterations, gather + D2H of the masks - with only the graph resident.  A step is one such batch
        # (N > 1: every rank's shard of the fixed target set, and the masks of all ranks all-gathered as edge entries over RCCL before
        # they leave the device); consecutive batches overlap their stages (pipeline.BatchPipeline): batch k + 1 is prepared and batch
        # k - 1 fetched while batch k optimises.  K steps are timed from the submission of the first batch to the arrival of the last
        # result on the slowest rank; the region is repeated --reps times and the median repetition reported.
        from gnn_model_explainer_amd.pipeline import BatchPipeline
        hook = None
        if dist is not None:
            def hook(vals_d, job_k):          # on the fetch stream, before the D2H copy of the rank's own values
                gather_bufs["mine"][:vals_d.numel()].copy_(vals_d)
                dist.all_gather(gather_bufs["all"], gather_bufs["mine"])
        pipe = BatchPipeline(graph, wl.ck["sd"], wl.label, hy, device_hook=hook,
                             rng_threads_big=rng_threads_for(float((dn.sizes.astype(np.float64) ** 2).sum()), world))
        for _ in pipe.run([my_targets] * max(1, args.warmup)):
            pass
        reps = []
        host_cpu_s = []
        last = None
        for _ in range(args.reps if args.reps > 0 else (9 if world == 1 else 5)):   # default: nine repetitions of a millisecond-scale region (syn1: 40 ms each), five of the sharded one
            pipe.stats.clear()
            gc.collect()
            gc.disable()                         # (no cyclic-GC pause inside a 80 ms timed region: the batches' objects are freed by reference counting)
            barrier()
            c0 = time.process_time()             # CPU time of every thread of this process (Python stages + the C++ draw pool)
            t0 = time.perf_counter()
            for last in pipe.run([my_targets] * args.steps):
                pass
            barrier()
            dt_r = time.perf_counter() - t0
            gc.enable()
            host_cpu_s.append((time.process_time() - c0) / args.steps)
            if dist is not None:
                tt = torch.tensor([dt_r], device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_r = float(tt.item())
            keys = sorted({k for st in pipe.stats for k in st})
            reps.append((dt_r, {k: float(np.mean([st[k] for st in pipe.stats if k in st])) for k in keys}))
        order = sorted(range(len(reps)), key=lambda i: reps[i][0])
        dt, e2e_stats = reps[order[len(order) // 2]]
        value = n_targets * args.steps / dt
        e2e_em = last
        e2e_stats["repetitions"] = {"count": len(reps), "reported": "median", "values": [n_targets * args.steps / r[0] for r in reps],
                                    "spread_pct": 100.0 * (max(r[0] for r in reps) - min(r[0] for r in reps)) / dt}
        # Host work per step and where it would bind: the ranks of a node share the container's CPU quota, the total host work of
        # the fixed target set does not shrink with N, the loop per rank does: host-bound once  host core-seconds / quota  >  loop / N.
        quota = cpu_quota_cores() or float(os.cpu_count() or 1)
        hcs = float(np.median(host_cpu_s)) * world          # this rank's share x ranks = the whole job's host work per step
        loop_s = loop_only["ms_per_step"] * 1e-3 * (1 if world == 1 else world)    # (N > 1: loop_only above is the slowest rank's shard)
        e2e_stats["host_bound_projection"] = {"host_core_seconds_per_step": hcs, "cpu_quota_cores": quota, "host_floor_ms_per_step": hcs / quota * 1e3,
                                              "loop_ms_per_step_one_gpu": loop_s * 1e3, "knee_n_gpus": (loop_s * quota / hcs) if hcs > 0 else None,
                                              "note": "process CPU time (all threads) per batch x ranks; beyond knee_n_gpus GPUs on this box the step is bound by "
                                                      "host core-seconds / quota, not by the optimisation"}
        e2e_stats["rng_threads"] = pipe.rng_threads
        e2e_stats["cpu_affinity"] = {"cpus": len(os.sched_getaffinity(0)), "confined_to_one_numa_node": _pkg.NUMA_CPUS is not None,
                                     "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}
        e2e_stats["prepare_workers"] = pipe.prepare_workers
        e2e_stats["optimisations_in_flight"] = pipe.depth_now
        e2e_stats["optimisations_in_flight_rule"] = ("auto: ceil(1.3 x 256 CUs / CUs one launch keeps busy), 2..5" if pipe.auto_depth else "fixed")
        e2e_stats["rng_threads_big_batches"] = pipe.rng_threads_big
        e2e_stats["stream_candidates_rejected"] = len(pipe._rejected)          # (hardware-queue calibration of the pipeline's six streams)
        e2e_stats["streams_without_own_queue"] = getattr(pipe, "queue_fallbacks", 0)
        log(f"end-to-end pipelined region done: median {dt:.3f} s for {args.steps} batches over {len(reps)} repetitions")

    # ---------------------------------------------------------------- parity gate (same run) ----------------------------------
    parity = None
    branch_err = None
    em = job.fetch_edges()
    if e2e_stats is not None:       # the masks the TIMED end-to-end batches produced are the ones that are checked
        assert np.array_equal(e2e_em.eoff, em.eoff) and np.array_equal(e2e_em.rc, em.rc)
        assert np.array_equal(e2e_em.masked_adj, em.masked_adj) and np.array_equal(e2e_em.feat_mask, em.feat_mask), \
            "pipelined and resident-input runs of the same batch differ"
        em = e2e_em
    if wl.golden is not None and world == 1 and args.iters == int(wl.golden["epochs"]):
        z = wl.golden
        assert np.array_equal(em.eoff, z["eoff"]), "edge structure differs from the reference's sub-graphs"
        assert np.array_equal(dn.nb_flat.cpu().numpy()[:len(z["nb_flat"])], z["nb_flat"]), "k-hop lists differ from the reference's"
        import helpers
        # distance to the NEAREST legitimate outcome: the reference's output, or an alternate outcome the reference itself produces
        # under a 1-ulp perturbation of the initial mask (ReLU gates crossing zero within round-off of an iteration boundary
        # make the trajectory two-valued on a few targets; tests/golden/make_golden_branches.py)
        err, ferr, matched = helpers.branch_errors(z, helpers.load_branches(name), em.eoff, em.masked_adj,
                                                   1.0 / (1.0 + np.exp(-em.feat_mask.astype(np.float64))))
        well = (z["cond_mask"] <= WELL) & (z["cond_feat"] <= WELL)
        branch_err = (err, ferr)
        parity = {"reference": "outputs of /root/reference itself on every target (tests/golden/%s_full_explain.npz)" % name,
                  "targets": int(len(err)), "non_chaotic": int(well.sum()),
                  "matched_alternate_branch": int((matched[well] >= 0).sum()),
                  "max_abs_err": float(err[well].max()), "feat_max_abs_err": float(ferr[well].max()), "tolerance": PARITY_TOL,
                  "chaotic": {"targets": int((~well).sum()), "cpu_vs_cpu_max": float(z["cond_mask"].max()),
                                      "gpu_vs_reference_max": float(err[~well].max()) if (~well).any() else 0.0,
                                      "note": "targets on which the reference and the closed-form fp32 oracle (two CPU implementations) already "
                                              "differ by > 2e-6 after 300 epochs: Adam's scale-free step amplifies fp32 round-off wherever a "
                                              "gradient is ~0; reported, not gated"},
                  "khop_lists_bit_identical": True}
        s0_err, s0_ferr, _ = helpers.branch_errors(z, None, em.eoff, em.masked_adj, 1.0 / (1.0 + np.exp(-em.feat_mask.astype(np.float64))))
        calm = helpers.horizon_conditioning(name, z["cond_mask"], z["cond_feat"]) <= WELL
        ok, msg, unexplained = helpers.explained_outcome(name, "full", z["targets"], s0_err, s0_ferr, calm)
        parity["calm_targets"] = int(calm.sum())
        parity["unexplained_beyond_tolerance"] = unexplained
        inside = well & (err <= PARITY_TOL) & (ferr <= PARITY_TOL)
        # three numbers (VERDICT r2 "next" #1b): strict = against the reference's ONE output, no alternates; with the pre-declared
        # alternate set (24 one-ulp trials per target, tests/golden/make_golden_branches.py); and the ungated (chaotic) targets
        s_err, s_ferr, _ = helpers.branch_errors(z, None, em.eoff, em.masked_adj, 1.0 / (1.0 + np.exp(-em.feat_mask.astype(np.float64))))
        strict = (s_err <= PARITY_TOL) & (s_ferr <= PARITY_TOL)
        parity["three_numbers"] = {"strict_vs_reference_output": f"{int((strict & well).sum())} / {int(well.sum())} non-chaotic, {int(strict.sum())} / {len(strict)} of all targets",
                                   "with_predeclared_alternates": f"{int(inside.sum())} / {int(well.sum())} non-chaotic",
                                   "ungated_chaotic": f"{int((~well).sum())} targets, {int((strict & ~well).sum())} of them within 1e-5 anyway, worst {float(np.maximum(s_err, s_ferr)[~well].max()) if (~well).any() else 0.0:.2e} "
                                                      "(pinned window by window by tests/test_windowed_parity.py)"}
        parity.update(rule=msg, within_tolerance=int(inside.sum()),
                      beyond_tolerance=[{"target": int(z["targets"][k]), "mask": float(err[k]), "feat": float(ferr[k])} for k in np.nonzero(well & ~inside)[0]],
                      max_abs_err=float(err[inside].max()), feat_max_abs_err=float(ferr[inside].max()))
        if not ok and not args.no_parity_gate:
            raise SystemExit("PARITY FAILURE: " + json.dumps(parity))
        log(f"parity vs the reference's outputs: {parity['max_abs_err']:.2e} on {parity['within_tolerance']} of {parity['non_chaotic']} non-chaotic targets")

    out = None
    if rank == 0:
        route = job.route() if not args.no_resident else np.zeros(len(my_targets), np.int32)
        n2 = job.n.astype(np.float64) ** 2
        nnz = np.diff(em.eoff).astype(np.float64) * 2.0                 # directed edge entries per target
        kagg = job.D + 2 * job.H
        sum_n2 = job.sum_n2
        # ---- kernel timings: streaming kernels via gnnx_time_kernel, resident launches in situ (HIP events on their streams) ----
        launches = {}
        n_stream = int((route == 0).sum())
        per = []
        names = ["k_mask<true,true>", "k_conv<FWD1>", "k_conv<FWD2>", "k_node_head", "k_conv<BWD1>"]
        if n_stream:
            reps = 50 if n2[route == 0].sum() < 1e8 else 5
            per = [job.time_kernel(hy, k, reps) for k in (0, 1, 2, 3, 4)]
            launches["streaming"] = {"targets": n_stream, "ms_per_iter": sum(x[0] for x in per),
                                     "ms_total": sum(x[0] for x in per) * args.iters,
                                     "avg_launch_us": {nm: x[0] * 1e3 for nm, x in zip(names, per)}}
        rts = []
        for _ in range(5):
            job.reset_masks()
            job.launch(hy)
            torch.cuda.synchronize()
            rts.append(job.resident_times())
        rt = [float(np.mean(x)) for x in zip(*rts)]
        res_names = {1: "k_resident<1>", 4: "k_sparse_resident<.., 1024>", 5: "k_sparse_resident<.., 256>", 6: "k_sparse_resident<.., 64>",
                     7: "k_sparse_large", 8: "k_sparse_resident<.., 512>"}
        res_ms = {1: rt[0], 4: rt[3], 5: rt[4], 6: rt[5], 7: rt[6], 8: rt[7]}
        # ONE launch (k_sparse_resident_mixed) for the 512-thread targets, the single-tile targets (eight per workgroup) and - "pair" workgroups,
        # round 5 - the 256-thread targets two to a workgroup: it is timed in the slot of the class that starts it (512 threads, else 256)
        # packed single-wave launch (k_sparse_resident_tiny16 / 12, round 6): its targets and its time (reported in the 64-thread class's slot)
        pack_per_cu, n_packed = job.tiny_pack()
        packed = job.tiny_packed() if n_packed else np.zeros(len(route), bool)
        rt5_own = 0.0 if n_packed else rt[5]      # the 64-thread class's OWN launch (with a packed launch the slot holds the larger of the two; the rest of the class is small)
        pairs = bool((route == 5).any() and ((rt[7] and not rt[4]) or (rt[4] and not rt5_own and (route == 6).any() and not (route == 8).any())))
        mixed = bool((((route == 6) & ~packed).any() and (route == 8).any() and not rt5_own and rt[7]) or pairs)
        mixed_rv = 8 if rt[7] else 5
        if n_packed:
            res_names[6] = f"k_sparse_resident_tiny{pack_per_cu} ({pack_per_cu} single-wave targets per workgroup = per compute unit, slim LDS form)"
        tiny_per_wg = int(engine.get_library().gnnx_sparse_tiny_per_workgroup(int(job.D), int(job.H), int(job.C)))   # sp_mix_tiny() of gnnx_sparse.hpp
        if mixed:
            res_names[mixed_rv] = (f"k_sparse_resident_mixed (512-thread targets" + (", 256-thread targets two per workgroup" if pairs else "") +
                                   f" + {tiny_per_wg} single-tile targets per workgroup)")
        in_mixed = (route == 8) | ((route == 6) & ~packed) | ((route == 5) if pairs else np.zeros(len(route), bool))
        sel_of = {rv: in_mixed if (mixed and rv == mixed_rv) else (packed if (n_packed and rv == 6) else (route == rv)) for rv in res_ms}
        for rv, ms_v in res_ms.items():
            if ms_v:
                launches[res_names[rv]] = {"targets": int(sel_of[rv].sum()), "ms_total": ms_v}
        stream_total = launches.get("streaming", {}).get("ms_total", 0.0)
        top = max(res_ms, key=res_ms.get)
        if stream_total >= res_ms[top]:
            k = int(np.argmax([x[0] for x in per]))
            roof = {"kernel": names[k] + (" (fused sigmoid-mask + regulariser + Adam + G-tile MFMA)" if k == 0 else " (masked-adjacency contraction)"),
                    "bound": "hbm", "achieved": per[k][1] / (per[k][0] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": per[k][1] / (per[k][0] * 1e-3) / HBM_PEAK, "traffic": None, "avg_launch_us": per[k][0] * 1e3}
        else:
            # The dominant launch is an on-chip-resident kernel: all iterations in one launch, state in registers + LDS, the EDGE formulation
            # (only the mask entries on edges are live, rows beyond two hops pruned).  SURVEY.md section 8(d)'s dense figure does not bound it
            # (the same formula gave 0.42, 1.04 and 4.71 on three workloads); `frac` is the executed-work model of
            # gnn_model_explainer_amd/utils/work_model.py: the largest of three LOWER bounds on the launch time - executed flops and executed LDS
            # bytes on the busy CUs, and the critical chain of dependent operations of an iteration at the unloaded latencies measured by
            # tools/micro/chain_latency.hip - over the measured launch time.  The dense figure stays as `dense_equivalent`.
            from gnn_model_explainer_amd.utils import work_model as wm
            lat = wm.load_latency_table(ROOT)
            S_all = wm.target_structure(job.n, em.eoff, em.rc, dn.rows)
            xc = 2 if (job.D == 10 and job.H == 20 and bool(np.all(wl.feat == wl.feat[0]))) else 0   # gnnx_plan_analyze_features: the algebraic constant-feature form
            wgs_per_cu = {4: 1, 5: 2, 6: 6, 7: 1, 8: 1}
            # executed INSTRUCTIONS come from counters, not from a model: the committed PMC summary of this workload's loop (rocprofv3 --pmc SQ_INSTS_*;
            # tools/gpu_r5a.sh) - a fourth lower bound, VALU issue: a SIMD issues one wave64 VALU instruction per 2 cycles, one f32 32x32x2 MFMA per 64
            import glob
            tag = name + (f"_{args.targets}targets" if name == "ba100k" else "")
            cand_i = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{tag}_loop_only.json")))
            pmc_counts = json.load(open(cand_i[-1])).get("counters_mean_per_launch", {}) if cand_i else {}

            def model_of(rv):
                sel = sel_of[rv]
                idx = np.nonzero(sel)[0]
                Ssel = {kk: v[idx] for kk, v in S_all.items()}
                large = rv == 7
                fl = wm.executed_flops_per_iter(Ssel, job.D, job.H, job.H, job.C, xc)
                by = wm.executed_lds_bytes_per_iter(Ssel, job.D, job.H, job.H, job.C, xc, large=large)
                ops = wm.chain_ops_per_iter(Ssel, job.D, job.H, job.H, job.C, xc, large=large)
                ch = wm.chain_ns_per_iter(ops, lat)
                if mixed and rv == mixed_rv:      # workgroups [0, n_big): one 512-thread target each; then two 256-thread targets each; then tiny_per_wg single-tile targets each
                    r_sel = route[idx]
                    wg = np.zeros(len(idx), np.int64)
                    nb = int((r_sel == 8).sum())
                    wg[r_sel == 8] = np.arange(nb)
                    np_ = int((r_sel == 5).sum())
                    wg[r_sel == 5] = nb + np.arange(np_) // 2
                    wg[r_sel == 6] = nb + (np_ + 1) // 2 + np.arange(int((r_sel == 6).sum())) // tiny_per_wg
                elif n_packed and rv == 6:
                    wg = np.arange(len(idx)) // pack_per_cu
                else:
                    wg = np.arange(len(idx))
                b = wm.launch_bounds(fl, by, ch, args.iters, wg, 1 if (n_packed and rv == 6) else wgs_per_cu.get(rv, 1))
                ms_meas = res_ms[rv]
                t = {"flops": b["flops_s"], "lds": b["lds_s"], "chain": b["chain_s"]}
                cnt = pmc_counts.get(res_names[rv].split(" ")[0].split("<")[0], {})
                issue = None
                if cnt.get("SQ_INSTS_VALU"):
                    t["issue"] = (cnt["SQ_INSTS_VALU"] * 2.0) / (b["busy_cus"] * 4.0) / wm.CLOCK_HZ
                    issue = {"valu_instructions_per_launch": cnt["SQ_INSTS_VALU"], "salu": cnt.get("SQ_INSTS_SALU"), "lds": cnt.get("SQ_INSTS_LDS"),
                             "mfma_mops_f32_raw": cnt.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "cycles_per_valu_instruction_and_simd": 2.0,
                             "source": os.path.relpath(cand_i[-1], ROOT),
                             "note": "wave-level instruction counts of the same command's loop (PMC); bound = VALU instructions x 2 cycles / (busy CUs x 4 SIMDs)"}
                bound = max(t, key=t.get)
                out_m = {"kernel": res_names[rv], "targets": int(sel.sum()), "workgroups": b["workgroups"], "busy_cus": b["busy_cus"], "measured_ms": ms_meas,
                         "bound": bound, "frac": t[bound] / (ms_meas * 1e-3),
                         "lower_bounds_ms": {kk: v * 1e3 for kk, v in t.items()},
                         "frac_by_bound": {kk: v / (ms_meas * 1e-3) for kk, v in t.items()},
                         "executed": {"tflops": b["executed_flops"] / (ms_meas * 1e-3) / 1e12, "flops_per_launch": b["executed_flops"],
                                      "lds_GBps": b["executed_lds_bytes"] / (ms_meas * 1e-3) / 1e9, "lds_bytes_per_launch": b["executed_lds_bytes"],
                                      "flop_peak_busy_cus_TFLOPs": b["busy_cus"] * wm.CU_F32_FLOPS / 1e12,
                                      "lds_peak_busy_cus_GBps": b["busy_cus"] * wm.CU_LDS_BPS / 1e9},
                         "issue": issue,
                         "chain": {"ns_per_iteration_slowest_target": b["chain_ns_per_iter_slowest"], "ns_per_iteration_mean": b["chain_ns_per_iter_mean"],
                                   "measured_ns_per_iteration": ms_meas * 1e6 / args.iters,
                                   "dependent_ops_per_iteration_slowest_target": {kk: float(v[int(np.argmax(ch))]) for kk, v in ops.items() if float(v.max()) > 0},
                                   "latency_ns": {kk: lat[kk] for kk in ("lds", "l2", "shuffle", "dpp", "fma", "mfma", "transc", "handover", "barrier")},
                                   "latency_source": lat["source"],
                                   "regime": "critical path of the slowest target (workgroups <= CUs x workgroups per CU)" if b["workgroups"] <= NUM_CUS * wgs_per_cu.get(rv, 1)
                                             else "saturated: sum of the chains / (CUs x workgroups per CU)"}}
                if "loaded" in lat:      # the same chain at the latencies measured with two waves per SIMD on every CU: where the time goes, not a bound
                    lat2 = dict(lat)
                    lat2.update(lat["loaded"])
                    ch2 = wm.chain_ns_per_iter(ops, lat2)
                    b2 = wm.launch_bounds(fl, by, ch2, args.iters, wg, wgs_per_cu.get(rv, 1))
                    out_m["chain"]["at_two_waves_per_simd"] = {"ns_per_iteration_slowest_target": b2["chain_ns_per_iter_slowest"], "chain_ms": b2["chain_s"] * 1e3,
                                                               "frac": b2["chain_s"] / (ms_meas * 1e-3)}
                f_alg = 6.0 * n2[sel].sum() * kagg * args.iters        # F_alg = 6 n^2 (D + 2H) flop per target and iteration
                b_alg = 28.0 * n2[sel].sum() * args.iters              #                          B_alg = 28 n^2 bytes
                out_m["dense_equivalent"] = {"note": "SURVEY.md section 8(d) prices the reference's DENSE formulation; this kernel does not execute it (edge entries are "
                                                     "%.2f %% of n^2 here), so these are speed-ups over a dense implementation at peak, not utilisations" % (100.0 * nnz[sel].sum() / max(1.0, n2[sel].sum())),
                                             "mfma_frac": f_alg / (ms_meas * 1e-3) / MFMA_F32_PEAK, "hbm_frac": b_alg / (ms_meas * 1e-3) / HBM_PEAK,
                                             "alg_flops_per_launch": f_alg, "alg_bytes_per_launch": b_alg}
                return out_m
            models = {rv: model_of(rv) for rv, ms_v in res_ms.items() if ms_v and sel_of[rv].any()}
            m = models[top]
            ms = res_ms[top]
            n_wg = m["workgroups"]
            roof = {"kernel": res_names[top] + " (edge-sparse on-chip-resident optimisation, one workgroup per target, all iterations in one launch)",
                    "bound": m["bound"], "achieved": args.iters / (ms * 1e-3), "peak": args.iters / (m["lower_bounds_ms"][m["bound"]] * 1e-3),
                    "unit": "iterations/s of the launch", "frac": m["frac"], "traffic": None, "avg_launch_us": ms * 1e3,
                    "definition": "executed-work model (gnn_model_explainer_amd/utils/work_model.py): frac = max(executed flops / f32 rate of the busy CUs, executed LDS "
                                  "bytes / LDS rate of the busy CUs, critical chain of dependent operations x unloaded measured latencies) / measured launch time "
                                  "(HIP events on the launch stream, in situ); each term is a lower bound on the launch, so frac <= 1 and 1 - frac is what is lost to "
                                  "instruction issue, loaded latency, bank conflicts and barrier skew",
                    "model": m, "workgroups": n_wg, "cus": NUM_CUS,
                    "other_resident_launches": {res_names[rv]: {kk: mv[kk] for kk in ("targets", "workgroups", "measured_ms", "bound", "frac", "frac_by_bound")}
                                                for rv, mv in models.items() if rv != top}}
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{name}*.json")))      # (the newest round's summary of this workload)
        pmc = cand[-1] if cand else ""
        if pmc:   # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc, committed summary)
            try:
                roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch", {}).get(roof["kernel"].split(" ")[0].split("<")[0])
                roof["traffic_source"] = os.path.relpath(pmc, ROOT)
            except Exception:
                pass
        # ---- what the hardware saw: utilisation of the launch's compute units from the PMC summary of the same command's loop ----
        cand_u = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{name}" + (f"_{args.targets}targets" if name == "ba100k" else "") + "_loop_only.json")))
        if cand_u and "model" in roof:
            try:
                ps = json.load(open(cand_u[-1]))
                kname = roof["kernel"].split(" ")[0].split("<")[0]
                cs = ps.get("counters_mean_per_launch", {}).get(kname, {})
                # the summary's own launch duration (rocprofv3 kernel trace of the same passes) if it carries one, else this run's isolated launch
                prof_ms = ps.get("avg_launch_ms", {}).get(kname) or roof["avg_launch_us"] / 1e3
                busy = roof["model"]["busy_cus"]
                cyc = prof_ms * 1e-3 * wm.CLOCK_HZ                       # shader cycles of one launch
                hu = {"source": os.path.relpath(cand_u[-1], ROOT), "kernel": kname, "launch_ms_of_the_counter_passes": prof_ms, "busy_cus": busy,
                      "note": "fractions of what the BUSY compute units could issue / move during the launch (a launch of W workgroups occupies W of 256 CUs); "
                              "VALU: wave64 instruction = 2 cycles of a SIMD-32; MFMA: SQ_VALU_MFMA_BUSY_CYCLES per SIMD; LDS: SQ_LDS_IDX_ACTIVE cycles of the CU's "
                              "LDS pipe; HBM: (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md) against 8 TB/s; waves: SQ_WAIT_ANY, "
                              "SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles both)"}
                if cs.get("SQ_INSTS_VALU"):
                    hu["valu_issue_frac"] = cs["SQ_INSTS_VALU"] * 2.0 / (busy * 4.0 * cyc)
                if cs.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    hu["mfma_busy_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (busy * 4.0 * cyc)
                if cs.get("SQ_LDS_IDX_ACTIVE"):
                    hu["lds_pipe_frac"] = cs["SQ_LDS_IDX_ACTIVE"] / (busy * cyc)
                    hu["lds_bank_conflict_frac_of_lds_cycles"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
                hb = ps.get("hbm_bytes_per_launch", {}).get(kname)
                if hb:
                    hu["hbm_frac_of_8TBps"] = hb / (prof_ms * 1e-3) / HBM_PEAK
                if cs.get("SQ_WAVE_CYCLES"):
                    hu["waves_waiting_frac"] = cs.get("SQ_WAIT_ANY", 0.0) / cs["SQ_WAVE_CYCLES"]
                    hu["waves_issuing_frac"] = cs.get("SQ_ACTIVE_INST_ANY", 0.0) / cs["SQ_WAVE_CYCLES"]
                roof["hw_utilisation"] = hu
            except Exception as e:      # a summary of another layout: say so instead of failing the line
                roof["hw_utilisation"] = {"error": repr(e)}
        # ---- the same kernel INSIDE the pipeline (three launches share the chip): mean launch time of the timed batches and the model's frac at it ----
        if e2e_stats is not None and e2e_stats.get("launch_in_pipeline_ms") and "model" in roof:
            lp = float(e2e_stats["launch_in_pipeline_ms"])
            m_ = roof["model"]
            roof["in_pipeline"] = {"mean_launch_ms": lp, "isolated_launch_ms": roof["avg_launch_us"] / 1e3,
                                   "frac": m_["lower_bounds_ms"][m_["bound"]] / lp,
                                   "note": "device time of the dominant launch while the batches ahead and behind share the chip (HIP events on the launch stream of every "
                                           "timed batch, pipeline.py); `frac` above is for the isolated launch"}
        roof["launches"] = launches
        roof["whole_job"] = {"sum_n2": sum_n2, "directed_edge_entries": float(nnz.sum()),
                             "dense_alg_flops_per_step": 6.0 * sum_n2 * kagg * args.iters, "dense_alg_bytes_per_step": 28.0 * sum_n2 * args.iters,
                             "loop_wall_ms_per_iter": loop_only["ms_per_step"] / args.iters}
        out = {"metric": "explained nodes/sec (300 mask-opt iters, k-hop subgraph)", "value": value,
               "unit": "explained nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
               "vs_baseline": None, "dtype": "f32",
               "data": "synthetic (BA-House graphs; GCN trained by the reference train.py on syn1, fixture tests/golden/syn1_ckpt.npz)",
               "config": {"workload": wl.desc + f", 3-hop sub-graphs, {args.iters} iters, Adam lr 0.1",
                          "targets_total": n_targets, "targets_rank0": len(my_targets), "sum_n2_rank0": sum_n2,
                          "launch": "plain" if args.no_graph else "hipGraph", "resident_path": not args.no_resident,
                          "routing_rank0": {"streaming": int((route == 0).sum()), "dense_resident": int(((route >= 1) & (route <= 3)).sum()),
                                            "sparse_resident": int(((route >= 4) & (route != 7)).sum()), "sparse_large": int((route == 7).sum())},
                          "parallelism": (f"target-sharded x{world}: LPT on the per-class GPU-time model (parallel.target_cost), every rank runs its shard END TO END (k-hop -> D2H) through the pipeline, masks all-gathered as edge entries over RCCL inside the timed region"
                                          if world > 1 else "single GPU")},
               "roofline": roof}
        out["loop_only"] = loop_only
        if e2e_stats is not None:
            out["value_definition"] = ("STEADY STATE of K identical batched jobs: targets x K / wall time, every job the whole hot path of SURVEY.md section 8(d) with only the "
                                       "graph resident - device k-hop, plan, device-side packing, routing, seeded initial masks, H2D + scatter, the 300 iterations, gather + "
                                       "D2H of the masks - through pipeline.BatchPipeline (%d prepare workers, up to %d optimisations sharing the chip, one fetch stream), "
                                       "fill and drain inside the timed region, median of the repetitions.  ONE job alone, end to end: `pcie_inclusive` "
                                       "(%.0f nodes/s here); the optimisation alone on resident inputs: `loop_only`"
                                       % (e2e_stats.get("prepare_workers", 0), e2e_stats.get("optimisations_in_flight", 0), 0.0))
            out["end_to_end_stage_ms"] = e2e_stats
        if parity is not None:
            out["parity"] = parity
        if name == "ba100k":      # no full-size fixture of the live reference exists for this graph (its dense 100k x 100k neighbourhood matrix): say what is compared
            out.setdefault("parity", {})["coverage"] = ("of this workload's targets, 39 + 13 route-stratified ones (n = 6 ... 8297) are compared with the LIVE reference "
                                                        "(tests/golden/ba100k_windows.npz: its Adam state every 50 epochs and every decision of every epoch; "
                                                        "ba100k_explain.npz: its outputs) by tests/test_decision_parity.py and tests/test_gpu_full_configs.py; in this "
                                                        "run the CPU oracle checks the cpu_baseline sample (`vs_cpu_oracle`, when that leg runs)")
        log("kernel timings done")
        step_s = dt / args.steps
        pipe = {k: e2e[k] for k in ("khop_device_ms", "plan_pack_analyze_ms", "host_rng_ms", "mask_h2d_scatter_ms", "first_run_ms", "edges_d2h_ms")}
        steady = e2e["warm"]["total_ms"] if "warm" in e2e else (pipe["khop_device_ms"] + pipe["plan_pack_analyze_ms"] + pipe["host_rng_ms"] +
                                                                 pipe["mask_h2d_scatter_ms"] + step_s * 1e3 + pipe["edges_d2h_ms"])
        out["pcie_inclusive"] = {"value": len(my_targets) / (steady * 1e-3), "unit": "explained nodes/s", "batch_total_ms": steady,
                                 "warm_batch": e2e.get("warm"), "first_batch": pipe, "gpu_ms": step_s * 1e3,
                                 "first_batch_total_ms": e2e["total_ms"], "host_rng_threads": e2e.get("host_rng_threads"),
                                 "note": "one batch end to end on rank 0 (wall clock of the second, warm batch), graph resident: k-hop walk sets on the "
                                         "device (gnnx_khop), plan + device-side packing + routing, host RNG of the initial masks (seed protocol, private "
                                         "generators), one pinned H2D copy + gnnx_scatter_masks, the 300-iteration optimisation, edge-list D2H "
                                         "(gnnx_gather_edges); the first batch additionally pays one-time code-object loads and allocator warm-up; "
                                         "never used as `value`"}
        if "value_definition" in out:
            out["value_definition"] = out["value_definition"].replace("(0 nodes/s here)", "(%.0f nodes/s here)" % out["pcie_inclusive"]["value"])
    if world > 1:
        # per-rank load (sum n^2) and, on rank 0, the SAME workload on one GPU (for the scaling denominator)
        loads = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(loads, torch.tensor([job.sum_n2], device=dev, dtype=torch.float64))
        costs = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(costs, torch.tensor([float(parallel.target_cost(job.n, cost_table).sum())], device=dev, dtype=torch.float64))
        if rank == 0:
            out["config"]["sum_n2_per_rank"] = [float(x.item()) for x in loads]
            out["config"]["modelled_gpu_us_per_rank"] = [float(x.item()) for x in costs]
            out["config"]["cost_table_us"] = {"classes": "n <= 32 / 128 / 512 / 16383", "measured_on_rank0": cost_table.tolist(),
                                             "default": parallel.DEFAULT_COST_TABLE.tolist()}
            out["config"]["gathered_edge_entries_per_rank"] = gather_bufs["counts"]
        job.close()
        del job
        if rank == 0 and not args.no_single_gpu_leg:
            # the same fixed target set on rank 0 alone, the same way the N = 1 line of this file measures it: end to end through the pipeline
            torch.cuda.empty_cache()
            from gnn_model_explainer_amd.pipeline import BatchPipeline
            pipe1 = BatchPipeline(graph, wl.ck["sd"], wl.label, hy, rng_threads_big=rng_threads_for(1e9, 1))
            for _ in pipe1.run([wl.targets]):
                pass
            k1 = max(1, min(3, args.steps))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe1.run([wl.targets] * k1):
                pass
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t0) / k1
            out["single_gpu_same_workload"] = {"value": n_targets / d1, "unit": "explained nodes/s", "ms_per_step": d1 * 1e3,
                                               "note": "the whole fixed target set on rank 0 alone, end to end through the same pipeline, after the timed region "
                                                       "(the other ranks idle: the host's cores are not shared)"}
        dist.barrier()
    elif not args.no_cpu_baseline:
        # CPU baseline on a size-stratified sample of the same targets; the oracle's masks are compared with the GPU's
        lists = dn.lists()
        order = np.argsort(dn.sizes, kind="stable")
        pick = order[np.linspace(0, len(order) - 1, 32).astype(int)]
        import helpers
        sample = [(int(k), wl.dense_subgraph(int(my_targets[k]), lists[k], dn.rows[k], helpers.seeded_mask0(int(my_targets[k]), int(dn.sizes[k])).numpy()))
                  for k in pick]
        base, one, res = cpu_baselines(wl, sample, args.iters)
        out["cpu_baseline"], out["cpu_baseline_1thread"] = base, one
        import helpers
        errs = []
        for k, (ma, fs) in res.items():
            a, b = em.eoff[k], em.eoff[k + 1]
            r, c = em.rc[a:b, 0], em.rc[a:b, 1]
            e1 = float(np.abs(ma[r, c] - em.masked_adj[a:b]).max()) if b > a else 0.0
            e2 = float(np.abs(fs - 1.0 / (1.0 + np.exp(-em.feat_mask[k].astype(np.float64)))).max())
            if branch_err is not None and max(e1, e2) > PARITY_TOL:      # a two-valued target: distance to the nearest branch
                e1, e2 = min(e1, float(branch_err[0][k])), min(e2, float(branch_err[1][k]))
            errs.append((k, e1, e2))
        wellk = [e for e in errs if wl.golden is None or (wl.golden["cond_mask"][e[0]] <= WELL and wl.golden["cond_feat"][e[0]] <= WELL)]
        vs = {"targets": len(errs), "non_chaotic": len(wellk), "max_abs_err": max(e[1] for e in wellk),
              "feat_max_abs_err": max(e[2] for e in wellk), "tolerance": PARITY_TOL,
              "note": "GPU masks vs the CPU oracle's on the cpu_baseline sample, computed in this run"}
        out.setdefault("parity", {})["vs_cpu_oracle"] = vs
        if (vs["max_abs_err"] > helpers.BRANCH_JUMP_MAX or vs["feat_max_abs_err"] > helpers.BRANCH_JUMP_MAX or
                sum(1 for e in wellk if max(e[1], e[2]) > PARITY_TOL) > max(1, len(wellk) // 16)) and not args.no_parity_gate:
            raise SystemExit("PARITY FAILURE vs the CPU oracle: " + json.dumps(vs))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
