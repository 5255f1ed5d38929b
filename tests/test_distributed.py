"""world_size-2 gloo test of the target-sharding path (runs on CPU; each rank computes its shard with the
emulator build of the HIP sources, i.e. the same code path as on 2 GPUs minus the device)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.emu_engine import emu_job
    from gnn_model_explainer_amd import parallel
    from gnn_model_explainer_amd.engine import Hyper, Subgraph
    ck, gx = helpers.load_ckpt("syn4"), helpers.load_explain("syn4")
    targets = [int(t) for t in gx["targets"]]
    costs = [len(gx[f"{t}:neighbors"]) ** 2 for t in targets]

    def compute(ts):
        subs = []
        for t in ts:
            nb = gx[f"{t}:neighbors"]
            A, X, lab, yhat = helpers.subgraph(ck, nb)
            new = int(gx[f"{t}:node_idx_new"])
            subs.append(Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy()))
        return emu_job(subs, ck["sd"]).run([s.mask0 for s in subs], Hyper(num_iters=6)).masked_adj

    res = parallel.run_sharded(targets, costs, compute)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(k): v for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path):
    world, port = 2, 29531 + os.getpid() % 500
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(r0.files) == sorted(r1.files) and len(r0.files) == 4
    from emu.emu_engine import emu_job
    from gnn_model_explainer_amd.engine import Hyper, Subgraph
    ck, gx = helpers.load_ckpt("syn4"), helpers.load_explain("syn4")
    for t in gx["targets"]:
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        s = Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(int(t), len(nb)).numpy())
        solo = emu_job([s], ck["sd"]).run([s.mask0], Hyper(num_iters=6)).masked_adj[0]
        assert np.array_equal(r0[str(t)], solo) and np.array_equal(r1[str(t)], solo)   # sharding changes no bit


def _api_worker(rank, world, port, out_dir):
    """The reference-shaped API under torch.distributed: Explainer.explain_nodes shards the targets over the ranks
    (parallel.lpt_shards on parallel.target_cost) and returns the FULL list on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from emu.emu_engine import emu_library
    from gnn_model_explainer_amd import models
    from gnn_model_explainer_amd.explainer import explain
    explain._ENGINE.update(lib=emu_library(), device="cpu")
    ck = helpers.load_ckpt("syn1")
    args = argparse.Namespace(logdir=out_dir, ckptdir=out_dir, dataset="syn1", bmname=None, opt="adam", opt_scheduler="none",
                              lr=0.1, num_epochs=4, hidden_dim=20, output_dim=20, num_gc_layers=3, method="base", name_suffix="",
                              explainer_suffix="r%d" % rank, graph_idx=-1, mask_act="sigmoid", mask_bias=False, bn=False, bias=True, gpu=True)
    model = models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args)
    model.load_state_dict({k: torch.tensor(v) for k, v in ck["sd"].items()})
    ex = explain.Explainer(model, ck["adj"][None].astype(np.float64), ck["feat"][None].astype(np.float64), ck["label"][None],
                           ck["pred"][None], None, args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    torch.manual_seed(11)
    out = ex.explain_nodes([302, 555, 309, 640, 699], args)
    np.savez(os.path.join(out_dir, f"api_rank{rank}.npz"), *out)
    dist.barrier()
    dist.destroy_process_group()


def test_explainer_api_shards_targets_over_ranks(tmp_path):
    world, port = 2, 30131 + os.getpid() % 500
    mp.spawn(_api_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "api_rank0.npz"), np.load(tmp_path / "api_rank1.npz")
    # the single-process answer with the same RNG stream
    import argparse
    from emu.emu_engine import emu_library
    from gnn_model_explainer_amd import models
    from gnn_model_explainer_amd.explainer import explain
    explain._ENGINE.update(lib=emu_library(), device="cpu")
    try:
        ck = helpers.load_ckpt("syn1")
        args = argparse.Namespace(logdir=str(tmp_path), ckptdir=str(tmp_path), dataset="syn1", bmname=None, opt="adam",
                                  opt_scheduler="none", lr=0.1, num_epochs=4, hidden_dim=20, output_dim=20, num_gc_layers=3,
                                  method="base", name_suffix="", explainer_suffix="solo", graph_idx=-1, mask_act="sigmoid",
                                  mask_bias=False, bn=False, bias=True, gpu=True)
        model = models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args)
        model.load_state_dict({k: torch.tensor(v) for k, v in ck["sd"].items()})
        ex = explain.Explainer(model, ck["adj"][None].astype(np.float64), ck["feat"][None].astype(np.float64), ck["label"][None],
                               ck["pred"][None], None, args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
        torch.manual_seed(11)
        solo = ex.explain_nodes([302, 555, 309, 640, 699], args)
    finally:
        explain._ENGINE.update(lib=None, device=None)
    assert len(r0.files) == 5
    for k, want in enumerate(solo):
        assert np.array_equal(r0[f"arr_{k}"], want) and np.array_equal(r1[f"arr_{k}"], want)      # sharding changes no bit


def test_lpt_shards_of_the_all_node_sample_with_the_giant_tail():
    """The --gpus N workload (bench.py --workload ba100k-all): 8 448 targets stratified over ALL nodes of BA-House x100k, 512 of them beyond 16 383
    sub-graph nodes, the largest 47 913 (tests/golden/ba100k_all_sample.npy: node ids and sub-graph sizes as the GPU run's device k-hop pass found
    them).  Cut by parallel.target_cost the shards of 2 / 4 / 8 ranks stay within 5 % modelled cost with that tail present (VERDICT r5 "next" 8),
    every target lands on exactly one rank, and the single largest target stays below a rank's share at 8 GPUs in the MODEL - its 0.5 s chain on one
    compute unit is the critical path there all the same."""
    import os as _os
    from gnn_model_explainer_amd import parallel
    z = np.load(_os.path.join(helpers.GOLDEN, "ba100k_all_sample.npy"))
    ids, sizes = z[0], z[1]
    assert len(ids) >= 8000 and (sizes > 16383).sum() >= 256 and sizes.max() > 40000 and (sizes <= 32).sum() >= 1000
    cost = parallel.target_cost(sizes)
    assert np.all(np.diff(cost[np.argsort(sizes, kind="stable")]) >= -1e-9)
    for world in (2, 4, 8):
        shards = parallel.lpt_shards(cost, world)
        assert sorted(i for s in shards for i in s) == list(range(len(ids)))
        load = np.asarray([cost[np.asarray(s, np.int64)].sum() for s in shards])
        assert (load.max() - load.min()) / load.max() < 0.05, (world, load)
        assert cost.max() < load.min()
