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
