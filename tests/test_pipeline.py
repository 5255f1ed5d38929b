"""pipeline.BatchPipeline: batches of targets through the whole hot path with the stages overlapped (prepare / optimise / fetch on
three streams) must return, batch by batch and in order, exactly what the sequential device-side path returns."""
import numpy as np
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob
from gnn_model_explainer_amd.pipeline import BatchPipeline
from gnn_model_explainer_amd.utils.graph_utils import KHopIndex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,edge_draw_min,device_walk", [("syn4", 2e7, None), ("syn1", 2e7, None), ("syn1", 0.0, False), ("syn1", 0.0, True)])
def test_pipelined_batches_equal_sequential_jobs(name, edge_draw_min, device_walk):
    """(edge_draw_min = 0: the host keeps only the values on the edges of its draw - gnnx_host_draw_edge_masks, the path of the large
    BA-House x100k batches - and the results must still be the sequential path's bit for bit; device_walk: the engine of that draw on the
    host (a single process's default) or on the device (gnnx_mt_edge_words: the default of a node's ranks))"""
    ck = helpers.load_ckpt(name)
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])
    first = 300 if name == "syn1" else 511
    rng = np.random.default_rng(3)
    batches = [np.sort(rng.choice(np.arange(first, ck["num_nodes"]), k, replace=False)) for k in (7, 64, 1, 33, 64)]   # ragged, one repeated size
    hy = Hyper(num_iters=25)
    pipe = BatchPipeline(graph, ck["sd"], ck["label"], hy, rng_threads=4, edge_draw=True, edge_draw_min_values=edge_draw_min, device_walk=device_walk)
    got = list(pipe.run(batches))
    assert len(got) == len(batches) and len(pipe.stats) == len(batches)
    assert all(("host_rng_edges_only" in st) == (edge_draw_min == 0.0) for st in pipe.stats)
    assert all(("host_rng_device_walk" in st) == bool(device_walk and edge_draw_min == 0.0 and engine.pair_staging_ok()) for st in pipe.stats)
    for targets, em in zip(batches, got):
        dn = engine.khop_device(graph, targets, 3)
        job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"])
        job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
        job.launch(hy)
        want = job.fetch_edges()
        assert np.array_equal(em.eoff, want.eoff) and np.array_equal(em.rc, want.rc)
        assert np.array_equal(em.masked_adj, want.masked_adj) and np.array_equal(em.feat_mask, want.feat_mask)
        assert np.array_equal(em.n, dn.sizes)


def test_pipeline_reports_a_bad_target():
    """An isolated node has an empty walk set (the reference fails on it too): the error reaches the caller, nothing hangs."""
    import scipy.sparse as sp
    ck = helpers.load_ckpt("syn4")
    n = ck["num_nodes"] + 1                      # one extra, isolated node
    e = ck["edges"]
    csr = sp.csr_matrix((np.ones(2 * len(e), np.float32), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n))
    feat = np.vstack([ck["feat"], ck["feat"][:1]])
    pred = np.vstack([ck["pred"], ck["pred"][:1]])
    graph = engine.device_graph(csr, feat, pred)
    pipe = BatchPipeline(graph, ck["sd"], np.r_[ck["label"], 0], Hyper(num_iters=3))
    with pytest.raises(Exception):
        list(pipe.run([np.asarray([600, 601]), np.asarray([n - 1])]))


def test_pipeline_mixed_batches_dense_and_xl_parts():
    """Batches whose larger targets take the XL route (engine.XLJob: sub-graph CSRs from the resident graph, edge-list state) while the others are packed
    for the resident classes: the merged edge lists come back in TARGET order, each part bit-identical to its own sequential job, with the device and
    the host engine walk."""
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])
    rng = np.random.default_rng(8)
    batches = [np.sort(rng.choice(np.arange(300, 700), k, replace=False)) for k in (40, 9, 64)]
    batches.append(np.asarray([300, 301]))        # n = 310 twice: an all-XL batch
    hy = Hyper(num_iters=25)
    for dw in (False, True):
        pipe = BatchPipeline(graph, ck["sd"], ck["label"], hy, rng_threads=4, xl_min_n=100, xl_device_walk=dw)
        got = list(pipe.run(batches))
        for targets, em in zip(batches, got):
            dn = engine.khop_device(graph, targets, 3)
            assert np.array_equal(em.n, dn.sizes)
            big = dn.sizes > 100
            assert dict(em.routes).get("xl", 0) == int(big.sum()) and dict(em.routes).get("dense", 0) == int((~big).sum())
            for sel, xl in ((~big, False), (big, True)):
                if not sel.any():
                    continue
                sub = targets[sel]
                dns = engine.khop_device(graph, sub, 3)
                if xl:
                    job = engine.XLJob(graph, dns, None, ck["label"][sub], ck["sd"])
                    job.set_masks_seeded(1000 + sub, threads=2)
                else:
                    job = MaskOptimJob.from_csr(graph, dns, None, ck["label"][sub], ck["sd"])
                    job.set_masks_raw(engine.init_edge_masks_raw(dns.sizes, seeds=1000 + sub))
                job.launch(hy)
                want = job.fetch_edges()
                for j, k in enumerate(np.nonzero(sel)[0]):
                    a, b, c, d = int(em.eoff[k]), int(em.eoff[k + 1]), int(want.eoff[j]), int(want.eoff[j + 1])
                    assert b - a == d - c and np.array_equal(em.rc[a:b], want.rc[c:d])
                    assert np.array_equal(em.masked_adj[a:b], want.masked_adj[c:d]) and np.array_equal(em.feat_mask[k], want.feat_mask[j])
            # and against the all-dense routes (other kernels: round-off apart)
            job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"])
            job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
            job.launch(hy)
            ref = job.fetch_edges()
            assert np.array_equal(em.eoff, ref.eoff) and np.array_equal(em.rc, ref.rc)
            assert np.abs(em.masked_adj - ref.masked_adj).max() < 2e-5
