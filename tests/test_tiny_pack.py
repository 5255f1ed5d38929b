"""The packed single-wave launch (k_sparse_resident_tiny16 / 12, csrc/gnnx_sparse.hpp; round 6): targets of the 64-thread class whose slim LDS
form fits a slice run sixteen (twelve) to a compute unit in a launch of their own.  It is the SAME body as the class's other launches (the
mixed launch's single-wave slices, the class's own 64-thread launch) over another LDS layout and under another register cap, so its results must
equal theirs BIT FOR BIT - on the emulator (layout, routing, the id lists of the two launches) and on the GPU (the capped build: spills, LDS
races, co-resident workgroups).  The loop it runs: explain.py:137-146."""
import numpy as np
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, Subgraph
from oracle import closed_form
from test_emu_kernels import _Backend


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def _subs(rng, sizes, density=0.2):
    """constant feature rows (the reference's synthetic datasets, gengraph.py:60-61): the plan then runs the algebraic form"""
    row = rng.uniform(0.2, 1.5, 10).astype(np.float32)
    out = []
    for n, t in sizes:
        d = density if n <= 32 else min(density, 4.0 / n)
        A, X = helpers.random_graph(rng, n, 10, density=d)
        X = np.tile(row, (n, 1)).astype(np.float32)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        out.append(Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0))
    return out


@pytest.fixture(autouse=True)
def _pack_small_batches(monkeypatch):
    """the plan packs only batches that saturate the chip (>= 2048 single-wave targets); the tests' batches are small"""
    monkeypatch.setenv("GNNX_TINY_PACK_MIN", "1")


def _run(be, subs, sd, iters):
    job = be.job(subs, sd)
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=iters))
    return job, res


@pytest.mark.parametrize("per_cu", [16, 12])
def test_packed_launch_beside_the_mixed_launch_is_bit_identical(be, monkeypatch, per_cu):
    """A 512-thread target, a pair workgroup and 21 single-wave targets, some of them too large for a 16-per-CU slice: the packed launch takes most, the mixed
    launch the rest; every target equals the run without the packed launch (GNNX_TINY_PACK=0) bit for bit, and the closed form."""
    rng = np.random.default_rng(31)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    sizes = [(200, 3), (70, 5), (50, 0)] + [(int(rng.integers(6, 29)), 2) for _ in range(20)] + [(32, 1)]
    subs = _subs(rng, sizes)
    subs[-1] = _subs(rng, [(32, 1)], density=0.12)[0]      # ~87 edges on 32 nodes: beyond a 16-per-CU slice (41 edges at that size), within the class (128)
    monkeypatch.setenv("GNNX_TINY_PACK", str(per_cu))
    job, res = _run(be, subs, sd, 4)
    assert list(job.route()) == [8, 5, 5] + [6] * 21
    got_per_cu, n_packed = job.tiny_pack()
    assert got_per_cu == per_cu and (15 <= n_packed < 21 if per_cu == 16 else n_packed == 21), (got_per_cu, n_packed)      # (these random graphs are denser than k-hop sub-graphs: a few exceed a slice)
    monkeypatch.setenv("GNNX_TINY_PACK", "0")
    job0, res0 = _run(be, subs, sd, 4)
    assert job0.tiny_pack() == (0, 0) and list(job0.route()) == list(job.route())
    for i, s in enumerate(subs):
        assert np.array_equal(res.masked_adj[i], res0.masked_adj[i]), i
        assert np.array_equal(res.feat_mask[i], res0.feat_mask[i]), i
        o = closed_form.ClosedFormOracle(s.adj, s.feat, sd, s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(4)).max() < 2e-5, i      # (the algebraic form rounds differently from the closed form: sanity only - the gate is the bit identity above)


def test_a_batch_of_single_wave_targets_only(be, monkeypatch):
    """No larger target: the packed launch is the batch's ONLY launch (19 targets: two full workgroups and a partial one whose idle waves leave);
    equal to the 64-thread class's own launch bit for bit; a resumed run (gnnx_run_resume: moments + feature-mask state in) too."""
    rng = np.random.default_rng(32)
    sd = helpers.random_model(rng, 10, 20, 20, 2)
    subs = _subs(rng, [(int(rng.integers(5, 30)), 0) for _ in range(19)], density=0.1)
    job, res = _run(be, subs, sd, 6)
    assert list(job.route()) == [6] * 19 and job.tiny_pack()[0] == 16 and job.tiny_pack()[1] >= 15
    monkeypatch.setenv("GNNX_TINY_PACK", "0")
    job0, res0 = _run(be, subs, sd, 6)
    for i in range(len(subs)):
        assert np.array_equal(res.masked_adj[i], res0.masked_adj[i]) and np.array_equal(res.feat_mask[i], res0.feat_mask[i]), i
    monkeypatch.delenv("GNNX_TINY_PACK")
    # edgeless and two-node targets keep working (degenerate sub-graphs reach the class too)
    iso = _subs(rng, [(6, 0), (9, 3)])
    iso[0].adj[:] = 0
    job2, res2 = _run(be, iso, sd, 3)
    if job2.tiny_pack()[1]:
        monkeypatch.setenv("GNNX_TINY_PACK", "0")
        _, res20 = _run(be, iso, sd, 3)
        for i in range(2):
            assert np.array_equal(res2.masked_adj[i], res20.masked_adj[i]), i


def test_other_forms_keep_their_classes(be, monkeypatch):
    """Random feature rows (the general form), another hidden width, more than four classes: no packed launch."""
    rng = np.random.default_rng(33)
    for D, H, C, const in [(10, 20, 4, False), (10, 16, 4, True), (10, 20, 6, True)]:
        sd = helpers.random_model(rng, D, H, H, C)
        subs = _subs(rng, [(int(rng.integers(6, 29)), 2) for _ in range(4)])
        if not const:
            for s in subs:
                s.feat[:] = rng.standard_normal(s.feat.shape).astype(np.float32)
        job = be.job(subs, sd)
        assert job.tiny_pack() == (0, 0), (D, H, C, const)


def test_small_batches_are_not_packed(be, monkeypatch):
    """Below eight single-wave targets per compute unit of the chip a batch runs faster spread (gnnx_capi.hip: tiny pack rule)."""
    monkeypatch.delenv("GNNX_TINY_PACK_MIN")
    rng = np.random.default_rng(34)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    job = be.job(_subs(rng, [(int(rng.integers(6, 20)), 2) for _ in range(6)]), sd)
    assert job.tiny_pack() == (0, 0) and not job.tiny_packed().any()
