"""Jump-ahead of the mask streams' engine (utils/mt_jump.py; k_mt_segment_starts / k_mt_edge_words_seg in csrc/gnnx_xl.hpp).

CPU: the characteristic polynomial and the jump polynomial are recomputed and compared with the committed file; a jump equals the plain walk.
Emulator + GPU: the segmented walk (small strides, so that small sub-graphs have several segments) gives bit for bit the masks of
`torch.manual_seed(seed); torch.FloatTensor(n, n).normal_(1, std)` on every edge - ragged streams, tails in the last segment, one segment, many."""
import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Subgraph
from gnn_model_explainer_amd.utils import mt_jump
from test_emu_kernels import _Backend


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def test_committed_jump_polynomial_and_the_jump_itself():
    phi = mt_jump.charpoly()
    assert phi.bit_length() - 1 == mt_jump.DEG and phi.bit_count() == 135       # MT19937's characteristic polynomial has 135 terms
    assert np.array_equal(mt_jump.poly_words(mt_jump.jump_poly(mt_jump.JUMP, phi)), mt_jump.load_jump_poly())
    assert mt_jump.JUMP % 624 == 0
    x = mt_jump.mt_words(987654321, 624 * 6 + 20000)
    for J in (1, 623, 624, 3 * 624, 17001):
        g = mt_jump.poly_words(mt_jump.jump_poly(J, phi))
        for m in (1, 624, 1000):
            assert np.array_equal(mt_jump.jump_apply(x[m:m + 624], g), x[m + J:m + J + 624]), (J, m)


def test_jump_polynomials_of_the_tree_levels():
    """x^(4 J), x^(16 J) mod phi from x^J mod phi by squarings = computed from scratch"""
    phi = mt_jump.charpoly()
    for J in (624, 5 * 624):
        P = mt_jump.jump_polys(mt_jump.poly_words(mt_jump.jump_poly(J, phi)), 3, phi)
        assert np.array_equal(P[1], mt_jump.poly_words(mt_jump.jump_poly(4 * J, phi)))
        assert np.array_equal(P[2], mt_jump.poly_words(mt_jump.jump_poly(16 * J, phi)))


@pytest.mark.parametrize("levels", [1, 2, 3])
def test_segment_starts_as_a_tree_of_jumps(be, levels, monkeypatch):
    """The radix-4 tree of segment starts (strides J, 4 J, 16 J): stride = one block, so that n = 129 has 26 segments (one jump of 16, chains of 4s
    and 1s), n = 200 has 64 (three jumps of 16), n = 64 six - every target's edge entries bit for bit torch's, whatever the number of levels."""
    if not engine.pair_staging_ok():
        pytest.skip("the host's normal_ lacks the pair-staging property")
    monkeypatch.setenv("GNNX_MT_JUMP_LEVELS", str(levels))
    engine.enable_mt_jump(be.lib, mt_jump.poly_words(mt_jump.jump_poly(624)), 624)
    try:
        rng = np.random.default_rng(77)
        sd = helpers.random_model(rng, 10, 20, 20, 4)
        sgs, seeds = [], []
        for n, dens in ((129, 0.05), (200, 0.02), (64, 0.3), (37, 0.3), (5, 1.0)):
            A, X = helpers.random_graph(rng, n, 10, density=dens)
            sgs.append(Subgraph(A, X, 0, 0, rng.integers(0, 4, n), None))
            seeds.append(int(rng.integers(0, 2 ** 31)))
        xj = engine.xl_job_from_subgraphs(sgs, sd, device=be.device, lib=be.lib)
        seeds = np.asarray(seeds, np.int64)
        xj.set_masks_seeded_device(seeds, threads=2)
        got = xj.M_e[:xj.E].cpu().numpy()
        eoff, rc = xj.edge_ids()
        rc = rc.cpu().numpy()
        for k, sgr in enumerate(sgs):
            n = sgr.adj.shape[0]
            torch.manual_seed(int(seeds[k]))
            m0 = torch.empty(n, n).normal_(1.0, np.sqrt(2.0) * np.sqrt(2.0 / (n + n))).numpy()
            e = rc[eoff[k]:eoff[k + 1]]
            want = np.stack([m0[e[:, 0], e[:, 1]], m0[e[:, 1], e[:, 0]]], 1)
            assert np.array_equal(got[eoff[k]:eoff[k + 1]], want), (k, n)
    finally:
        engine.enable_mt_jump(be.lib, jump=0)
        engine._MT_JUMP_SET.discard(id(be.lib))


@pytest.mark.parametrize("blocks", [1, 3, 40])
def test_segmented_engine_walk_draws_the_reference_masks(be, blocks):
    """stride = `blocks` x 624 draws: n = 37 has 1369 draws (ragged; 1 / 2 segments), n = 129 has 16 641 (26 / 8 segments), n = 64 exactly 4096;
    blocks = 40: most targets walk as ONE segment from the jumped-from first block"""
    if not engine.pair_staging_ok():
        pytest.skip("the host's normal_ lacks the pair-staging property")
    J = blocks * 624
    engine.enable_mt_jump(be.lib, mt_jump.poly_words(mt_jump.jump_poly(J)), J)
    try:
        rng = np.random.default_rng(5 + blocks)
        sd = helpers.random_model(rng, 10, 20, 20, 4)
        sgs, seeds = [], []
        for n, dens in ((3, 1.0), (4, 1.0), (37, 0.3), (64, 0.9), (129, 0.05), (200, 0.02), (25, 1.0)):
            A, X = helpers.random_graph(rng, n, 10, density=dens)
            sgs.append(Subgraph(A, X, 0, 0, rng.integers(0, 4, n), None))
            seeds.append(int(rng.integers(0, 2 ** 31)))
        xj = engine.xl_job_from_subgraphs(sgs, sd, device=be.device, lib=be.lib)
        seeds = np.asarray(seeds, np.int64)
        xj.set_masks_seeded_device(seeds, threads=2)
        got = xj.M_e[:xj.E].cpu().numpy()
        eoff, rc = xj.edge_ids()
        rc = rc.cpu().numpy()
        for k, sgr in enumerate(sgs):
            n = sgr.adj.shape[0]
            torch.manual_seed(int(seeds[k]))
            m0 = torch.empty(n, n).normal_(1.0, np.sqrt(2.0) * np.sqrt(2.0 / (n + n))).numpy()
            e = rc[eoff[k]:eoff[k + 1]]
            want = np.stack([m0[e[:, 0], e[:, 1]], m0[e[:, 1], e[:, 0]]], 1)
            assert np.array_equal(got[eoff[k]:eoff[k + 1]], want), (k, n)
    finally:
        engine.enable_mt_jump(be.lib, jump=0)
        engine._MT_JUMP_SET.discard(id(be.lib))
