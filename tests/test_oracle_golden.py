"""Pin the oracle against the REAL reference's outputs (fixtures from tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import helpers
from oracle import closed_form, reference_restatement as rr

TOL = 1e-5   # parity tolerance of BASELINE.md on masked_adj and sigma(feat_mask)
from helpers import ILL_CONDITIONED, ILL_TOL_MASK, ILL_TOL_FEAT  # noqa: E402


def _tensor_sd(sd):
    return {k: torch.tensor(v) for k, v in sd.items()}


@pytest.mark.parametrize("name,targets", [("syn1", [302, 309, 555]), ("syn4", [511, 520, 700, 870]),
                                          ("syn5", [511, 515, 1000, 1230])])
def test_restatement_bit_exact_vs_reference(name, targets):
    """Same torch build + same seed => the restatement must reproduce the reference bit for bit."""
    torch.set_num_threads(1)
    ck, gx = helpers.load_ckpt(name), helpers.load_explain(name)
    for t in targets:
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        assert nb[new] == t
        m0 = helpers.seeded_mask0(t, len(nb))
        if f"{t}:mask0" in gx.files:
            assert np.array_equal(m0.numpy(), gx[f"{t}:mask0"])
        o = rr.MaskOptimOracle(torch.tensor(A), torch.tensor(X), _tensor_sd(ck["sd"]), lab[new], yhat, new, mask0=m0)
        out = o.run(int(gx["epochs"]), record=True)
        rc = gx[f"{t}:edge_rc"]
        want = gx[f"{t}:masked_adj_edges"]
        got = out[rc[:, 0], rc[:, 1]].astype(np.float32)
        assert np.array_equal(got, want), f"{name}/{t}: max diff {np.abs(got - want).max()}"
        assert np.all(out[A == 0] == 0)
        assert np.array_equal(torch.sigmoid(o.feat_mask).detach().numpy(), gx[f"{t}:feat_mask_sigmoid"])
        assert np.allclose(o.trace[:, 0], gx[f"{t}:loss"], rtol=3e-7, atol=0)   # logged scalar only: 1-ulp summation-order noise


@pytest.mark.parametrize("name,targets,epochs", [("syn1", [302, 309], 300), ("syn1", [555], 300), ("syn4", [511, 870], 300),
                                                 ("syn5", [515], 300)])
def test_closed_form_matches_reference(name, targets, epochs):
    """Analytic gradients + explicit Adam (the kernel spec) against the reference's own output."""
    ck, gx = helpers.load_ckpt(name), helpers.load_explain(name)
    for t in targets:
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        m0 = helpers.seeded_mask0(t, len(nb)).numpy()
        o = closed_form.ClosedFormOracle(A, X, ck["sd"], lab[new], yhat, new, m0)
        out = o.run(epochs)
        rc = gx[f"{t}:edge_rc"]
        got = out[rc[:, 0], rc[:, 1]]
        want = gx[f"{t}:masked_adj_edges"]
        assert np.abs(got - want).max() <= TOL, f"{name}/{t}: {np.abs(got - want).max()}"
        assert np.abs(closed_form.sigmoid(o.f) - gx[f"{t}:feat_mask_sigmoid"]).max() <= TOL
        loss = np.asarray(o.trace)[:, 0]
        assert np.allclose(loss, gx[f"{t}:loss"], rtol=2e-5, atol=2e-5)


def test_graph_mode_oracles_vs_reference():
    z = np.load(helpers.GOLDEN + "/graphmode_explain.npz")
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    epochs = int(z["epochs"])
    for g in range(3):
        A, X, lab = z["adj"][g], z["feat"][g], int(z["label"][g])
        m0 = helpers.seeded_mask0(g, A.shape[0])
        o = rr.MaskOptimOracle(torch.tensor(A), torch.tensor(X), _tensor_sd(sd), lab, None, 0, graph_mode=True, mask0=m0)
        out = o.run(epochs, record=True)
        assert np.array_equal(out.astype(np.float32), z[f"{g}:masked_adj"])
        c = closed_form.ClosedFormOracle(A, X, sd, lab, None, 0, m0.numpy(), graph_mode=True)
        out2 = c.run(epochs)
        assert np.abs(out2 - z[f"{g}:masked_adj"]).max() <= TOL
        assert np.abs(closed_form.sigmoid(c.f) - z[f"{g}:feat_mask_sigmoid"]).max() <= TOL


def test_ill_conditioned_targets_amplify_roundoff_even_on_cpu():
    """syn5 targets 511 / 1000 / 1230 sit on a loss plateau for ~70 epochs and then drop (3.7 -> 0.8 within a few
    epochs).  The epoch at which the drop happens depends on fp32 round-off, so ANY re-ordering of the arithmetic
    - here NumPy closed form vs the reference's torch autograd, both on the CPU - ends 1e-4..5e-3 apart, while
    well-conditioned targets agree to 1e-7.  This bounds what a GPU implementation can be held to on them
    (tests/test_gpu_parity.py uses the same relaxed tolerances for exactly these targets)."""
    ck, gx = helpers.load_ckpt("syn5"), helpers.load_explain("syn5")
    worst = 0.0
    for t in ILL_CONDITIONED["syn5"]:
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        o = closed_form.ClosedFormOracle(A, X, ck["sd"], lab[new], yhat, new, helpers.seeded_mask0(t, len(nb)).numpy())
        out = o.run(300)
        rc = gx[f"{t}:edge_rc"]
        err = np.abs(out[rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max()
        ferr = np.abs(closed_form.sigmoid(o.f) - gx[f"{t}:feat_mask_sigmoid"]).max()
        worst = max(worst, err)
        assert err <= ILL_TOL_MASK and ferr <= ILL_TOL_FEAT
        assert abs(o.trace[-1][0] - float(gx[f"{t}:loss"][-1])) <= 1e-2
    assert worst > TOL        # documents that these targets really are beyond the 1e-5 class
