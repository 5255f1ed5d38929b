#!/usr/bin/env python
"""Measurement tool (CPU, emulator): how far does ONE target's 300-iteration result move when the hardware-form intrinsics
(rcp / sqrt / exp) are perturbed by +-1 ulp at random?  Runs the product's kernel sources through tests/emu with
GNNX_EMU_ULP_NOISE=<seed> for a number of seeds (one process each: the switch is read once) and prints the deviations from the
reference fixture.   python tools/ulp_sensitivity.py syn5 767 16"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 4 and sys.argv[4] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from emu.emu_engine import emu_library
    from gnn_model_explainer_amd import engine
    from gnn_model_explainer_amd.engine import Hyper, Subgraph
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    name, tt = sys.argv[1], int(sys.argv[2])
    z = np.load(os.path.join(ROOT, "tests", "golden", name + "_full_explain.npz"))
    ck = helpers.load_ckpt(name)
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    k = int(np.nonzero(z["targets"] == tt)[0][0])
    nb = z["nb_flat"][z["nb_off"][k]:z["nb_off"][k + 1]].astype(np.int64)
    new = int(z["node_idx_new"][k])
    A = idx.sub_adjacency(nb)
    m0 = helpers.seeded_mask0(tt, A.shape[0]).numpy()
    r, c = np.nonzero(np.triu(A, 1))
    main = z["vals"][z["eoff"][k]:z["eoff"][k + 1]]
    sg = Subgraph(A, ck["feat"][nb], int(ck["label"][tt]), new, np.argmax(ck["pred"][nb], 1), m0)
    res = engine.MaskOptimJob([sg], ck["sd"], device="cpu", lib=emu_library()).run([m0], Hyper(num_iters=300))
    fs = 1.0 / (1.0 + np.exp(-res.feat_mask[0][:ck["feat"].shape[1]].astype(np.float64)))
    print("DEV %.3e" % max(np.abs((res.masked_adj[0] * A)[r, c] - main).max(), np.abs(fs - z["feat_sig"][k]).max()))
    sys.exit(0)
name, tt, trials = sys.argv[1], sys.argv[2], int(sys.argv[3])
devs = []
for s in range(trials):
    env = dict(os.environ, GNNX_EMU_ULP_NOISE=str(s))
    out = subprocess.run([sys.executable, __file__, name, tt, "0", "--child"], env=env, capture_output=True, text=True).stdout
    devs.append(float(out.split("DEV")[1]))
    print(f"seed {s}: {devs[-1]:.3e}", flush=True)
devs = np.asarray(devs)
print(f"{name} target {tt}: {trials} runs with +-1 ulp hardware forms: max {devs.max():.2e}, {np.mean(devs > 1e-5) * 100:.0f} % beyond 1e-5, {np.mean(devs > 1e-4) * 100:.0f} % beyond 1e-4")
