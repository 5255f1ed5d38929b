"""Shared test helpers: load the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_ckpt(name):
    """-> dict(adj [N,N] f32 dense, feat [N,D], label [N], pred [N,C], sd {state_dict name: ndarray})."""
    z = np.load(os.path.join(GOLDEN, name + "_ckpt.npz"))
    n = int(z["num_nodes"])
    adj = np.zeros((n, n), np.float32)
    e = z["edges"]
    adj[e[:, 0], e[:, 1]] = 1
    adj[e[:, 1], e[:, 0]] = 1
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    return dict(adj=adj, feat=z["feat"], label=z["label"], pred=z["pred"], sd=sd, edges=e, num_nodes=n)


def load_explain(name):
    return np.load(os.path.join(GOLDEN, name + "_explain.npz"))


def subgraph(ck, nb):
    """Dense sub-adjacency / features / labels on the node id list nb (ascending)."""
    nb = np.asarray(nb)
    return ck["adj"][np.ix_(nb, nb)], ck["feat"][nb], ck["label"][nb], np.argmax(ck["pred"][nb], axis=1)


def seeded_mask0(target, n):
    """Seed protocol of the golden runs: torch.manual_seed(1000 + target) then one normal_ draw."""
    import math
    torch.manual_seed(1000 + int(target))
    std = math.sqrt(2.0) * math.sqrt(2.0 / (n + n))
    return torch.empty(n, n).normal_(1.0, std)


def dense_from_edges(n, rc, vals):
    out = np.zeros((n, n), np.float64)
    out[rc[:, 0], rc[:, 1]] = vals
    return out


# Targets whose optimisation crosses a loss plateau: fp32 round-off of any re-ordering is amplified (shown on the
# CPU alone by tests/test_oracle_golden.py::test_ill_conditioned_targets_amplify_roundoff_even_on_cpu).
ILL_CONDITIONED = {"syn5": (511, 1000, 1230)}
# (round 6: these two bound the CPU-vs-CPU amplification test of tests/test_oracle_golden.py only.  The GPU outcome tests no longer hold the three targets to
#  any number at the 300-epoch horizon - two CPU implementations differ by 5.5e-3 there, so no outcome bound says anything about a kernel; they are REPORTED, and gated
#  like every target where a bound means something: every 50- / 10-epoch window of them against the reference's own state, tests/test_decision_parity.py.)
ILL_TOL_MASK = 5e-4     # measured CPU-vs-CPU: up to 7.3e-5 (the GPU outcome tests keep this one on the masked adjacency)
ILL_TOL_FEAT = 1e-2     # measured CPU-vs-CPU: up to 5.5e-3


def random_model(rng, D, H, O, C):
    g = lambda *s: rng.standard_normal(s).astype(np.float32) * 0.5
    return {"conv_first.weight": g(D, H), "conv_first.bias": g(H) * 0.2, "conv_block.0.weight": g(H, H),
            "conv_block.0.bias": g(H) * 0.2, "conv_last.weight": g(H, O), "conv_last.bias": g(O) * 0.2,
            "pred_model.weight": g(C, 2 * H + O), "pred_model.bias": g(C) * 0.2}


def random_graph(rng, n, D, density=0.15):
    A = (rng.random((n, n)) < density).astype(np.float32)
    A = np.triu(A, 1)
    A = A + A.T
    for i in range(n - 1):          # keep it connected-ish
        A[i, i + 1] = A[i + 1, i] = 1
    return A, rng.standard_normal((n, D)).astype(np.float32)


def load_branches(name):
    """Alternate outcomes of the reference under 1-ulp perturbations of the initial mask (make_golden_branches.py)."""
    p = os.path.join(GOLDEN, name + "_branches.npz")
    return np.load(p) if os.path.exists(p) else None


def branch_errors(z, br, eoff, vals, feat_sig, early=False):
    """Per target: (err_mask, err_feat, matched) = distance of a result to the NEAREST legitimate outcome of the reference -
    its own output (fixture z, `vals` / `feat_sig`, or the `_early` horizon) or one of the alternate outcomes it produces
    under a 1-ulp perturbation of the initial mask (fixture br) - and which one matched (-1: the unperturbed output)."""
    sfx = "_early" if early else ""
    want_v, want_f = z["vals" + sfx].astype(np.float64), z["feat_sig" + sfx].astype(np.float64)
    vals, feat_sig = np.asarray(vals, np.float64), np.asarray(feat_sig, np.float64)
    T = len(eoff) - 1
    em = np.asarray([np.abs(vals[a:b] - want_v[a:b]).max() if b > a else 0.0 for a, b in zip(eoff[:-1], eoff[1:])])
    ef = np.abs(feat_sig - want_f).max(1)
    matched = np.full(T, -1, np.int64)
    if br is not None:
        for j in np.nonzero(br["alt_early"] == (1 if early else 0))[0]:
            k = int(br["alt_target"][j])
            a, b = eoff[k], eoff[k + 1]
            av = br["alt_vals"][br["alt_off"][j]:br["alt_off"][j + 1]].astype(np.float64)
            dm = np.abs(vals[a:b] - av).max() if b > a else 0.0
            df = np.abs(feat_sig[k] - br["alt_feat"][j]).max()
            if max(dm, df) < max(em[k], ef[k]):
                em[k], ef[k], matched[k] = dm, df, j
    return em, ef, matched


PARITY_TOL = 1e-5        # masked_adj and sigmoid(feat_mask), BASELINE.md section 3
WELL = 2e-6              # CPU-vs-CPU deviation (reference vs closed-form fp32 oracle) up to which a target is not chaotic
BRANCH_JUMP_MAX = 5e-3   # largest distance between two outcomes of one non-chaotic target seen under 1-ulp perturbations (syn4: 3.2e-3)
# Graph mode (config 4) after 300 epochs: the max-pool's arg-max rows switch on ties, 42 of the 64 fixture graphs move under a 1-ulp
# perturbation of the initial mask (38 by more than 1e-5, up to 6e-2 - make_golden_branches.py), so the full horizon only asks for
# 70 % within 1e-5 and bounds the rest by that jump; after 50 epochs 95 % (molecule-like graphs are full of symmetric atoms whose
# activations are equal in exact arithmetic: which of them wins the max-pool is decided by the summation order of each
# implementation - graph 2477 switches rows at epoch ~32 in the edge-sparse kernel while the dense streaming kernels stay with the
# reference, both on the same GPU; tests/golden/debug_graph2477.py).
# Round 3, 512 graphs instead of 64: a flipped max-pool tie followed by 250 more epochs moves single masks by up to 0.45 on graphs that are
# NOT calm, so a percentage of graphs within 1e-5 says nothing in graph mode.  The full-horizon gate of config 4 is decision-based
# (tests/test_decision_parity.py: every decision of every epoch against the live reference's; calm graphs bounded by CONFIG4_WINDOW_JUMP);
# the outcome comparisons (test_gpu_full_configs.py, bench.py --workload config4) require that EVERY miss has a window the CPU-only
# analysis of make_golden_windows.py flags.  (The 70 % rule of rounds 2-4 is gone.)
CONFIG4_WINDOW_JUMP = 6e-2


# ---------------------------------------------------------------------------------------------------------------------------
# Outcome rule of the full-config tests and of bench.py's in-run gate (round 6: NO percentage anywhere - VERDICT r5 "next" 4).
# The gate proper is decision-based (tests/test_decision_parity.py: every decision of every epoch against the live reference's).  What an
# OUTCOME comparison (300 / 50 epochs from the seeds against the reference's one output) may assert on top of it:
#   * a CALM target - conditioning over the horizon <= 2e-6, measured on the CPU alone before any implementation ran (CPU-vs-CPU deviation of the
#     fixture + the window probes: horizon_conditioning) - lies within 1e-5 of the reference's output, OR it is on the committed list
#     tests/golden/<name>_ties.json: the calm targets the decision suite, run on the GPU at the commit that wrote the list
#     (GNNX_WRITE_TIES=1 pytest tests/test_decision_parity.py -m gpu), found beyond 1e-5 - each with its reason: the first differing decision is
#     a tie of the reference (epoch, margin), or every decision identical and the drift within the accumulated round-off bound - and then it must
#     stay within the largest jump a flipped tie causes;
#   * anything else FAILS: a new miss is a regression until the decision suite has explained it and the list is regenerated.
# Targets that are not calm cannot be gated at a horizon by any implementation (two CPU implementations already differ on them); they are
# reported, and covered window by window by the decision suite.
# ---------------------------------------------------------------------------------------------------------------------------
def ties_path(name):
    return os.path.join(GOLDEN, name + "_ties.json")


def load_ties(name):
    """-> {"windows": {(id, w, sub): row}, "resolved": {(id, w): row}, "full": {id: row}, "early": {id: row}} or None when the list has not been generated
    ("resolved": 50-epoch windows that miss at 50 epochs while their five 10-epoch sub-windows are all gated - tests/test_decision_parity.py::_verdict)"""
    import json
    p = ties_path(name)
    if not os.path.exists(p):
        return None
    z = json.load(open(p))
    return {"windows": {(int(r["id"]), int(r["w"]), int(r["sub"])): r for r in z.get("windows", [])},
            "resolved": {(int(r["id"]), int(r["w"])): r for r in z.get("resolved", [])},
            "full": {int(r["id"]): r for r in z.get("full", [])},
            "early": {int(r["id"]) for r in z.get("windows", []) if int(r["w"]) == 0}}


def horizon_conditioning(name, cond_mask, cond_feat, early=False):
    """One number per target, CPU-only: the largest of the CPU-vs-CPU deviation at the horizon (fixture) and the conditioning probes of the windows the
    horizon spans (all six; the first one for the 50-epoch horizon)."""
    W = Windows(name)
    with np.load(os.path.join(GOLDEN, name + "_noise.npz")) as f:
        Nz = {k: f[k] for k in f.files}
    win = np.maximum(np.maximum(np.maximum(W.z["cond50"], W.z["sens50"]), Nz["noise50"]), Nz["ssens50"])
    win = win[:, 0] if early else win.max(1)
    return np.maximum(np.maximum(cond_mask, cond_feat), win)


def explained_outcome(name, horizon, ids, err, ferr, calm, jump_max=None):
    """The outcome rule above.  err / ferr: per target distance to the reference's ONE output (masked adjacency, sigmoid(feat_mask)); calm: the
    CPU-only classification.  -> (ok, message, unexplained ids)"""
    jump_max = BRANCH_JUMP_MAX if jump_max is None else jump_max
    ties = load_ties(name)
    listed = set() if ties is None else (set(ties["full"]) if horizon == "full" else ties["early"])
    e = np.maximum(err, ferr)
    miss = [k for k in np.nonzero(calm & (e > PARITY_TOL))[0]]
    unexplained = [int(ids[k]) for k in miss if int(ids[k]) not in listed]
    too_far = [int(ids[k]) for k in miss if e[k] > jump_max]
    inside = int((calm & (e <= PARITY_TOL)).sum())
    msg = (f"{inside} / {int(calm.sum())} calm targets within 1e-5 of the reference's output; beyond: {len(miss)} "
           f"({len(miss) - len(unexplained)} on the decision suite's committed list {os.path.basename(ties_path(name))}"
           f"{'' if ties is not None else ' - NOT GENERATED'}, unexplained: {unexplained}), worst {float(e[calm].max()) if calm.any() else 0.0:.2e} "
           f"(jump limit {jump_max:.0e}, beyond it: {too_far}); not calm (reported): {int((~calm).sum())} targets, "
           f"{int(((~calm) & (e <= PARITY_TOL)).sum())} of them within 1e-5 anyway, worst {float(e[~calm].max()) if (~calm).any() else 0.0:.2e}")
    return (not unexplained and not too_far), msg, unexplained


# ---------------------------------------------------------------------------------------------------------------------------
# Windowed ("teacher-forced") parity: tests/golden/<name>_windows.npz (make_golden_windows.py) holds the optimiser state of the
# live reference every 50 epochs on every target (and every 10 epochs inside the windows two CPU implementations already
# disagree on).  An implementation is started from the reference's state at a boundary and compared with the reference's
# state at the next one: round-off has 50 (10) iterations to act, not 300.
# ---------------------------------------------------------------------------------------------------------------------------
WIN_TOL = 1e-5           # masked adjacency (from the mask entries of both directions) and sigmoid(feat_mask) at the end of a window
WIN_FLAG = 2e-6          # CPU-vs-CPU deviation inside a window above which it is "flagged" (decided by make_golden_windows.py)


def _sig64(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def abar_from_mask_rc(mask_rc):
    """masked adjacency on unit-weight edges from the two directed mask entries [E, 2] (explain.py:665-678), float64."""
    return 0.5 * (_sig64(mask_rc[:, 0]) + _sig64(mask_rc[:, 1]))


class _Arrays(dict):
    """the arrays of an .npz file held in memory, with the `.files` attribute of the NpzFile they came from"""
    @property
    def files(self):
        return list(self.keys())


class Windows:
    """Accessors of a <name>_windows.npz fixture.  Boundary b = 0..6 is the state after 50 b steps (b = 0: the seeded initial
    mask with zero moments - not stored, None)."""

    def __init__(self, name):
        with np.load(os.path.join(GOLDEN, name + "_windows.npz")) as f:
            self.z = z = _Arrays({k: f[k] for k in f.files})      # in memory once: an NpzFile decompresses an array on EVERY access
        self.ids = z["targets"] if "targets" in z.files else z["graphs"]
        self.eoff = z["eoff"]
        self.T, self.W = z["cond50"].shape
        self.win = int(z["epochs"][0])
        self.sub = int(z["sub"])
        self.nsub = self.win // self.sub
        # conditioning of a window = max(CPU-vs-CPU deviation, deviation under a 1-ulp perturbation of its own start) - both measured on
        # the CPU by make_golden_windows.py before any implementation under test ran
        self.cond50 = self._conditioning(z["cond50"], z["sens50"], z["gate50"], float(z["gate"]))
        self.flagged = self.cond50 > WIN_FLAG                       # [T, W]
        self.fine_row = {(int(k), int(w)): i for i, (k, w) in enumerate(z["fine_tw"])}

    def edge_index(self, ks):
        """indices into the fixture's edge arrays of the targets ks, concatenated in that order"""
        parts = [np.arange(self.eoff[k], self.eoff[k + 1]) for k in ks]
        return np.concatenate(parts) if parts else np.zeros(0, np.int64)

    def boundary(self, b, ks):
        """state after 50 b steps of the targets ks: (first_iter, M, m, v [E', 2], f, mf, vf [T', D]) or None for b = 0"""
        if b == 0:
            return None
        z, e = self.z, self.edge_index(ks)
        return (self.win * b, z["M"][b - 1][e], z["m"][b - 1][e], z["v"][b - 1][e], z["f"][b - 1][ks], z["mf"][b - 1][ks], z["vf"][b - 1][ks])

    def fine(self, w, s, ks):
        """state after 50 w + 10 s steps (s = 1..4) of targets ks whose window w is flagged"""
        z = self.z
        rows = [self.fine_row[(int(k), int(w))] for k in ks]
        e = np.concatenate([np.arange(z["fine_off"][i], z["fine_off"][i + 1]) for i in rows]) if rows else np.zeros(0, np.int64)
        return (self.win * w + self.sub * s, z["fine_M"][s - 1][e], z["fine_m"][s - 1][e], z["fine_v"][s - 1][e],
                z["fine_f"][s - 1][rows], z["fine_mf"][s - 1][rows], z["fine_vf"][s - 1][rows])

    def sub_state(self, w, s, ks):
        """state at the start (s = 0..4) or end (s = 5) of 10-epoch sub-window s of window w"""
        if s == 0:
            return self.boundary(w, ks)
        if s == self.nsub:
            return self.boundary(w + 1, ks)
        return self.fine(w, s, ks)

    @staticmethod
    def _conditioning(cond, sens, gate, gate_thr):
        """One number per window, all measured on the CPU by make_golden_windows.py before any implementation under test ran: the
        larger of the CPU-vs-CPU deviation and the deviation under a 1-ulp perturbation of the window's own start, or 1.0 when a
        decision that reaches the loss (ReLU gate / max-pool) comes within fp32 round-off (`gate`) of its boundary inside the window."""
        c = np.maximum(cond, sens)
        return np.where(gate < gate_thr, np.maximum(c, 1.0), c)

    def cond10(self, w, ks):
        c = self._conditioning(self.z["cond10"], self.z["sens10"], self.z["gate10"], float(self.z["gate"]))
        return np.stack([c[self.fine_row[(int(k), int(w))]] for k in ks]) if len(ks) else np.zeros((0, self.nsub))


def run_window(job, start, iters, trace=False):
    """Start `job` (its M holding the seeded initial masks) from a fixture state (Windows.boundary / .fine; None = the initial
    state) and run `iters` iterations.  -> (mask_rc [E, 2], feat_mask [T, D]) after them; with trace=True also the decision
    trace of those iterations (MaskOptimJob.fetch_trace: gates per target, pool rows or None)."""
    from gnn_model_explainer_amd.engine import Hyper
    st = None
    if start is not None:
        st = job.set_state_edges(*start)
    job.launch(Hyper(num_iters=int(iters)), state=st, keep_state=True, trace=trace)
    mask_rc, _, _, fs = job.fetch_state_edges()
    if trace:
        gates, pool = job.fetch_trace()
        return mask_rc, fs[:, 0, :], gates, pool
    return mask_rc, fs[:, 0, :]


def window_errors(eoff, mask_rc, feat, want):
    """per target: (|masked adjacency - reference's|_max on its edges, |sigmoid(feat_mask) - reference's|_max)"""
    d = np.abs(abar_from_mask_rc(mask_rc) - abar_from_mask_rc(want[1]))
    em = np.asarray([d[a:b].max() if b > a else 0.0 for a, b in zip(eoff[:-1], eoff[1:])])
    ef = np.abs(_sig64(feat) - _sig64(want[4])).max(1)
    return em, ef


# ---------------------------------------------------------------------------------------------------------------------------
# Decision parity: tests/golden/<name>_decisions.npz (make_golden_decisions.py) holds the side of every discrete decision of the
# LIVE reference's forward - ReLU gates that reach the loss, graph-mode max-pool rows - at every epoch of every target, plus the
# NEAR list (gates with |U| < 1e-5, pools whose winner leads by < 1e-5).  The engine's own decisions come from gnnx_set_trace.
# ---------------------------------------------------------------------------------------------------------------------------
class Decisions:
    def __init__(self, name):
        with np.load(os.path.join(GOLDEN, name + "_decisions.npz")) as f:
            self.z = z = {k: f[k] for k in f.files}          # in memory once: an NpzFile decompresses an array on EVERY access
        self.ids = z["targets"] if "targets" in z else z["graphs"]
        self.T = len(self.ids)
        self.graph_mode = "pool0" in z
        self.near_tol = float(z["near"])          # the NEAR list holds every gate with |U| below this (1e-4) and every pool with a smaller margin
        self.near_tol_strict = 1e-5               # = the parity tolerance: a decision the reference takes by less may differ in any window
        self._cache = {}

    @staticmethod
    def _forward_fill(first, epoch, where, value, E):
        """[E, *first.shape]: `first` at epoch 0, entry `where` set to `value` from `epoch` on (change events; values >= 0)"""
        c = np.full((E,) + first.shape, -1, np.int64)
        c[0] = first
        c[(epoch,) + tuple(where)] = value
        idx = np.where(c >= 0, np.arange(E).reshape((E,) + (1,) * first.ndim), 0)
        np.maximum.accumulate(idx, axis=0, out=idx)
        return np.take_along_axis(c, idx, axis=0)

    def _all_gate_words(self, k):
        """uint32 [epochs, n_k, 2] of target k, built once from the change events (cached: a window test asks for every window of a target)"""
        c = self._cache.get(("g", k))
        if c is None:
            z = self.z
            a, b = int(z["row_off"][k]), int(z["row_off"][k + 1])
            E = int(z["epochs"])
            ev = z["ev"][int(z["ev_off"][k]):int(z["ev_off"][k + 1])]
            c = self._forward_fill(z["gates0"][a:b].astype(np.int64), ev[:, 0], (ev[:, 1], ev[:, 2]), ev[:, 3].astype(np.int64) & 0xffffffff, E).astype(np.uint32)
            self._cache[("g", k)] = c
        return c

    def gate_words(self, k, e0, e1):
        """uint32 [e1 - e0, n_k, 2]: the reference's sign words at epochs e0 .. e1 - 1 of target k (fixture index)"""
        return self._all_gate_words(k)[e0:e1]

    def pool_rows(self, k, e0, e1):
        c = self._cache.get(("p", k))
        if c is None:
            z = self.z
            ev = z["pev"][int(z["pev_off"][k]):int(z["pev_off"][k + 1])]
            c = self._forward_fill(z["pool0"][k].astype(np.int64), ev[:, 0], (ev[:, 1], ev[:, 2]), ev[:, 3].astype(np.int64), int(z["epochs"])).astype(np.int32)
            self._cache[("p", k)] = c
        return c[e0:e1]

    def near_gates(self, k):
        """{(epoch, layer, row, column): U} of the gates of target k whose |U| < NEAR in the reference"""
        z = self.z
        a, b = int(z["near_off"][k]), int(z["near_off"][k + 1])
        return {tuple(int(x) for x in ev): float(v) for ev, v in zip(z["near_ev"][a:b], z["near_val"][a:b])}

    def near_pools(self, k):
        z = self.z
        a, b = int(z["pnear_off"][k]), int(z["pnear_off"][k + 1])
        return {tuple(int(x) for x in ev): float(v) for ev, v in zip(z["pnear_ev"][a:b], z["pnear_val"][a:b])}

    def first_disagreement(self, k, e0, gates, pool=None):
        """Engine trace of target k for epochs e0 .. e0 + len(gates) - 1 (gates uint32 [iters, n, 2], pool int32 [iters, 3, 32] or None)
        against the reference's decisions.  -> None when every decision of every epoch agrees, else
        (epoch, what, margin): the first epoch with a difference, a description of the differing decisions, and the LARGEST |U| /
        pool margin the reference has on any of them (inf when one of them is not on its NEAR list: not a tie)."""
        iters = gates.shape[0]
        want = self.gate_words(k, e0, e0 + iters)
        dg = (want != gates).reshape(iters, -1).any(1)
        dp = np.zeros(iters, bool)
        if pool is not None:
            wp = self.pool_rows(k, e0, e0 + iters)
            dp = (wp != pool[:, :, :20]).reshape(iters, -1).any(1)
        bad = np.nonzero(dg | dp)[0]
        if not len(bad):
            return None
        i = int(bad[0])
        e = e0 + i
        margin, what = 0.0, []
        if dg[i]:
            near = self.near_gates(k)
            r, l = np.nonzero(want[i] != gates[i])
            for rr, ll in zip(r, l):
                x = int(want[i][rr, ll] ^ gates[i][rr, ll])
                for c in range(20):
                    if x >> c & 1:
                        v = near.get((e, int(ll), int(rr), c))
                        margin = max(margin, abs(v) if v is not None else np.inf)
                        what.append(("gate", int(ll), int(rr), c, v))
        if dp[i]:
            near = self.near_pools(k)
            l, c = np.nonzero(wp[i] != pool[i][:, :20])
            for ll, cc in zip(l, c):
                v = near.get((e, int(ll), int(cc)))
                margin = max(margin, abs(v) if v is not None else np.inf)
                what.append(("pool", int(ll), int(cc), int(wp[i][ll, cc]), int(pool[i][ll, cc]), v))
        return e, what, margin
