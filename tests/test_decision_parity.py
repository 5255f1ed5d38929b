"""Decision-conditional parity: EVERY window of EVERY target is either gated at 1e-5 or explained by a tie of the reference itself.

The reference's forward (explain.py:685-715 -> models.py:230-316) contains two kinds of discrete decisions: the ReLU gates of the two
hidden layers (models.py:241, 251) and, in graph mode, the rows the three max-pools pick (models.py:283, 291, 300).  Between two epochs at
which one of them changes sides the optimiser state is a smooth function of its predecessor, so:

  (a) a window in which the engine takes the SAME side of every decision at every epoch as the live reference did (fixture
      <name>_decisions.npz, recorded with forward hooks on the reference's own modules; engine side: gnnx_set_trace) must end within
      1e-5 of the reference's optimiser state (masked adjacency from the mask entries of both directions, sigmoid(feat_mask)) - no
      exceptions, no percentages, whatever CPU-side conditioning probes say about the window;
  (b) a window in which a decision differs is accepted only if, at the FIRST epoch with a difference, every differing decision is one
      the reference itself takes by less than the parity tolerance (|U| < 1e-5 at the gate, winner of the max-pool less than 1e-5
      ahead: its NEAR list) - the engine's state may differ from the reference's by the tolerance, so such a decision cannot be
      required to agree - and the window's error stays below the largest jump a flipped tie causes (5e-3; graph mode 6e-2).  These
      windows are listed with the epoch, the gate and the reference's margin: that list is the trace of every miss.
  (c) "within round-off" needs a scale.  This very test found smooth stretches - every decision identical - at whose end the engine
      sits 1e-5 .. 7e-5 from the reference (a dozen 50-epoch Tree-Grid windows, all in sigmoid(feat_mask)).  The scale is the window's own
      conditioning c, measured on the reference's side by CPU-only probes before any implementation ran: CPU-vs-CPU deviation and 1-ulp
      sensitivity of the window's starting MASK (make_golden_windows.py), the deviation under 1 ulp of summand-scale noise on the gradient
      sums in every iteration, and - round 5 - the 1-ulp sensitivity of the window to its WHOLE starting state (make_golden_noise_probe.py:
      ssens50).  The last one is what explains those windows: tools/drift_per_epoch.py (profiles/r05_syn5_drift_per_epoch.txt) shows the
      engine's distance opening in the window's FIRST steps with a one-ulp difference of a feature-mask parameter and then growing at the
      window's own rate - the closed form's distance to the reference grows at the same rate from a later start - and a window that turns
      one ulp of its feature mask into 3e-5 cannot be reproduced to better than that by any implementation whose update differs by an ulp.
      With it every such window ends within 1.7 c (round 4 needed 50 c of the three older probes: they perturbed the mask entries only).
      A window with identical decisions must end within max(1e-5, 4 c) - 4 = the room for a worse case than the eight random trials the
      probe samples - and never more than the largest jump of a flipped tie.  Windows with c > 2e-6 are re-run as 10-epoch sub-windows where
      the fixture has the snapshots and judged there by the same rule; the same 4 c is the margin up to which a differing decision counts
      as a tie in (b).  Every window beyond 1e-5 is listed with its c.
  Anything else - a differing decision the reference takes by a clear margin, an agreed window beyond max(1e-5, 4 c) - fails.

Windows are the 50-epoch windows of tests/golden/<name>_windows.npz (the live reference's Adam state, teacher forcing through
gnnx_run_resume); a window of kind (b) is re-run as its five 10-epoch sub-windows where the fixture holds the 10-epoch snapshots, so
that only the 10 epochs around the tie stay ungated.  GPU: every target of syn1 / syn4 / syn5 and the 512 config-4 graphs; emulator: a few
targets per config.

Round 6: no percentage anywhere.  The windows of kind (b) / (c) - everything this suite does NOT gate at plain 1e-5 - are a COMMITTED list
(tests/golden/<name>_ties.json: id, window, sub-window, kind, epoch, the reference's margin, the error, the conditioning), written by this suite on the
GPU with GNNX_WRITE_TIES=1 and checked on every later run: a window of kind (b) / (c) that is not on the list fails (the regression floor the
90 % share of rounds 4-5 stood for), and the outcome tests (tests/test_gpu_full_configs.py, bench.py's in-run gate) accept a calm target beyond
1e-5 only when the full-horizon part of the list explains it (helpers.explained_outcome).
"""
import os

import numpy as np
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
from test_emu_kernels import _Backend
from test_windowed_parity import _node_subgraph_job

TOL = helpers.WIN_TOL
ROUNDOFF_BUDGET = 4.0      # factor over the window's CPU-measured conditioning c an implementation may differ by (see (c) above; 50 in round 4)


def _judge(Dn, k, e0, gates, pool, err, ident, w, sub, smooth):
    """-> row dict for one (target, window[, sub-window]); smooth = max(CPU-vs-CPU deviation, 1-ulp sensitivity) of that window, from the
    fixture (make_golden_windows.py probes (i) and (ii))"""
    fd = Dn.first_disagreement(k, e0, gates, pool)
    row = dict(id=int(ident), w=int(w), sub=int(sub), e0=int(e0), iters=int(gates.shape[0]), err=float(err), agree=fd is None,
               smooth=float(smooth), expansive=bool(smooth > helpers.WIN_FLAG))
    if fd is not None:
        row.update(epoch=int(fd[0]), what=fd[1], margin=float(fd[2]))
    return row


def _decision_windows(W, Dn, make_job, ks_all, coarse_windows=None):
    """All 50-epoch windows of the targets ks_all; windows with a differing decision are re-run as 10-epoch sub-windows where the
    fixture has the snapshots.  -> list of row dicts."""
    rows = []
    name = "config4" if Dn.graph_mode else [n for n in ("syn1", "syn4", "syn5") if np.array_equal(helpers.Windows(n).ids, W.ids)][0]
    with np.load(os.path.join(helpers.GOLDEN, name + "_noise.npz")) as f:
        Nz = {k: f[k] for k in f.files}
    job = make_job(ks_all)
    eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[ks_all])])
    for w in (range(W.W) if coarse_windows is None else coarse_windows):
        mask_rc, feat, gates, pool = helpers.run_window(job, W.boundary(w, ks_all), W.win, trace=True)
        em, ef = helpers.window_errors(eoff, mask_rc, feat, W.boundary(w + 1, ks_all))
        redo = []
        for i, k in enumerate(ks_all):
            row = _judge(Dn, k, W.win * w, gates[i], None if pool is None else pool[i], max(em[i], ef[i]), W.ids[k], w, -1,
                         max(W.z["cond50"][k, w], W.z["sens50"][k, w], Nz["noise50"][k, w], Nz["ssens50"][k, w]))
            if (not row["agree"] or row["expansive"]) and (int(k), int(w)) in W.fine_row:
                redo.append(int(k))
                row["redo"] = True      # judged through its five 10-epoch sub-windows (below); kept for the list's "resolved" part (see _verdict)
            rows.append(row)
        if not redo:
            continue
        ks = np.asarray(redo, np.int64)
        sub_job = make_job(ks)     # fresh: sub-window 0 of window 0 starts from the seeded initial masks
        sub_eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[ks])])
        for s in range(W.nsub):
            mask_rc, feat, gates, pool = helpers.run_window(sub_job, W.sub_state(w, s, ks), W.sub, trace=True)
            em, ef = helpers.window_errors(sub_eoff, mask_rc, feat, W.sub_state(w, s + 1, ks))
            for i, k in enumerate(ks):
                f = W.fine_row[(int(k), int(w))]
                rows.append(_judge(Dn, k, W.win * w + W.sub * s, gates[i], None if pool is None else pool[i], max(em[i], ef[i]), W.ids[k], w, s,
                                   max(W.z["cond10"][f, s], W.z["sens10"][f, s], Nz["noise10"][f, s], Nz["ssens10"][f, s])))
    return rows


_TIES = {}      # name -> {"windows": [...], "full": [...]} collected by this session's GPU runs (written by _write_ties when GNNX_WRITE_TIES=1)


def _tie_row(r, **extra):
    d = r.get("what", [None])[0]
    out = dict(id=int(r["id"]), kind="tie" if not r["agree"] else "drift", epoch=int(r.get("epoch", -1)),
               margin=(None if not np.isfinite(r.get("margin", 0.0)) else float(r.get("margin", 0.0))), err=float(r["err"]),
               c=float(r.get("smooth", r.get("cond", 0.0))), decision=(None if d is None else [x if not isinstance(x, (np.integer, np.floating)) else x.item() for x in d]))
    out.update(extra)
    return out


def _write_ties(name):
    import json
    if os.environ.get("GNNX_WRITE_TIES") != "1" or name not in _TIES:
        return
    path = helpers.ties_path(name)
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(_TIES[name])
    old["generated_by"] = "GNNX_WRITE_TIES=1 python -m pytest tests/test_decision_parity.py -m gpu (MI355X, libgnnx_hip.so of this commit)"
    json.dump(old, open(path, "w"), indent=0, sort_keys=True)


def _check_listed(name, part, keys_rows, key_of):
    """Every row this run could not gate at plain 1e-5 must be on the committed list (unless the list is being written)."""
    if name is None:
        return
    _TIES.setdefault(name, {})[part] = [r for _, r in keys_rows]
    if os.environ.get("GNNX_WRITE_TIES") == "1":
        _write_ties(name)
        return
    ties = helpers.load_ties(name)
    assert ties is not None, f"{helpers.ties_path(name)} missing: generate it with GNNX_WRITE_TIES=1 on the GPU"
    new = [k for k, _ in keys_rows if key_of(k) not in ties[part]]
    assert not new, f"{name}: {len(new)} {part} rows beyond plain 1e-5 that the committed list does not hold (a regression until explained): {new[:10]}"


def _verdict(what, rows, Dn, jump, list_name=None):
    """rows of _decision_windows -> summary string; asserts the rules of the module docstring.  list_name: the fixture name whose committed list
    (tests/golden/<name>_ties.json, part "windows") every row not gated at plain 1e-5 must be on (GPU runs over ALL targets; None: a partial run)."""
    bound = lambda r: max(TOL, ROUNDOFF_BUDGET * r["smooth"])
    # 50-epoch windows that were re-run as 10-epoch sub-windows are judged by those; the ones that end beyond 1e-5 (or with a differing decision)
    # at 50 epochs while ALL their sub-windows - teacher-forced from the reference's own 10-epoch states - are gated at plain 1e-5 with identical
    # decisions go on the list's "resolved" part: the outcome tests over whole windows (tests/test_windowed_parity.py) look them up there.
    redone = [r for r in rows if r.get("redo")]
    rows = [r for r in rows if not r.get("redo")]
    subs_of = {}
    for r in rows:
        if r["sub"] >= 0:
            subs_of.setdefault((r["id"], r["w"]), []).append(r)
    resolved = [r for r in redone if (not r["agree"] or r["err"] > TOL) and all(x["agree"] and x["err"] <= TOL for x in subs_of.get((r["id"], r["w"]), []))]
    agreed = [r for r in rows if r["agree"]]
    ties = [r for r in rows if not r["agree"]]
    total = sum(r["iters"] for r in rows)
    gated = sum(r["iters"] for r in agreed)
    strict = sum(r["iters"] for r in agreed if r["err"] <= TOL)
    over = [r for r in agreed if r["err"] > TOL]                              # identical decisions, beyond 1e-5: listed, bounded by 4 c
    bad_agreed = [r for r in over if r["err"] > min(bound(r), jump)]
    tie_margin = lambda r: max(Dn.near_tol_strict, ROUNDOFF_BUDGET * r["smooth"])
    unknown = [r for r in ties if not np.isfinite(r["margin"])]                # a differing gate beyond the fixture's NEAR list (|U| >= 1e-4)
    unjust = [r for r in ties if (np.isfinite(r["margin"]) and r["margin"] >= tie_margin(r)) or (not np.isfinite(r["margin"]) and tie_margin(r) < Dn.near_tol)]
    mg = np.asarray([r["margin"] for r in ties if np.isfinite(r["margin"])])
    msg = (f"{what}: {len(rows)} windows ({total} target-epochs); every decision identical to the reference's in {len(agreed)} windows = "
           f"{100.0 * gated / max(1, total):.2f} % of the target-epochs (within 1e-5: {len(agreed) - len(over)} windows = {100.0 * strict / max(1, total):.2f} %; the other "
           f"{len(over)} within 4 x their CPU-measured conditioning, worst {max([r['err'] for r in over], default=0.0):.2e}; beyond that: {len(bad_agreed)}); "
           f"a differing decision in {len(ties)} windows: {len(ties) - len(unjust)} first at a tie of the reference (its margin there: {int((mg < 1e-7).sum())} below 1e-7, "
           f"{int((mg < 1e-6).sum())} below 1e-6, {int((mg < 1e-5).sum())} below 1e-5, {len(unknown)} not on the fixture's list), {int(sum(r['err'] <= TOL for r in ties))} of them "
           f"within 1e-5 anyway, worst {max([r['err'] for r in ties], default=0.0):.2e}; not at a tie: {len(unjust)}")
    print(msg)
    for r in sorted(over, key=lambda r: -r["err"])[:40]:
        print(f"{what}: same decisions, beyond 1e-5: id {r['id']} window {r['w']} sub {r['sub']}: error {r['err']:.2e}, conditioning of the window measured on the CPU "
              f"{r['smooth']:.2e} (bound {bound(r):.1e})")
    for r in sorted(ties, key=lambda r: -r["err"])[:60]:
        d = r["what"][0]
        desc = (f"U{d[1] + 1}[{d[2]}][{d[3]}] = {d[4]}" if d[0] == "gate" else f"pool {d[1] + 1} column {d[2]}: row {d[3]} vs {d[4]}, margin {d[5]}")
        print(f"{what}: tie   id {r['id']} window {r['w']} sub {r['sub']}: first differing decision at epoch {r['epoch']} ({len(r['what'])} decision(s); {desc}), "
              f"largest margin of the reference on them {r['margin']:.2e} (conditioning of the window {r['smooth']:.1e}), error at the end of the window {r['err']:.2e}")
    dump = os.environ.get("GNNX_DUMP_WINDOWS")
    if dump:
        os.makedirs(dump, exist_ok=True)
        np.save(os.path.join(dump, what.split(" ")[0] + "_decision_rows.npy"),
                np.asarray([(r["id"], r["w"], r["sub"], r["iters"], r["err"], r["agree"], r.get("epoch", -1), r.get("margin", 0.0), r["smooth"]) for r in rows], np.float64))
    assert not bad_agreed, msg + f"; {[(r['id'], r['w'], r['sub'], r['err'], r['smooth']) for r in bad_agreed[:10]]}"
    assert not unjust, msg + f"; first: {unjust[0]}"
    assert all(r["err"] <= jump for r in ties), msg
    _check_listed(list_name, "windows", [((r["id"], r["w"], r["sub"]), _tie_row(r, w=int(r["w"]), sub=int(r["sub"]))) for r in over + ties], lambda k: k)
    _check_listed(list_name, "resolved", [((r["id"], r["w"]), _tie_row(r, w=int(r["w"]), sub=-1)) for r in resolved], lambda k: k)
    if resolved:
        print(f"{what}: {len(resolved)} 50-epoch windows beyond 1e-5 / with a differing decision whose five 10-epoch sub-windows are all gated at 1e-5: "
              f"{[(r['id'], r['w'], float('%.2e' % r['err'])) for r in resolved[:20]]}")
    return msg


@pytest.mark.parametrize("name,picks", [("syn4", 3), ("syn5", 4), ("syn1", 2)])
def test_decision_windows_on_the_emulator_few_targets(name, picks):
    be = _Backend("emu")
    W, Dn = helpers.Windows(name), helpers.Decisions(name)
    assert np.array_equal(W.ids, Dn.ids)
    full = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    size = np.diff(full["nb_off"])
    small = np.nonzero(size <= (60 if name == "syn1" else 40))[0]
    fl = [k for k in small if W.flagged[k].any()]
    ks = list(small[np.linspace(0, len(small) - 1, picks).astype(int)])
    if fl:
        ks[-1] = fl[0]
    ks = np.asarray(sorted(set(int(k) for k in ks)), np.int64)
    rows = _decision_windows(W, Dn, _node_subgraph_job(be, name, W, full), ks)
    _verdict(f"{name} (emulator, targets {[int(W.ids[k]) for k in ks]})", rows, Dn, helpers.BRANCH_JUMP_MAX)


def _config4_job_maker(W, be=None):
    from gnn_model_explainer_amd.utils import synthetic
    sd = {k[2:]: W.z[k] for k in W.z.files if k.startswith("w:")}
    A, X, nn, y = synthetic.molecule_like_graphs(int(W.ids.max()) + 1, seed=0)

    def make(kk):
        subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(g, A.shape[1]).numpy()) for g in W.ids[kk]]
        job = be.job(subs, sd, graph_mode=True) if be is not None else MaskOptimJob(subs, sd, graph_mode=True)
        job.set_masks([s.mask0 for s in subs])
        return job
    return make


def test_decision_windows_graph_mode_on_the_emulator():
    W, Dn = helpers.Windows("config4"), helpers.Decisions("config4")
    assert np.array_equal(W.ids, Dn.ids)
    ks = np.asarray([0, int(np.nonzero(W.flagged.any(1))[0][0])], np.int64)
    rows = _decision_windows(W, Dn, _config4_job_maker(W, _Backend("emu")), ks, coarse_windows=(0, 3))
    _verdict("config4 (emulator)", rows, Dn, helpers.CONFIG4_WINDOW_JUMP)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn1", "syn4", "syn5"])
def test_decision_windows_every_target_every_window_gpu(name):
    """BASELINE configs 2 and 3: ALL 400 / 360 / 720 targets x 6 windows."""
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    W, Dn = helpers.Windows(name), helpers.Decisions(name)
    ck = helpers.load_ckpt(name)
    full = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    assert np.array_equal(W.ids, full["targets"]) and np.array_equal(W.ids, Dn.ids)
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])

    def make(ks):
        targets = W.ids[ks]
        nbs = [full["nb_flat"][full["nb_off"][k]:full["nb_off"][k + 1]].astype(np.int64) for k in ks]
        job = MaskOptimJob.from_csr(graph, nbs, full["node_idx_new"][ks], ck["label"][targets], ck["sd"])
        job.set_masks_raw(engine.init_edge_masks_raw([len(nb) for nb in nbs], seeds=1000 + targets))
        return job
    _verdict(name, _decision_windows(W, Dn, make, np.arange(W.T)), Dn, helpers.BRANCH_JUMP_MAX, list_name=name)


@pytest.mark.gpu
def test_decision_windows_config4_512_graphs_gpu():
    """BASELINE config 4 (graph mode): 512 size-stratified graphs of the 4337-graph job x 6 windows; the decisions include the rows
    the three max-pools pick (molecule-like graphs are full of symmetric atoms whose activations tie)."""
    W, Dn = helpers.Windows("config4"), helpers.Decisions("config4")
    assert np.array_equal(W.ids, Dn.ids)
    _verdict("config4", _decision_windows(W, Dn, _config4_job_maker(W), np.arange(W.T)), Dn, helpers.CONFIG4_WINDOW_JUMP, list_name="config4")


# ------------------------------------------------------------------ the full horizon: 300 epochs from the seeded masks ------------------------------------------------------------------
def _full_horizon_verdict(what, Dn, ids, err, gates, pool, cond, jump=None, list_name=None):
    """300 epochs from the seeded masks, no teacher forcing.  cond[k] = the target's conditioning over the WHOLE horizon, measured on the CPU
    alone: the largest of the CPU-vs-CPU deviation after 300 epochs (cond_mask / cond_feat of the fixture) and the three window probes of
    its six windows.  On the calm targets (cond <= 2e-6: nothing amplifies round-off beyond the tolerance anywhere along the trajectory) the
    rule of the windowed test applies to the whole run, with the per-window bound max(1e-5, 4 cond) counted once per 50-epoch window
    passed (nothing resets the engine to the reference's state here, so the windows' deviations add up): decisions identical in all
    300 epochs -> within 6 x that of the reference's ONE output; otherwise the first differing decision, at epoch e, must be one the
    reference takes by less than (2 + e // 50) x that (tie margin + completed windows + the running one).  On the
    other targets the engine's state has left the reference's by more than the tolerance long before a decision differs (Tree-Grid:
    chaotic), so the first difference says nothing - they are reported, and covered window by window above."""
    rows = []
    for k in range(len(ids)):
        fd = Dn.first_disagreement(k, 0, gates[k], None if pool is None else pool[k])
        rows.append(dict(id=int(ids[k]), err=float(err[k]), agree=fd is None, cond=float(cond[k]), calm=bool(cond[k] <= helpers.WIN_FLAG),
                         **({} if fd is None else dict(epoch=int(fd[0]), what=fd[1], margin=float(fd[2])))))
    # without teacher forcing the deviations of the windows passed so far add up: w windows -> w times the per-window bound b
    bound = lambda r, epoch=299: max(TOL, ROUNDOFF_BUDGET * r["cond"]) * (1 + epoch // 50)
    # a decision at epoch e may differ when the reference takes it by less than: the tie margin of the windowed rule (b: what two runs from the
    # SAME state may differ by) + the drift of the completed windows (e // 50 x b) + the running window's own drift up to e (<= b)
    tie_bound = lambda r: max(TOL, ROUNDOFF_BUDGET * r["cond"]) * (2 + r["epoch"] // 50)
    calm = [r for r in rows if r["calm"]]
    same = [r for r in calm if r["agree"]]
    over = [r for r in same if r["err"] > TOL]
    bad = [r for r in same if r["err"] > bound(r)]
    ties = [r for r in calm if not r["agree"]]
    unjust = [r for r in ties if not r["margin"] < tie_bound(r)]
    rest = [r for r in rows if not r["calm"]]
    msg = (f"{what} [300 epochs from the seeds]: {len(rows)} targets, {len(calm)} calm (conditioning over the whole horizon <= 2e-6 on the CPU): every decision of "
           f"all 300 epochs identical to the reference's on {len(same)} of them - {len(same) - len(over)} within 1e-5 of the reference's output, the other {len(over)} within "
           f"4 x their conditioning (worst {max([r['err'] for r in over], default=0.0):.2e}; beyond: {len(bad)}) - and a differing decision on {len(ties)}: "
           f"{len(ties) - len(unjust)} first at a tie of the reference, {int(sum(r['err'] <= TOL for r in ties))} of them within 1e-5 anyway, worst "
           f"{max([r['err'] for r in ties], default=0.0):.2e}; not at a tie: {len(unjust)}.  The other {len(rest)} targets (reported): decisions identical on "
           f"{int(sum(r['agree'] for r in rest))}, within 1e-5 of the reference's output {int(sum(r['err'] <= TOL for r in rest))}, worst {max([r['err'] for r in rest], default=0.0):.2e}")
    print(msg)
    for r in sorted(over, key=lambda r: -r["err"])[:20]:
        print(f"{what}: same decisions, beyond 1e-5: id {r['id']}: {r['err']:.2e} from the reference's output, conditioning {r['cond']:.2e} (bound {bound(r):.1e})")
    for r in sorted([r for r in ties if r["err"] > TOL], key=lambda r: -r["err"])[:40]:
        d = r["what"][0]
        desc = (f"U{d[1] + 1}[{d[2]}][{d[3]}] = {d[4]}" if d[0] == "gate" else f"pool {d[1] + 1} column {d[2]}: row {d[3]} vs {d[4]}, margin {d[5]}")
        print(f"{what}: tie   id {r['id']} (calm, {r['err']:.2e} from the reference's output): first differing decision at epoch {r['epoch']} "
              f"({desc}), largest margin of the reference on the {len(r['what'])} differing decision(s) {r['margin']:.2e}")
    # the other targets beyond the tolerance, each with the epoch and the decision at which the engine first leaves the reference's side (or
    # "none": a smooth drift) and the conditioning that keeps it out of the gate - so that every target a bench line lists beyond 1e-5 has
    # its trace here (syn1 / 464)
    for r in sorted([r for r in rest if r["err"] > TOL], key=lambda r: -r["err"])[:40]:
        if r["agree"]:
            desc = "every decision of all 300 epochs identical to the reference's (a smooth drift)"
        else:
            d = r["what"][0]
            desc = (f"first differing decision at epoch {r['epoch']}: " + (f"U{d[1] + 1}[{d[2]}][{d[3]}] = {d[4]}" if d[0] == "gate" else
                    f"pool {d[1] + 1} column {d[2]}: row {d[3]} vs {d[4]}, margin {d[5]}") + f", the reference's margin on it {r['margin']:.2e}")
        print(f"{what}: other id {r['id']} (conditioning over the horizon {r['cond']:.2e} > 2e-6: not gated here, covered window by window), {r['err']:.2e} from the "
              f"reference's output: {desc}")
    assert not bad, msg + f"; {[(r['id'], r['err'], r['cond']) for r in bad]}"
    assert not unjust, msg + f"; first: {unjust[0]}"
    if jump is not None:
        assert all(r["err"] <= jump for r in calm), msg
    # the calm targets beyond 1e-5 - each explained above - are the "full" part of the committed list the outcome tests consult
    _check_listed(list_name, "full", [(r["id"], _tie_row(r)) for r in over + [t for t in ties if t["err"] > TOL]], lambda k: k)
    return msg


def _horizon_conditioning(name, z_cond_mask, z_cond_feat):
    return helpers.horizon_conditioning(name, z_cond_mask, z_cond_feat)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn1", "syn4", "syn5"])
def test_full_horizon_decisions_node_configs_gpu(name):
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    Dn = helpers.Decisions(name)
    ck = helpers.load_ckpt(name)
    z = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    targets = z["targets"]
    assert np.array_equal(targets, Dn.ids)
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])
    dn = engine.khop_device(graph, targets, 3)
    job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"])
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
    job.launch(Hyper(num_iters=300), trace=True)
    em = job.fetch_edges()
    gates, pool = job.fetch_trace()
    assert np.array_equal(em.eoff, z["eoff"])
    err, ferr, _ = helpers.branch_errors(z, None, em.eoff, em.masked_adj, helpers._sig64(em.feat_mask))
    _full_horizon_verdict(name, Dn, targets, np.maximum(err, ferr), gates, pool, _horizon_conditioning(name, z["cond_mask"], z["cond_feat"]),
                          helpers.BRANCH_JUMP_MAX, list_name=name)


@pytest.mark.gpu
def test_full_horizon_decisions_config4_gpu():
    """BASELINE config 4 at the full horizon on the 512 fixture graphs: no 70 % rule - every graph either takes the reference's side of
    every gate and every max-pool in all 300 epochs (and then, where two CPU implementations agree, ends within 1e-5 of the reference's
    output) or leaves it first at a decision the reference itself takes by less than 1e-5 (the ties of the symmetric atoms)."""
    W, Dn = helpers.Windows("config4"), helpers.Decisions("config4")
    job = _config4_job_maker(W)(np.arange(W.T))
    job.launch(Hyper(num_iters=300), trace=True)
    em = job.fetch_edges()
    gates, pool = job.fetch_trace()
    assert np.array_equal(em.eoff, W.eoff)
    z = W.z
    d = np.abs(em.masked_adj.astype(np.float64) - z["vals"].astype(np.float64))
    err = np.asarray([d[a:b].max() if b > a else 0.0 for a, b in zip(W.eoff[:-1], W.eoff[1:])])
    ferr = np.abs(helpers._sig64(em.feat_mask) - z["feat_sig"].astype(np.float64)).max(1)
    # (jump: the largest move of a graph under a 1-ulp perturbation of its initial mask over the full horizon, measured on the CPU alone by
    #  make_golden_branches.py before any implementation ran - a calm graph that leaves at a pool tie must stay inside it)
    _full_horizon_verdict("config4", Dn, W.ids, np.maximum(err, ferr), gates, pool, _horizon_conditioning("config4", z["cond_mask"], z["cond_feat"]),
                          helpers.CONFIG4_WINDOW_JUMP, list_name="config4")


# ------------------------------------------------------------------ BASELINE config 5: BA-House x100k against the reference's own state ------------------------------------------------------------------
@pytest.mark.gpu
def test_windows_ba100k_route_stratified_targets_against_the_reference_gpu():
    """tests/golden/ba100k_windows.npz: the LIVE reference's ExplainModule (explain.py:582-820) on sparse-BFS sub-graphs of the 99 997-node
    graph, 39 route-stratified targets from n = 6 to n > 4095, its Adam state every 50 epochs and its decisions at every epoch.
    k_sparse_large - the kernel of the scaling workload's largest targets, pinned only to the dense streaming kernels in round 3 - is started
    from the reference's state at every boundary and must reproduce its state 50 epochs later within max(1e-5, 4 c) (c: the CPU-only
    conditioning probes, computed for n <= 700; 0 beyond: plain 1e-5); the targets of the LDS-resident classes run with the decision trace
    and are judged by the rules of this file."""
    from gnn_model_explainer_amd.utils import synthetic
    W, Dn = helpers.Windows("ba100k"), helpers.Decisions("ba100k")
    z = W.z
    ck = helpers.load_ckpt("syn1")
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, ck["sd"])
    graph = engine.device_graph(csr, feat, pred)
    targets = W.ids
    dn = engine.khop_device(graph, targets, 3)
    assert np.array_equal(dn.nb_off.cpu().numpy(), z["nb_off"]) and np.array_equal(dn.nb_flat.cpu().numpy()[:len(z["nb_flat"])], z["nb_flat"])
    assert np.array_equal(dn.rows, z["node_idx_new"])
    lists = dn.lists()
    probe = MaskOptimJob.from_csr(graph, dn, None, label[targets], ck["sd"])
    route = probe.route()
    probe.close()
    print("ba100k routes:", dict(zip(*np.unique(route, return_counts=True))), "sizes", int(z["size"].min()), "...", int(z["size"].max()))
    assert 7 in route and 0 not in route and (z["size"][route == 7] > 4095).any()
    cond = np.maximum(np.maximum(z["cond50"], z["sens50"]), z["noise50"])          # [T][6]; zeros where not probed

    def make(ks):
        job = MaskOptimJob.from_csr(graph, [lists[k] for k in ks], z["node_idx_new"][ks], label[targets[ks]], ck["sd"])
        job.set_masks_raw(engine.init_edge_masks_raw(z["size"][ks], seeds=1000 + targets[ks]))
        return job

    # (a) the targets of the LDS-resident classes: decision trace + the rules above
    res_k = np.nonzero(route != 7)[0]
    rows = []
    job = make(res_k)
    eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[res_k])])
    for w in range(W.W):
        mask_rc, fm, gates, pool = helpers.run_window(job, W.boundary(w, res_k), W.win, trace=True)
        em, ef = helpers.window_errors(eoff, mask_rc, fm, W.boundary(w + 1, res_k))
        rows += [_judge(Dn, k, W.win * w, gates[i], None, max(em[i], ef[i]), targets[k], w, -1, cond[k, w]) for i, k in enumerate(res_k)]
    _verdict("ba100k (LDS-resident classes)", rows, Dn, helpers.BRANCH_JUMP_MAX)
    # (b) k_sparse_large (round 5: its logging form records the decision trace too): the same rules, every window listed
    big_k = np.nonzero(route == 7)[0]
    job = make(big_k)
    assert set(job.route()) == {7}
    eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[big_k])])
    out, rows = [], []
    for w in range(W.W):
        mask_rc, fm, gates, pool = helpers.run_window(job, W.boundary(w, big_k), W.win, trace=True)
        em, ef = helpers.window_errors(eoff, mask_rc, fm, W.boundary(w + 1, big_k))
        out += [(int(targets[k]), int(z["size"][k]), w, float(max(em[i], ef[i])), float(cond[k, w])) for i, k in enumerate(big_k)]
        rows += [_judge(Dn, k, W.win * w, gates[i], None, max(em[i], ef[i]), targets[k], w, -1, cond[k, w]) for i, k in enumerate(big_k)]
    err = np.asarray([o[3] for o in out])
    bound = np.maximum(TOL, ROUNDOFF_BUDGET * np.asarray([o[4] for o in out]))
    print(f"ba100k (k_sparse_large, {len(big_k)} targets, n = {int(z['size'][big_k].min())} ... {int(z['size'][big_k].max())}): {len(out)} windows against the reference's state, "
          f"{int((err <= TOL).sum())} within 1e-5, worst {err.max():.2e}; beyond 1e-5: {[o for o in out if o[3] > TOL][:20]}")
    _verdict("ba100k (k_sparse_large)", rows, Dn, helpers.BRANCH_JUMP_MAX)
    assert (err <= np.maximum(bound, TOL)).all() or all(not r["agree"] for r, e, b in zip(rows, err, bound) if e > b), [o for o in out if o[3] > TOL]
