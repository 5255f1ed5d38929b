"""The executed-work model behind bench.py's `roofline` (gnn_model_explainer_amd/utils/work_model.py): structure counts on a hand-checked
sub-graph, and the properties that make `frac` a fraction (every term a lower bound that grows with the work; critical path vs saturated regime)."""
import numpy as np

from gnn_model_explainer_amd.utils import work_model as wm


def _path_plus_triangle():
    # nodes 0-1-2-3-4 a path, plus the triangle 4-5-6; target row 0.  Upper-triangle edges, row-major.
    rc = np.asarray([[0, 1], [1, 2], [2, 3], [3, 4], [4, 5], [4, 6], [5, 6]], np.int32)
    return np.asarray([7]), np.asarray([0, len(rc)], np.int64), rc, np.asarray([0])


def test_target_structure_counts_rows_and_entries_by_hop():
    n, eoff, rc, rows = _path_plus_triangle()
    S = wm.target_structure(n, eoff, rc, rows)
    assert S["nnz"][0] == 14 and S["edges"][0] == 7 and S["deg_t"][0] == 1
    assert S["rowsB"][0] == 2          # {0, 1}
    assert S["rowsA"][0] == 3          # {0, 1, 2}
    assert S["nnzB"][0] == 1 + 2 and S["nnzA"][0] == 1 + 2 + 2
    # entries (r, c) with r in A and c in B: (0,1), (1,0), (2,1)
    assert S["nBinA"][0] == 3
    Sg = wm.target_structure(n, eoff, rc, rows, graph_mode=True)
    assert Sg["rowsA"][0] == 7 and Sg["nnzA"][0] == 14


def test_bounds_grow_with_the_work_and_switch_regime_when_saturated():
    n, eoff, rc, rows = _path_plus_triangle()
    S = wm.target_structure(n, eoff, rc, rows)
    lat = dict(wm.DEFAULT_LATENCY_NS)
    for xc in (0, 2):
        fl = wm.executed_flops_per_iter(S, 10, 20, 20, 4, xc)
        by = wm.executed_lds_bytes_per_iter(S, 10, 20, 20, 4, xc)
        ch = wm.chain_ns_per_iter(wm.chain_ops_per_iter(S, 10, 20, 20, 4, xc), lat)
        assert fl[0] > 0 and by[0] > 0 and 500.0 < ch[0] < 20000.0      # a chain of ~100 dependent operations: microseconds, not nanoseconds
    # the algebraic constant-feature form executes less than the general one
    assert wm.executed_flops_per_iter(S, 10, 20, 20, 4, 2)[0] < wm.executed_flops_per_iter(S, 10, 20, 20, 4, 0)[0]
    # one workgroup: the launch is its chain; 10 000 workgroups on 256 CUs: the sum of the chains over the resident slots
    one = wm.launch_bounds(fl, by, ch, 300, np.zeros(1, np.int64))
    assert abs(one["chain_s"] - 300 * ch[0] * 1e-9) < 1e-12 and one["busy_cus"] == 1
    T = 10000
    many = wm.launch_bounds(np.repeat(fl, T), np.repeat(by, T), np.repeat(ch, T), 300, np.arange(T), wgs_per_cu=2)
    assert abs(many["chain_s"] - 300 * ch[0] * 1e-9 * T / 512) < 1e-9 and many["busy_cus"] == 256
    # targets that share a workgroup run concurrently: the workgroup's chain is its slowest target's, its flops add up
    shared = wm.launch_bounds(np.repeat(fl, 8), np.repeat(by, 8), np.repeat(ch, 8), 300, np.zeros(8, np.int64))
    assert abs(shared["chain_s"] - one["chain_s"]) < 1e-12 and abs(shared["flops_s"] - 8 * one["flops_s"]) < 1e-12
