"""CPU: host-side logic of the drop-in package (data types, index bookkeeping, column-generation
driver, solver seam, partition arithmetic) against the reference goldens.  Nothing here touches a GPU."""
import datetime
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, bits_equal, csr_from, same_csr
import workloads


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ----------------------------------------------------------------------------- data types
def test_parameters_have_the_reference_values():
    from smart_crossover import parameters as P
    want = dict(TOLERANCE_FOR_ARTIFICIAL_VARS=1e-8, TOLERANCE_FOR_REDUCED_COSTS=1e-6, COLUMN_GENERATION_RATIO=2,
                OPTIMAL_FACE_ESTIMATOR=1e-3, OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO=1e-5, PERTURB_THRESHOLD=1e-6,
                CONSTANT_SCALE_FACTOR=1e-2, PRIMAL_DUAL_GAP_THRESHOLD=1e-8, PROJECTOR_THRESHOLD=1e-8,
                PERTURB_UPPER_BOUND=1e6)
    for k, v in want.items():
        assert getattr(P, k) == v, k


def test_output_and_basis_types():
    from smart_crossover.output import Basis, Output
    b = Basis(-np.ones(3), np.zeros(2))
    assert b.vbasis.dtype.kind == "i" and b.cbasis.dtype.kind == "i"
    o = Output(obj_val=1.5, runtime=datetime.timedelta(seconds=2), iter_count=3, bar_iter_count=4)
    assert str(o) == "Output(obj_val=1.5, runtime=0:00:02, iter_count=3, bar_iter_count=4)"
    with pytest.raises(Exception):
        o.x = np.zeros(1)


def test_timer_accumulates():
    from smart_crossover.timer import Timer
    t = Timer()
    t.start_timer()
    t.end_timer()
    first = t.total_duration
    t.accumulate_time(datetime.timedelta(seconds=5))
    assert t.total_duration == first + datetime.timedelta(seconds=5)
    t.clear()
    assert t.total_duration == datetime.timedelta(0)


def test_formats_validation_and_structure(g1, g4):
    from smart_crossover.formats import GeneralLP, MinCostFlow, OptTransport, StandardLP
    A = csr_from(g1, "A")
    lp = GeneralLP(A, g1["b"], g1["c"], g1["l"], g1["u"], g1["sense"])
    assert same_csr(lp.get_standard_A(), csr_from(g1, "Astd")) and bits_equal(lp.get_standard_c(), g1["std_c"])
    assert lp.get_free_ind().size == 0 and lp.get_nonfree_ind().size == lp.get_standard_c().size
    cp = lp.copy()
    assert cp.A is not lp.A and cp.c is not lp.c and bits_equal(cp.c, lp.c)
    with pytest.raises(AssertionError):
        GeneralLP(A, g1["b"], g1["c"], g1["l"], g1["u"], np.full(g1["b"].size, ">"))
    free = GeneralLP(A, g1["b"], g1["c"], np.where(np.arange(A.shape[1]) == 3, -np.inf, 0.0), np.full(A.shape[1], np.inf),
                     g1["sense"])
    assert np.array_equal(free.get_free_ind(), [3])
    s = StandardLP(A, g1["b"], g1["c"], g1["u"])
    assert np.array_equal(s.l, np.zeros_like(g1["u"])) and np.all(s.to_general().sense == "=")
    with pytest.raises(ValueError):
        MinCostFlow(A=A, b=np.ones(A.shape[0]), c=g1["c"], u=g1["u"])
    with pytest.raises(ValueError):
        OptTransport(np.ones(3), np.ones(4), np.ones((3, 4)))
    ot = OptTransport(g4["s"], g4["d"], g4["M"])
    mcf = ot.to_MCF()
    ref = csr_from(g4, "Amcf")
    assert sp.isspmatrix_csr(mcf.A) and mcf.A.shape == ref.shape and (mcf.A != ref).nnz == 0
    assert bits_equal(mcf.b, g4["b_mcf"]) and bits_equal(mcf.c, g4["c_mcf"])


# ----------------------------------------------------------------------------- LPManager (host parts)
@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_lp_manager_partition_and_recovery(gname, request):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods.lp_manager import LPManager
    from smart_crossover.output import Basis
    g = request.getfixturevalue(gname)
    lp = GeneralLP(csr_from(g, "A"), g["b"], g["c"], g["l"], g["u"], g["sense"].copy())
    mgr = LPManager(lp)
    assert np.array_equal(mgr.var_info["non_fix"], np.arange(lp.c.size)) and mgr.get_num_fixed_variables() == 0
    mgr.fix_variables(ind_fix_to_low=g["fix_low"], ind_fix_to_up=g["fix_up"])
    mgr.fix_constraints(g["fixed_rows"])
    for key in ("fix_low", "fix_up", "non_fix", "fix"):
        assert np.array_equal(mgr.var_info[key], g[key]) and mgr.var_info[key].dtype == np.int64
    assert mgr.get_num_fixed_variables() == g["fix"].size and mgr.get_num_fixed_constraints() == g["fixed_rows"].size
    assert bits_equal(mgr.get_subx(g["x"]), g["sub_of_x"])
    assert bits_equal(mgr.recover_x_from_sub_x(g["x_sub"]), g["recover_x"])      # quirk Q1 inside
    assert bits_equal(mgr.get_orix(g["x_sub"]), g["orix"])
    rb = mgr.recover_basis_from_sub_basis(Basis(g["vb_sub"], g["cb_sub"]))
    assert np.array_equal(rb.vbasis, g["recover_vb"]) and np.array_equal(rb.cbasis, g["recover_cb"])
    # nothing fixed -> lp_sub aliases lp, sense of fixed rows rewritten in place (no device needed)
    plain = LPManager(lp)
    plain.fix_constraints(g["fixed_rows"])
    plain.update_subproblem()
    assert plain.lp_sub is lp and np.all(lp.sense[g["fixed_rows"]] == "=")


def test_perturb_direction_and_global_rng_side_effect():
    from smart_crossover.lp_methods.algorithms import _perturb_direction
    from oracle import lp_path as L
    for _ in range(2):                                     # second round comes from the cache
        np.random.seed(123)
        xi = _perturb_direction(1000)
        assert bits_equal(xi, L.xi_vector(1000))
        after = np.random.random(2)
        np.random.seed(42)
        np.random.uniform(0.9, 1, 1000)
        assert np.array_equal(after, np.random.random(2))


# ----------------------------------------------------------------------------- network managers (host parts)
def test_mcf_manager_index_bookkeeping(g3):
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.net_manager import MCFManagerStd
    from smart_crossover.output import Basis
    mcf = MinCostFlow(A=csr_from(g3, "A"), b=g3["b"].copy(), c=g3["c"].copy(), u=g3["u"].copy())
    mgr = MCFManagerStd(mcf)
    mgr.rescale_cost(np.max(np.abs(mcf.c)))
    assert bits_equal(mcf.c, g3["c_scaled"]) and mgr.recover_obj_val(2.0) == 2.0 * float(g3["factor"])
    mgr.fix_variables(ind_fix_to_up=g3["fix_up0"], ind_fix_to_low=g3["fix_low0"])
    assert mgr.var_info["non_fix"].size == 0 and mgr.var_info["fix"].size == mcf.c.size          # quirk Q9
    # emulate the extension's effect on the index sets, then release the first batch
    V, E = mcf.A.shape
    mgr.var_info["non_fix"] = np.append(mgr.var_info["non_fix"], np.arange(E, E + V, dtype=np.int64))
    mgr.mcf = MinCostFlow(csr_from(g3, "A1"), g3["b1"], g3["c1"], g3["u1"])
    lft, rgt = (int(v) for v in g3["batch0"])
    mgr.add_free_variables(g3["queue_ref"][lft:rgt])
    assert np.array_equal(mgr.var_info["non_fix"], g3["non_fix1"])
    assert np.array_equal(mgr.var_info["fix_up"], g3["fix_up1"]) and np.array_equal(mgr.var_info["fix_low"], g3["fix_low1"])
    assert bits_equal(mgr.recover_x_from_sub_x(g3["x_sub"]), g3["recover_x"])
    rb = mgr.recover_basis_from_sub_basis(Basis(g3["vb_sub"], g3["cb_sub"]))
    assert np.array_equal(rb.vbasis, g3["recover_vb"])
    mgr.n, mgr.m = E, V
    mgr.var_info["fix_up"] = g3["fix_up0"]
    mgr.set_initial_basis()
    assert np.array_equal(mgr.basis.vbasis, g3["vb0"]) and np.array_equal(mgr.basis.cbasis, g3["cb0"])


def test_ot_manager_host_parts_and_tree_basis(g4):
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.net_manager import OTManager
    from smart_crossover.network_methods import tree_BI
    ot = OptTransport(g4["s"].copy(), g4["d"].copy(), g4["M"].copy())
    mgr = OTManager(ot)
    mgr.get_mcf()
    # K14 + K15 are host steps; the spanning tree itself (K13) is a device kernel, checked in the GPU tests
    vbasis, pushes = tree_BI.push_tree_to_bfs(mgr, g4["tree_edges"].astype(np.int64))
    assert np.array_equal(vbasis, g4["tree_vb"].astype(int)) and pushes == int(g4["push_iter"])
    from smart_crossover.output import Basis
    basis = Basis(vbasis, np.concatenate([-np.ones(mgr.m - 1), np.array([0])]))
    assert np.array_equal(basis.cbasis, g4["tree_cb"].astype(int))
    mgr.add_free_variables(basis.vbasis == 0)                       # quirk Q10: boolean mask on the flat problem
    assert np.array_equal(np.flatnonzero(mgr.mask_sub_ot), np.flatnonzero(basis.vbasis == 0))
    ext = OTManager(OptTransport(g4["s"].copy(), g4["d"].copy(), g4["M"].copy()))
    ext.extend_by_bigM(ext.m * np.max(g4["M"]))
    ext.set_initial_basis()
    assert bits_equal(ext.ot.M, g4["M1"]) and np.array_equal(ext.mask_sub_ot, g4["mask1"])
    assert np.array_equal(ext.basis.vbasis, g4["vb0"].astype(int)) and np.array_equal(ext.basis.cbasis, g4["cb0"].astype(int))
    lft, rgt = (int(v) for v in g4["batch0"])
    ext.add_free_variables(g4["queue_ref"][lft:rgt])
    assert np.array_equal(ext.mask_sub_ot, g4["mask2"])
    x_sub = np.arange(int(ext.mask_sub_ot.sum()), dtype=float)
    x = ext.recover_x_from_sub_x(x_sub)
    assert np.array_equal(x[ext.mask_sub_ot.ravel()], x_sub) and x.size == ext.mask_sub_ot.size


def test_tree_solver_rejects_disconnected_support():
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.net_manager import OTManager
    from smart_crossover.network_methods import tree_BI
    ot = OptTransport(np.array([0.5, 0.5]), np.array([0.5, 0.5]), np.ones((2, 2)))
    mgr = OTManager(ot)
    mgr.get_mcf()
    with pytest.raises(ValueError):          # a 2-arc forest on 4 nodes: arcs (0,0) and (1,1)
        tree_BI.push_tree_to_bfs(mgr, np.array([0, 3], dtype=np.int64))


# ----------------------------------------------------------------------------- column generation driver
class FakeManager:
    def __init__(self, m, n, rounds):
        self.m, self.n, self.rounds, self.slices, self.basis, self._left = m, n, rounds, [], None, 0

    def add_free_variables(self, idx):
        self.slices.append([self._left, self._left + len(idx)])
        self._left += len(idx)

    def update_subproblem(self):
        pass

    def solve_subproblem(self, solver, settings):
        from smart_crossover.output import Basis, Output
        return Output(x=np.zeros(1), y=np.zeros(1), obj_val=1.0, runtime=datetime.timedelta(seconds=1), iter_count=3,
                      basis=Basis(np.zeros(1), np.zeros(1)), status="OPTIMAL")

    def recover_obj_val(self, v):
        return 10 * v

    def set_basis(self, b):
        self.basis = b

    def recover_basis_from_sub_basis(self, b):
        return b

    def recover_x_from_sub_x(self, x):
        return x

    def check_optimality_condition(self, x, y):
        return len(self.slices) >= self.rounds


def test_column_generation_schedule_matches_reference():
    from smart_crossover.network_methods.algorithms import column_generation
    with open(os.path.join(GOLDEN, "g5_cg_schedule.json")) as f:
        cases = json.load(f)
    for cs in cases:
        fm = FakeManager(cs["m"], cs["n"], cs["rounds"])
        buf = io.StringIO()
        with redirect_stdout(buf):
            out = column_generation(fm, np.arange(cs["qlen"]), "HGS", None)
        assert fm.slices == [list(s) for s in cs["slices"]], cs
        assert out.iter_count == cs["iter_count"] and out.obj_val == 10.0
        assert out.runtime >= datetime.timedelta(seconds=len(fm.slices))          # solver runtimes are accumulated
        exhausted = len(fm.slices) < cs["rounds"]
        assert ("Column generation fails!" in buf.getvalue()) == exhausted


# ----------------------------------------------------------------------------- solver seam
def test_solver_seam_with_highs_backend():
    from smart_crossover.formats import GeneralLP, MinCostFlow, OptTransport
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import (generate_solver_caller, solve_lp, solve_mcf, solve_ot, solve_problem)
    s = SolverSettings()
    assert (s.presolve, s.crossover, s.barrierTol, s.optimalityTol, s.timeLimit, s.log_file, s.log_console, s.iterLimit,
            s.simplexPricing) == ("on", "on", 1e-8, 1e-6, 3600, "", 1, 1000, "")
    inst = workloads.config1()
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    quietly = SolverSettings(log_console=0)
    bar = solve_lp(lp, "HGS", "barrier", SolverSettings(crossover="off", log_console=0))
    assert bar.status == "OPTIMAL" and bar.basis is None and bits_equal(bar.x, bar.x_bar) and bar.bar_iter_count > 0
    rc = lp.c - lp.A.T @ bar.y                                              # y follows the c - A^T y convention:
    interior = (bar.x > lp.l + 1e-3) & (bar.x < lp.u - 1e-3)                # columns strictly inside their bounds
    assert interior.any() and np.all(np.abs(rc[interior]) < 1e-5)           # have a vanishing reduced cost
    cross = solve_lp(lp, "HGS", "barrier", quietly)
    assert cross.basis is not None
    assert np.count_nonzero(cross.basis.vbasis == 0) + np.count_nonzero(cross.basis.cbasis == 0) == lp.b.size
    warm = solve_lp(lp, "HGS", "primal_simplex", quietly, warm_start_basis=cross.basis, warm_start_solution=(cross.x, cross.y))
    assert warm.status == "OPTIMAL" and warm.iter_count == 0 and warm.obj_val == pytest.approx(cross.obj_val, rel=1e-9)
    for method in ("default", "simplex", "dual_simplex", "network_simplex"):
        assert solve_lp(lp, "HGS", method, quietly).obj_val == pytest.approx(cross.obj_val, rel=1e-8)
    with pytest.raises(ValueError):
        solve_lp(lp, "HGS", "interior", quietly)
    with pytest.raises(ValueError):
        solve_lp(object(), "HGS")
    with pytest.raises(ValueError):
        generate_solver_caller("XPRESS")
    with pytest.raises(ImportError):
        generate_solver_caller("GRB")
    # infeasible LP -> status only
    bad = GeneralLP(sp.csr_matrix(np.array([[1.0], [1.0]])), np.array([1.0, 2.0]), np.ones(1), np.zeros(1), np.full(1, np.inf),
                    np.array(["=", "="]))
    out = solve_lp(bad, "HGS", "default", quietly)
    assert out.status == "INFEASIBLE" and out.x is None and out.runtime is not None
    # MCF / OT wrappers
    mi = workloads.mcf(30, 150, seed=4)
    mo = solve_mcf(MinCostFlow(A=mi.A, b=mi.b, c=mi.c, u=mi.u), "HGS", "default", quietly)
    assert mo.status == "OPTIMAL" and np.allclose(mi.A @ mo.x, mi.b, atol=1e-7)
    oi = workloads.ot(6, 7, seed=2)
    oo = solve_ot(OptTransport(oi.s, oi.d, oi.M), "HGS", "default", quietly)
    assert oo.status == "OPTIMAL" and np.allclose(oo.x.reshape(6, 7).sum(axis=1), oi.s, atol=1e-8)
    caller = generate_solver_caller("HGS", quietly)
    caller.read_genlp(lp)
    g = caller.return_genlp()
    assert g.A.shape == lp.A.shape and set(np.unique(g.sense)) <= {"=", "<"}
    assert solve_problem(caller, "default", quietly).status == "OPTIMAL"


# ----------------------------------------------------------------------------- partition arithmetic
def test_distributed_partition_and_price_reduction():
    from smart_crossover import distributed as D
    blocks = D.split_even(10, 4)
    assert [(b.start, b.stop) for b in blocks] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [b.size for b in D.split_even(2, 4)] == [1, 1, 0, 0]
    ptr = np.array([0, 10, 10, 11, 40, 41, 41, 80])
    parts = D.split_by_nnz(ptr, 3)
    assert parts[0].start == 0 and parts[-1].stop == 7 and all(a.stop == b.start for a, b in zip(parts, parts[1:]))
    loads = [int(ptr[p.stop] - ptr[p.start]) for p in parts]
    assert sum(loads) == 80 and max(loads) <= 45
    rec = D.unpack_price(D.pack_price(-1.5, 7, 3))
    assert rec == (-1.5, 7, 3)
    best = D.reduce_price_records([(-1.0, 5, 2), (float("nan"), -1, 1), (-1.0, 0, 0), (-3.0, 2, 4)], [0, 100, 200, 300])
    assert best == (-3.0, 302, 7)
    tie = D.reduce_price_records([(-2.0, 9, 0), (-2.0, 1, 0)], [50, 10])
    assert tie == (-2.0, 11, 0)
    none = D.reduce_price_records([(0.0, -1, 0)], [0])
    assert none[1] == -1 and np.isnan(none[0])


MPS_TINY = """NAME          TINY
ROWS
 N  COST
 L  LIM1
 G  LIM2
 E  MYEQN
 L  RNG
COLUMNS
    X         COST         1.0   LIM1         1.0
    X         LIM2         1.0   RNG          1.0
    Y         COST         2.0   LIM1         1.0
    Y         MYEQN       -1.0   RNG          1.0
    Z         COST        -1.0   MYEQN        1.0
RHS
    RHS       LIM1         4.0   LIM2         1.0
    RHS       MYEQN        7.0   RNG          3.0
RANGES
    RNG       RNG          2.5
BOUNDS
 UP BND       X            4.0
 LO BND       Y           -1.0
 UP BND       Y            1.0
ENDATA
"""


def test_mps_file_to_general_lp_row_form(tmp_path):
    """read_model_from_file + return_genlp (reference: gurobi.py:25-29 + caller.py:124-126): only '=' and
    '<' rows come back -- a 'G' row negated, a ranged row split in two -- and the LP solves to the same
    optimum through the file, through the returned GeneralLP and through the device-free HiGHS path."""
    from smart_crossover.solver_caller.solving import generate_solver_caller, solve_lp
    from smart_crossover.solver_caller.caller import SolverSettings
    path = tmp_path / "tiny.mps"
    path.write_text(MPS_TINY)
    quiet = SolverSettings(log_console=0)
    caller = generate_solver_caller("HGS", quiet)
    caller.read_model_from_file(str(path))
    lp = caller.return_genlp()
    assert set(lp.sense) <= {"=", "<"}
    assert lp.A.shape == (5, 3) and list(lp.sense).count("=") == 1       # E + L + G + 2 x ranged
    dense = {tuple(np.round(r, 12)) + (round(float(b), 12), s) for r, b, s in zip(lp.A.toarray(), lp.b, lp.sense)}
    assert (0.0, -1.0, 1.0, 7.0, "=") in dense                           # MYEQN
    assert (1.0, 1.0, 0.0, 4.0, "<") in dense                            # LIM1
    assert (-1.0, 0.0, 0.0, -1.0, "<") in dense                          # LIM2:  x >= 1  ->  -x <= -1
    assert (1.0, 1.0, 0.0, 3.0, "<") in dense and (-1.0, -1.0, 0.0, -0.5, "<") in dense   # 0.5 <= x + y <= 3
    assert np.array_equal(lp.l, [0.0, -1.0, 0.0]) and np.array_equal(lp.u, [4.0, 1.0, np.inf])
    caller.run_default()
    out_file = caller.return_output()
    out_lp = solve_lp(lp, solver="HGS", method="default", settings=quiet)
    assert out_file.status == out_lp.status == "OPTIMAL"
    # min x + 2y - z with z = 7 + y, i.e. min x + y - 7 over x + y >= 0.5, x >= 1, y >= -1: optimum -6.5
    # on the face x + y = 0.5 (not a unique vertex)
    assert abs(out_file.obj_val - (-6.5)) < 1e-9 and abs(out_lp.obj_val - (-6.5)) < 1e-9
    x, y, z = out_lp.x
    assert abs(x + y - 0.5) < 1e-9 and abs(z - (7.0 + y)) < 1e-9 and x >= 1 - 1e-9 and y >= -1 - 1e-9


def test_gurobi_style_summary_lines_in_log(tmp_path):
    """The two sentences the reference's log analysis parses (visualization.py:31-34,345-346) are appended
    to SolverSettings.log_file by every backend run."""
    import re
    from smart_crossover.solver_caller.solving import solve_lp
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.formats import GeneralLP
    inst = workloads.config1()
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    log = tmp_path / "run.log"
    st = SolverSettings(log_console=0, log_file=str(log))
    out_b = solve_lp(lp, solver="HGS", method="barrier", settings=st)
    out_s = solve_lp(lp, solver="HGS", method="primal_simplex", settings=st)
    assert out_b.status == out_s.status == "OPTIMAL"
    text = log.read_text()
    bar = re.findall(r"Barrier solved model in (\d+) iterations and (\d+\.\d+) seconds", text)
    spx = re.findall(r"Solved in (\d+) iterations and (\d+\.\d+) seconds", text)
    assert len(bar) == 1 and int(bar[0][0]) == out_b.bar_iter_count
    assert len(spx) == 2 and int(spx[1][0]) == out_s.iter_count


def test_barrier_without_crossover_returns_a_consistent_interior_point():
    """The input of the perturbation crossover: HiGHS ends such runs "Unknown" on larger models although
    IPX converged; the stand-in certifies the interior point itself (KKT residuals + duality gap) and
    solves the unpresolved model, because the presolved run hands back inconsistent column values."""
    from smart_crossover.formats import GeneralLP
    from smart_crossover.solver_caller.solving import solve_lp
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.sparse_lp(300, 1500, 5, seed=22, frac_upper=0.3)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    out = solve_lp(lp, "HGS", method="barrier", settings=SolverSettings(crossover="off", log_console=0))
    assert out.status == "OPTIMAL" and out.basis is None and out.bar_iter_count > 0
    assert float(lp.c @ out.x) == pytest.approx(out.obj_val, rel=1e-9)
    ref = solve_lp(lp, "HGS", method="default", settings=SolverSettings(log_console=0))
    assert out.obj_val == pytest.approx(ref.obj_val, rel=1e-6)
    assert np.all(lp.A @ out.x <= lp.b + 1e-5) and np.all(out.x >= lp.l - 1e-7) and np.all(out.x <= lp.u + 1e-7)
    # an interior point: the variables that vanish at the vertex are tiny but strictly inside their bounds
    assert np.count_nonzero((out.x > lp.l) & (out.x < lp.l + 1e-6)) > 100


def test_host_thread_pools_are_only_ever_cut(monkeypatch):
    """smart_crossover/hip/host_threads.py: the CPU quota is read from the cgroup / affinity mask, BLAS / OpenMP pools
    larger than a quarter of it are cut, smaller ones stay, SX_BLAS_THREADS=0 leaves everything alone."""
    threadpoolctl = pytest.importorskip("threadpoolctl")
    from smart_crossover.hip import host_threads as ht
    quota = ht.cpu_quota()
    assert 1 <= quota <= (os.cpu_count() or 1)
    before = {p["filepath"]: p["num_threads"] for p in threadpoolctl.threadpool_info()}
    try:
        monkeypatch.setenv("SX_BLAS_THREADS", "0")
        rec = ht.fit_to_quota(force=True)
        assert rec["limit"] is None and rec["pools"] == []
        assert {p["filepath"]: p["num_threads"] for p in threadpoolctl.threadpool_info()} == before
        monkeypatch.setenv("SX_BLAS_THREADS", "auto")
        rec = ht.fit_to_quota(force=True)
        limit = max(1, quota // 4)
        for p in threadpoolctl.threadpool_info():
            assert p["num_threads"] == min(before[p["filepath"]], limit)
        monkeypatch.setenv("SX_BLAS_THREADS", "1")
        ht.fit_to_quota(force=True)
        assert all(p["num_threads"] == 1 for p in threadpoolctl.threadpool_info())
        monkeypatch.setenv("SX_BLAS_THREADS", "auto")      # (never raises a pool again)
        ht.fit_to_quota(force=True)
        assert all(p["num_threads"] == 1 for p in threadpoolctl.threadpool_info())
    finally:
        for lib in threadpoolctl.ThreadpoolController().lib_controllers:
            if lib.filepath in before:
                lib.set_num_threads(before[lib.filepath])
