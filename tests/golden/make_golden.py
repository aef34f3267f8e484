#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: needs /root/reference (which never travels to the GPU
box).  The reference's numpy/scipy host code is imported unmodified.  Because
five of its modules import gurobipy / cplex / mosek at module top and those
packages are not installed, empty placeholder modules with those names are
put in ``sys.modules`` first -- they only let the import statements resolve;
no solver behaviour is provided and nothing that would run inside a solver is
captured.  Two compatibility shims cover API drift between the reference's
pinned numpy 1.21 / scipy 1.7 and this image (``np.Inf``; ``cg(tol=)`` ->
``cg(rtol=, atol=0)``, i.e. the same relative stopping rule).

Outputs (all data, no source):  g1_lp_afiro.npz, g2_lp_small.npz,
g3_mcf_small.npz, g4_ot_small.npz, g5_cg_schedule.json, g6_control_flow.json, digests.json.

Usage:  python tests/golden/make_golden.py
"""
import hashlib
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

for _name, _attrs in {"gurobipy": ("GRB", "Model"), "cplex": ("Cplex",), "mosek": (), "mosek.fusion": ("Model",)}.items():
    _mod = types.ModuleType(_name)
    for _a in _attrs:
        setattr(_mod, _a, type(_a, (), {}))
    sys.modules[_name] = _mod
sys.modules["mosek"].fusion = sys.modules["mosek.fusion"]
if not hasattr(np, "Inf"):
    np.Inf = np.inf
_cg_now = spl.cg


def _cg_compat(A, b, x0=None, tol=None, maxiter=None, **kw):
    if tol is not None:
        kw.setdefault("rtol", tol)
        kw.setdefault("atol", 0.0)
    return _cg_now(A, b, x0=x0, maxiter=maxiter, **kw)


spl.cg = _cg_compat

import smart_crossover.lp_methods.algorithms as ref_alg            # noqa: E402
import smart_crossover.network_methods.algorithms as ref_nalg      # noqa: E402
import smart_crossover.network_methods.net_manager as ref_nm       # noqa: E402
import smart_crossover.network_methods.tree_BI as ref_tree         # noqa: E402
from smart_crossover.formats import GeneralLP, MinCostFlow, OptTransport  # noqa: E402
from smart_crossover.output import Basis, Output                   # noqa: E402
import datetime                                                    # noqa: E402

import workloads                                                   # noqa: E402


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def csr_parts(A, prefix):
    A = sp.csr_matrix(A)
    return {prefix + "_data": A.data, prefix + "_indices": A.indices.astype(np.int32),
            prefix + "_indptr": A.indptr.astype(np.int64), prefix + "_shape": np.array(A.shape, dtype=np.int64)}


# ------------------------------------------------------------------ LP
def lp_case(inst, gamma=1e-3, gamma_dual=1e-3, sub_seed=11):
    lp = GeneralLP(inst.A.copy(), inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    out = {}
    out.update(csr_parts(inst.A, "A"))
    for k in ("b", "c", "l", "u", "x", "y"):
        out[k] = getattr(inst, k)
    out["sense"] = inst.sense.astype("U1")
    out["gamma"] = np.array([gamma, gamma_dual])
    out["s_d"] = lp.get_dual_slack(inst.y)
    out["s_p"] = lp.get_primal_slack(inst.x)
    out["std_x_of_x"] = lp.get_standard_x(inst.x)
    out["std_c"] = lp.get_standard_c()
    out.update(csr_parts(lp.get_standard_A(), "Astd"))

    seen = {}
    real_proj = ref_alg.get_projector_Xc
    real_apply = ref_alg.apply_projector

    def rec_proj(lp_, x_real):
        seen["x_real"] = x_real.copy()
        p = real_proj(lp_, x_real)
        seen["proj"] = p.copy()
        return p

    def rec_apply(Y, v, tol=1e-8, max_iter=1000):
        iters = [0]
        Yv = Y @ v
        z, info = _cg_now(Y @ Y.T, Yv, rtol=tol, atol=0.0, maxiter=max_iter,
                          callback=lambda xk: iters.__setitem__(0, iters[0] + 1))
        seen["cg_iters"] = iters[0]
        seen["cg_info"] = info
        return real_apply(Y, v, tol, max_iter)

    ref_alg.get_projector_Xc = rec_proj
    ref_alg.apply_projector = rec_apply
    try:
        for feas in (False, True):
            mgr = quiet(ref_alg.get_perturb_problem, lp, inst.x, inst.y, gamma, gamma_dual, feas)
            tag = "feas" if feas else "opt"
            out[f"c_pt_{tag}"] = mgr.lp.c.copy()
            if not feas:
                out["fix_low"] = mgr.var_info["fix_low"].astype(np.int64)
                out["fix_up"] = mgr.var_info["fix_up"].astype(np.int64)
                out["non_fix"] = mgr.var_info["non_fix"].astype(np.int64)
                out["fix"] = mgr.var_info["fix"].astype(np.int64)
                out["fixed_rows"] = mgr.fixed_constraints.astype(np.int64)
                out.update(csr_parts(mgr.lp_sub.A, "Asub"))
                out["b_sub"], out["c_sub"] = mgr.lp_sub.b, mgr.lp_sub.c
                out["l_sub"], out["u_sub"] = mgr.lp_sub.l, mgr.lp_sub.u
                out["sense_sub"] = mgr.lp_sub.sense.astype("U1")
                out["x_real"] = seen["x_real"]
                out["proj_norm"] = np.array(np.linalg.norm(seen["proj"]))
                out["sf"] = np.array(ref_alg.get_scale_factor(seen["proj"], lp.c.size + np.count_nonzero(lp.sense == "<")))
                out["cg_iters"] = np.array(seen["cg_iters"])
                out["cg_info"] = np.array(seen["cg_info"])
                rng = np.random.default_rng(sub_seed)
                nsub = mgr.var_info["non_fix"].size
                x_sub = rng.random(nsub)
                vb_sub = rng.integers(-2, 1, size=nsub)
                cb_sub = rng.integers(-1, 1, size=lp.b.size)
                out["x_sub"], out["vb_sub"], out["cb_sub"] = x_sub, vb_sub, cb_sub
                out["sub_of_x"] = mgr.get_subx(inst.x)
                out["recover_x"] = mgr.recover_x_from_sub_x(x_sub)
                out["orix"] = mgr.get_orix(x_sub)
                rb = mgr.recover_basis_from_sub_basis(Basis(vb_sub, cb_sub))
                out["recover_vb"], out["recover_cb"] = rb.vbasis, rb.cbasis
                obj = float(lp.c @ mgr.get_orix(x_sub))
                ok = quiet(ref_alg.check_perturb_output_precision, mgr, x_sub, lp.c, obj * (1 + 1e-12))
                bad = quiet(ref_alg.check_perturb_output_precision, mgr, x_sub, lp.c, obj * 1.5 + 1.0)
                out["gap_flags"] = np.array([ok is True, bad is None])
    finally:
        ref_alg.get_projector_Xc = real_proj
        ref_alg.apply_projector = real_apply
    out["x_min_raw"] = ref_alg.get_x_perturb_val(lp, inst.x)
    return out


# ------------------------------------------------------------------ MCF
def mcf_case(inst, y_seed=5):
    mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
    out = {}
    out.update(csr_parts(inst.A, "A"))
    for k in ("b", "c", "u", "x"):
        out[k] = getattr(inst, k)
    mgr = ref_nm.MCFManagerStd(mcf)
    with np.errstate(all="ignore"):
        queue, ind = mgr.get_sorted_flows(inst.x)
    out["ind"], out["queue_ref"] = ind, queue.astype(np.int64)
    mgr.rescale_cost(np.max(np.abs(mcf.c)))
    out["c_scaled"] = mgr.mcf.c.copy()
    out["factor"] = np.array(mgr.c_rescaling_factor)
    up = np.where(inst.x >= mcf.u / 2)[0]
    low = np.where(inst.x < mcf.u / 2)[0]
    mgr.fix_variables(ind_fix_to_up=up, ind_fix_to_low=low)
    out["fix_up0"], out["fix_low0"] = up.astype(np.int64), low.astype(np.int64)
    bigM = mgr.m * np.max(mcf.c)          # mcf.c was rebound by rescale_cost (quirk Q7)
    out["bigM"] = np.array(bigM)
    mgr.extend_by_bigM(bigM)
    out.update(csr_parts(mgr.mcf.A, "A1"))
    out["b1"], out["c1"], out["u1"] = mgr.mcf.b, mgr.mcf.c, mgr.mcf.u
    out["artificial"] = mgr.artificial_vars.astype(np.int64)
    mgr.update_subproblem()
    mgr.set_initial_basis()
    out["vb0"], out["cb0"] = mgr.basis.vbasis, mgr.basis.cbasis
    out.update(csr_parts(mgr.mcf_sub.A, "Asub0"))
    out["bsub0"] = mgr.mcf_sub.b
    # release the first batch exactly like column_generation does
    m_ext, n_ext = mgr.m, mgr.n
    batch = int(10 * m_ext) if n_ext / m_ext > 1000 else int(1.2 * m_ext)
    right = min(batch, len(queue))
    mgr.add_free_variables(queue[0:right])
    mgr.update_subproblem()
    out["batch0"] = np.array([0, right])
    out["non_fix1"] = mgr.var_info["non_fix"].astype(np.int64)
    out["fix_up1"] = mgr.var_info["fix_up"].astype(np.int64)
    out["fix_low1"] = mgr.var_info["fix_low"].astype(np.int64)
    out.update(csr_parts(mgr.mcf_sub.A, "Asub1"))
    out["bsub1"], out["csub1"], out["usub1"] = mgr.mcf_sub.b, mgr.mcf_sub.c, mgr.mcf_sub.u
    rng = np.random.default_rng(y_seed)
    y = rng.standard_normal(mgr.mcf.b.size) * 0.01
    out["y"] = y
    out["rc"] = mgr.get_reduced_cost_for_original_mcf(y)
    xs = np.zeros(mgr.mcf.c.size)
    out["opt_flag"] = np.array(bool(mgr.check_optimality_condition(xs, y)))
    # a dual for which every reduced cost is non-negative: y = 0 and c >= 0, flipped signs hurt -> use |.| test
    x_sub = rng.random(mgr.var_info["non_fix"].size)
    out["x_sub"] = x_sub
    out["recover_x"] = mgr.recover_x_from_sub_x(x_sub)
    vb_sub = rng.integers(-2, 1, size=mgr.var_info["non_fix"].size)
    cb_sub = rng.integers(-1, 1, size=mgr.mcf.b.size)
    rb = mgr.recover_basis_from_sub_basis(Basis(vb_sub, cb_sub))
    out["vb_sub"], out["cb_sub"], out["recover_vb"] = vb_sub, cb_sub, rb.vbasis
    return out


# ------------------------------------------------------------------ OT
def ot_case(inst, y_seed=9, with_tree=True):
    ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
    S, D = inst.M.shape
    out = dict(s=inst.s, d=inst.d, M=inst.M, x=inst.x)
    mgr = ref_nm.OTManager(ot)
    queue, ind = mgr.get_sorted_flows(inst.x)
    out["ind"], out["queue_ref"] = ind, queue.astype(np.int64)
    mcf = ot.to_MCF()
    out.update(csr_parts(mcf.A, "Amcf"))
    out["b_mcf"], out["c_mcf"] = mcf.b, mcf.c
    if with_tree:
        mgr.get_mcf()
        basis, push_iter = ref_tree.tree_basis_identify(mgr, ind)
        out["tree_edges"] = np.sort(ref_tree.max_weight_spanning_tree(ot, ind)).astype(np.int64)
        out["tree_vb"], out["tree_cb"] = basis.vbasis, basis.cbasis
        out["push_iter"] = np.array(push_iter)
    # cnet_ot set-up
    mgr2 = ref_nm.OTManager(OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy()))
    bigM = mgr2.m * np.max(inst.M)
    mgr2.extend_by_bigM(bigM)
    mgr2.get_mcf()
    mgr2.set_initial_basis()
    out["bigM"] = np.array(bigM)
    out["s1"], out["d1"], out["M1"] = mgr2.ot.s, mgr2.ot.d, mgr2.ot.M
    out["mask1"] = mgr2.mask_sub_ot.copy()
    out["artificial"] = mgr2.artificial_vars.astype(np.int64)
    out["vb0"], out["cb0"] = mgr2.basis.vbasis, mgr2.basis.cbasis
    batch = int(1.2 * mgr2.m)
    mgr2.add_free_variables(queue[:batch])
    out["batch0"] = np.array([0, batch])
    out["mask2"] = mgr2.mask_sub_ot.copy()
    sub = mgr2.get_sub_problem()
    out.update(csr_parts(sub.A, "Asub"))
    out["csub"], out["usub"] = sub.c, sub.u
    rng = np.random.default_rng(y_seed)
    y = rng.standard_normal(mgr2.mcf.b.size)
    out["y"] = y
    out["rc"] = mgr2.get_reduced_cost_for_original_OT(y)
    xs = np.zeros(mgr2.ot.s.size * mgr2.ot.d.size)
    out["opt_flag"] = np.array(bool(mgr2.check_optimality_condition(xs, y)))
    # a dual that is optimal-looking: y_i = 0 -> rc = M >= 0
    out["opt_flag_zero_y"] = np.array(bool(mgr2.check_optimality_condition(xs, np.zeros_like(y))))
    return out


# ------------------------------------------------------------------ column generation schedule
class _FakeManager:
    def __init__(self, m, n, rounds):
        self.m, self.n, self.rounds = m, n, rounds
        self.slices = []
        self.basis = None
        self._left = 0

    def add_free_variables(self, idx):
        self.slices.append((self._left, self._left + len(idx)))
        self._left += len(idx)

    def update_subproblem(self):
        pass

    def solve_subproblem(self, solver, settings):
        return Output(x=np.zeros(1), y=np.zeros(1), obj_val=1.0, runtime=datetime.timedelta(0), iter_count=3,
                      basis=Basis(np.zeros(1), np.zeros(1)), status="OPTIMAL")

    def recover_obj_val(self, v):
        return v

    def set_basis(self, b):
        self.basis = b

    def recover_basis_from_sub_basis(self, b):
        return b

    def recover_x_from_sub_x(self, x):
        return x

    def check_optimality_condition(self, x, y):
        return len(self.slices) >= self.rounds


def cg_cases():
    cases = []
    for (m, n, qlen, rounds) in [(10, 100, 100, 3), (10, 100, 100, 99), (5, 6000, 6000, 4), (65, 512, 512, 5),
                                 (27, 180, 180, 2), (3, 4000, 1000, 10)]:
        fm = _FakeManager(m, n, rounds)
        out = quiet(ref_nalg.column_generation, fm, np.arange(qlen), "GRB", None)
        cases.append(dict(m=m, n=n, qlen=qlen, rounds=rounds, slices=fm.slices, iter_count=int(out.iter_count)))
    return cases


# ------------------------------------------------------------------ control flow of run_perturb_algorithm (G6)
def canned_backend(inst, script, barrier_obj, log):
    """Stand-in for ``solve_lp`` at the module seam of lp_methods/algorithms.py (the reference calls the
    name ``solve_lp`` it imported, :38,50,69).  It performs no optimisation: every answer is a fixed function
    of its arguments, so that the reference and the build can be driven through the same branches.
      call 0 (the initial barrier solve)  -> (inst.x, inst.y, barrier_obj), OPTIMAL
      re-solve k (method 'barrier')       -> status script[k]; x, y = the warm start it was handed;
                                             vbasis = 0 where x > 1e-3 else -1; cbasis = -1
      final solve ('primal_simplex')      -> x, y = the warm start it was handed, OPTIMAL
    tests/test_gpu_control_flow.py implements the same function for the build."""
    state = {"resolves": 0}

    def solve_lp(lp, solver="GRB", method="default", settings=None, warm_start_basis=None, warm_start_solution=None):
        rec = dict(method=method, n=int(lp.c.size), m=int(lp.b.size), solver=solver,
                   presolve=settings.presolve, crossover=settings.crossover, barrierTol=settings.barrierTol,
                   optimalityTol=settings.optimalityTol, log_file=settings.log_file,
                   has_ws_solution=warm_start_solution is not None, has_ws_basis=warm_start_basis is not None)
        call = len(log)
        if call == 0:
            out = Output(x=inst.x.copy(), y=inst.y.copy(), obj_val=barrier_obj, status="OPTIMAL",
                         runtime=datetime.timedelta(0), iter_count=0, bar_iter_count=7)
        elif method == "barrier":
            status = script[state["resolves"]]
            state["resolves"] += 1
            xs, ys = warm_start_solution
            rec["c_sub"] = np.asarray(lp.c).tolist()
            rec["n_eq_rows"] = int(np.count_nonzero(np.asarray(lp.sense) == "="))
            out = Output(x=np.asarray(xs).copy(), y=np.asarray(ys).copy(), obj_val=float(lp.c @ xs), status=status,
                         runtime=datetime.timedelta(0), iter_count=0,
                         basis=Basis(np.where(np.asarray(xs) > 1e-3, 0, -1), np.full(lp.b.size, -1)))
        else:
            xs, ys = warm_start_solution
            rec["ws_x"] = np.asarray(xs).tolist()
            rec["ws_vbasis"] = np.asarray(warm_start_basis.vbasis).astype(int).tolist()
            rec["ws_cbasis"] = np.asarray(warm_start_basis.cbasis).astype(int).tolist()
            out = Output(x=np.asarray(xs).copy(), y=np.asarray(ys).copy(), obj_val=float(lp.c @ xs), status="OPTIMAL",
                         runtime=datetime.timedelta(0), iter_count=11,
                         basis=Basis(np.asarray(warm_start_basis.vbasis), np.asarray(warm_start_basis.cbasis)))
        rec["returned_status"] = out.status
        log.append((rec, out))
        return out

    return solve_lp


def _exact_nullspace_projection(A, v, A_f=None):
    """Harness stand-in for the reference's Gurobi QP (apply_projector_qp, algorithms.py:240-265) on the
    small golden LPs: the exact orthogonal projection by dense least squares."""
    Ad = np.asarray(sp.csr_matrix(A).todense())
    if A_f is not None:
        raise NotImplementedError("golden LPs have no free variables")
    z, *_ = np.linalg.lstsq(Ad @ Ad.T, Ad @ v, rcond=None)
    return v - Ad.T @ z


def control_flow_case(name, inst, script, early):
    lp = GeneralLP(inst.A.copy(), inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    # gamma in force at the re-solve that succeeds
    g, gd = 1e-3, 1e-3
    for status in script:
        if status in ("INFEASIBLE", "UNBOUNDED"):
            g, gd = g * 1e-5, gd * 1e-5 ** 2
    mgr = quiet(ref_alg.get_perturb_problem, lp, inst.x, inst.y, g, gd, False)
    exact = float(lp.c @ mgr.get_orix(mgr.get_subx(inst.x)))
    barrier_obj = exact if early else exact * 1.5 + 1.0
    log, gammas = [], []
    real_solve, real_qp, real_gpp = ref_alg.solve_lp, ref_alg.apply_projector_qp, ref_alg.get_perturb_problem

    def rec_gpp(lp_, x, y, gamma, gamma_dual, is_feas):
        m_ = real_gpp(lp_, x, y, gamma, gamma_dual, is_feas)
        gammas.append(dict(gamma=gamma, gamma_dual=gamma_dual, is_feas=bool(is_feas),
                           n_fix_low=int(m_.var_info["fix_low"].size), n_fix_up=int(m_.var_info["fix_up"].size),
                           n_fixed_rows=int(m_.fixed_constraints.size)))
        return m_

    ref_alg.solve_lp = canned_backend(inst, script, barrier_obj, log)
    ref_alg.apply_projector_qp = _exact_nullspace_projection
    ref_alg.get_perturb_problem = rec_gpp
    buf = io.StringIO()
    try:
        with redirect_stdout(buf):
            result = ref_alg.run_perturb_algorithm(lp, solver="CANNED", barrierTol=1e-7, optimalityTol=1e-5, log_file="")
    finally:
        ref_alg.solve_lp, ref_alg.apply_projector_qp, ref_alg.get_perturb_problem = real_solve, real_qp, real_gpp
    returned_by = [i for i, (_, o) in enumerate(log) if o is result]
    return dict(name=name, script=list(script), early=bool(early), barrier_obj=barrier_obj,
                calls=[r for r, _ in log], gammas=gammas, printed=buf.getvalue().splitlines(),
                returned_by_call=returned_by[0], result_status=result.status, result_x_len=int(np.asarray(result.x).size))


def control_flow_cases():
    c1 = workloads.config1()
    g2 = workloads.sparse_lp(300, 1500, 6, seed=12, stratified=False, frac_upper=0.3)
    return [control_flow_case("afiro_early_return", c1, ["OPTIMAL"], True),
            control_flow_case("afiro_gap_then_simplex", c1, ["OPTIMAL"], False),
            control_flow_case("afiro_infeasible_once_then_simplex", c1, ["INFEASIBLE", "OPTIMAL"], False),
            control_flow_case("small_unbounded_twice_early_return", g2, ["UNBOUNDED", "UNBOUNDED", "OPTIMAL"], True),
            control_flow_case("small_infeasible_unbounded_then_simplex", g2, ["INFEASIBLE", "UNBOUNDED", "OPTIMAL"], False)]


# ------------------------------------------------------------------ digests on larger, regenerated inputs
def digest_cases():
    d = {}
    inst = workloads.sparse_lp(2000, 10000, 20, seed=22, stratified=True)
    lp = GeneralLP(inst.A.copy(), inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    mgr = quiet(ref_alg.get_perturb_problem, lp, inst.x, inst.y, 1e-3, 1e-3, True)
    d["lp_2000x10000"] = dict(
        input=sha(inst.A.data, inst.A.indices, inst.A.indptr, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y),
        s_d=sha(lp.get_dual_slack(inst.y)), s_p=sha(lp.get_primal_slack(inst.x)),
        fix_low=sha(mgr.var_info["fix_low"].astype(np.int64)), fix_up=sha(mgr.var_info["fix_up"].astype(np.int64)),
        fixed_rows=sha(mgr.fixed_constraints.astype(np.int64)),
        counts=[int(mgr.var_info["fix_low"].size), int(mgr.var_info["fix_up"].size), int(mgr.fixed_constraints.size)],
        b_sub=sha(mgr.lp_sub.b), A_sub=sha(mgr.lp_sub.A.data, mgr.lp_sub.A.indices.astype(np.int32), mgr.lp_sub.A.indptr.astype(np.int64)))
    mi = workloads.mcf(4096, 32768, seed=33)
    mm = ref_nm.MCFManagerStd(MinCostFlow(A=mi.A.copy(), b=mi.b.copy(), c=mi.c.copy(), u=mi.u.copy()))
    _, ind = mm.get_sorted_flows(mi.x)
    d["mcf_4096x32768"] = dict(input=sha(mi.A.data, mi.A.indices, mi.A.indptr, mi.x, mi.u), ind=sha(ind),
                               ind_sum=float(ind.sum()))
    rng = np.random.default_rng(44)
    S = D = 784
    s = rng.random(S) + 0.05
    dd = rng.random(D) + 0.05
    x = (s[:, None] * dd[None, :] * rng.uniform(0.5, 1.5, (S, D))).ravel()
    om = ref_nm.OTManager(OptTransport(s, dd * (s.sum() / dd.sum()), workloads.grid_cost(28)))
    _, ind = om.get_sorted_flows(x)
    d["ot_784x784"] = dict(input=sha(s, dd, x), ind=sha(ind), note="d rescaled by s.sum()/d.sum() before the call")
    return d


def main():
    g1 = lp_case(workloads.config1())
    np.savez_compressed(os.path.join(HERE, "g1_lp_afiro.npz"), **g1)
    g2 = lp_case(workloads.sparse_lp(300, 1500, 6, seed=12, stratified=False, frac_upper=0.3))
    np.savez_compressed(os.path.join(HERE, "g2_lp_small.npz"), **g2)
    g3 = mcf_case(workloads.mcf(64, 512, seed=3))
    np.savez_compressed(os.path.join(HERE, "g3_mcf_small.npz"), **g3)
    g4 = ot_case(workloads.ot(12, 15, seed=7))
    np.savez_compressed(os.path.join(HERE, "g4_ot_small.npz"), **g4)
    with open(os.path.join(HERE, "g5_cg_schedule.json"), "w") as f:
        json.dump(cg_cases(), f, indent=1)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digest_cases(), f, indent=1)
    with open(os.path.join(HERE, "g6_control_flow.json"), "w") as f:
        json.dump(control_flow_cases(), f, indent=1)
    for name in sorted(os.listdir(HERE)):
        print(f"{name:28s} {os.path.getsize(os.path.join(HERE, name)):9d} B")
    print("g1: fixed", g1["fix"].size, "rows", g1["fixed_rows"].size, "cg iters", int(g1["cg_iters"]), "info", int(g1["cg_info"]))
    print("g2: fixed", g2["fix"].size, "rows", g2["fixed_rows"].size, "cg iters", int(g2["cg_iters"]), "info", int(g2["cg_info"]))
    print("g4: push_iter", int(g4["push_iter"]))


if __name__ == "__main__":
    main()
