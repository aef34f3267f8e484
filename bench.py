#!/usr/bin/env python3
"""Benchmark of the crossover scoring pass on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one primal-dual indicator scoring pass of ``get_perturb_problem``
(reference lp_methods/algorithms.py:99-106) plus the pricing reduction, over the
LP resident in HBM:
    K1  sx_score_columns   s_d = c - A^T y, column codes        (this rank's column block)
    K2  sx_score_rows      s_p = b - A x,  row flags            (this rank's row block)
        sx_select_indices  x3: fix_low, fix_up, fixed_rows      (np.where of the reference)
    K10 sx_price           min reduced cost + violation count   (this rank's column block)
    N>1: one RCCL all-gather of the 24-byte pricing records + one all-reduce(SUM) of the
         three set sizes (the only exchange the column-sharded path needs).
Workload (weak scaling): rank r owns column block r (m x n_block, CSC) and row block r
(m/N x N*n_block, CSR) of one global LP; default c5 = BASELINE.json configs[4] per GPU
(m=1e6, n_block=1e7, 8 nnz/col, 1.46 GB of algorithmic bytes for K1 alone -- far beyond the
256 MiB Infinity Cache, so HBM GB/s is honest).  ``--workload c2`` runs configs[1]
(2e4 x 1e5, cache resident).

Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` for the walk that takes
the largest share of the step (K1, K2 and K10 are all timed with HIP events inside the timed region and
reported under ``kernels``), ``roofline_uniform`` (the same three walks on the no-locality variant of the
workload, N = 1 only), ``cpu_baseline`` (the numpy/scipy oracle timed on the host cores, rank 0, N = 1 only)
and ``crossover`` (wall times of whole crossovers through the drop-in API).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X nominal HBM3E bandwidth (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # name: (m, n_block, nnz/col, description)
    "c5": (1_000_000, 10_000_000, 8, "synthetic netlib-style LP, 1e6 rows x 1e7 cols per GPU, 8 nnz/col, CSC+CSR"),
    "c2": (20_000, 100_000, 20, "synthetic random sparse LP, 2e4 rows x 1e5 cols, 20 nnz/col (cache resident)"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def crossover_host_path(cpu_budget_s: float):
    """Crossover wall-time of the part this repository accelerates: one ``get_perturb_problem`` call
    (reference lp_methods/algorithms.py:79-111 -- scoring, index sets, projector CG, perturbed cost,
    sub-problem) on BASELINE config 2 (2e4 x 1e5, 2e6 nnz) from an interior point in host memory to
    the restricted LP in host memory, through the drop-in Python API (uploads and downloads included).
    The LP re-solves that follow are third-party solver time on both sides and are not part of it.
    CPU side: the numpy/scipy oracle with the matrix-free CG -- faster than the reference's own
    explicit Y*Y^T path (55.8 s measured in SURVEY.md section 6), so the ratio is conservative."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods.algorithms import get_perturb_problem
    inst = workloads.config2()
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    times = []
    mgr = None
    for _ in range(5):                      # first call uploads the matrix; report the steady state too
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            mgr = get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        times.append((time.perf_counter() - t0) * 1e3)
    from oracle import lp_path as L         # checker / baseline only
    t0 = time.perf_counter()
    res = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    c_pt, info = L.perturbed_cost_full(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, inst.x, False, explicit=False)
    sub = L.sub_problem(inst.A, inst.b, c_pt, inst.l, inst.u, inst.sense, res["fix_low"], res["fix_up"], res["fixed_rows"])
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = (np.array_equal(mgr.var_info["fix_low"], res["fix_low"]) and np.array_equal(mgr.var_info["fix_up"], res["fix_up"])
            and np.array_equal(mgr.fixed_constraints, res["fixed_rows"])
            and np.array_equal(mgr.lp_sub.A.indices, sub["A"].indices) and np.array_equal(mgr.lp_sub.b, sub["b"]))
    if not same:
        raise SystemExit("bench: device get_perturb_problem disagrees with the CPU oracle")
    return {"workload": "c2: 2e4 x 1e5, 2e6 nnz, get_perturb_problem (is_feas=False), host memory to host memory",
            "gpu_ms_first_call": times[0], "gpu_ms": float(np.median(times[1:])), "cpu_ms": cpu_ms, "cpu_cores": 1,
            "cpu_kind": "port (matrix-free CG; the reference's explicit YY^T path took 55.8 s in SURVEY.md)",
            "speedup": cpu_ms / float(np.median(times[1:])), "cg_iters": int(mgr.perturb_info["cg_iters"]),
            "fixed_columns": int(mgr.get_num_fixed_variables()), "fixed_rows": int(mgr.get_num_fixed_constraints()),
            "index_sets_and_subproblem_match_cpu": True}


def crossover_lp_end_to_end(highs_limit_s: float, cpu_path_s: float = 0.0):
    """BASELINE metric 'crossover wall-time (ms)', LP case, config 2 (2e4 x 1e5): from the interior point (x, y)
    in host memory to the optimal vertex of the perturbed sub-problem and its basis in host memory, i.e.
    get_perturb_problem + the re-solve (reference lp_methods/algorithms.py:45-61) + the gap test (:63), all on the
    GPU (solver 'HIP': crossover from the interior point by the device simplex).  Beside it the CPU path: the same
    host arithmetic by the numpy/scipy oracle plus the re-solve of the same sub-problem by HiGHS (the stand-in
    for Gurobi), stopped at ``highs_limit_s`` seconds so that the default run stays short -- a limit that is hit
    is reported as such, not as a solve time."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.config2()
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        t1 = time.perf_counter()
        out = solve_lp(mgr.lp_sub, "HIP", "barrier", SolverSettings(presolve="on", log_console=0),
                       warm_start_solution=(mgr.get_subx(inst.x), inst.y))
        ok = alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x))
    t2 = time.perf_counter()
    if out.status != "OPTIMAL" or not ok:
        raise SystemExit("bench: the device crossover of config 2 did not reach an optimal vertex")
    rec = {"workload": "c2: 2e4 x 1e5, 2e6 nnz; interior point -> optimal vertex + basis of the perturbed 2e4 x 2e4 "
                       "sub-problem, host memory to host memory",
           "gpu_ms": (t2 - t0) * 1e3, "gpu_get_perturb_problem_ms": (t1 - t0) * 1e3, "gpu_resolve_ms": (t2 - t1) * 1e3,
           "simplex_pivots": int(out.iter_count), "gap_test_passed": True, "objective": float(mgr.lp_sub.c @ out.x)}
    path = _cpu_path_record("lp_c2_cpu_path.json", {"which": "c2"}, cpu_path_s, rec["objective"], rec["gpu_ms"], rec["gpu_resolve_ms"])
    if path is not None:
        rec["cpu_path"] = path          # (its ratio keys say whether both sides ran on this host)
    if highs_limit_s <= 0:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "lp_c2_highs.json")), reverse=True):
            cpu = json.load(open(path))
            cpu["recorded_in"] = os.path.relpath(path, ROOT) + " (python bench.py --highs-seconds SECONDS re-measures it)"
            rec["cpu"] = cpu
            key = "ratio_vs_recorded_cpu_other_host_resolve"
            if cpu.get("cpu_resolve_status") != "OPTIMAL":
                key += "_lower_bound_no_cpu_solution"
            rec[key] = cpu["cpu_resolve_s"] * 1e3 / rec["gpu_resolve_ms"]
            break
    if highs_limit_s > 0:
        t0 = time.perf_counter()
        status = "TIME_LIMIT"
        try:
            with redirect_stdout(io.StringIO()):
                ref = solve_lp(mgr.lp_sub, "HGS", "default",
                               SolverSettings(presolve="on", log_console=0, timeLimit=int(highs_limit_s)))
            status = ref.status
        except Exception as exc:                # the stand-in raises when it stops without a solution
            status = f"{type(exc).__name__}"
        secs = time.perf_counter() - t0
        solved = status == "OPTIMAL"
        rec.update({"cpu_kind": "HiGHS (scipy's bundled build, all host cores it chooses to use) on the same sub-problem; "
                                "the reference would call Gurobi here",
                    "cpu_resolve_status": status, "cpu_resolve_seconds": secs,
                    "cpu_resolve_time_limit_s": highs_limit_s,
                    "speedup_resolve": (secs * 1e3 / rec["gpu_resolve_ms"]) if solved else None,
                    "speedup_resolve_at_least": None if solved else secs * 1e3 / rec["gpu_resolve_ms"]})
    return rec


def _cpu_crossover_lp(inst, highs_limit_s: float):
    """The CPU path of an LP crossover, timed to completion: the host arithmetic of get_perturb_problem by the
    numpy/scipy oracle (matrix-free CG) + the re-solve of the perturbed sub-problem by HiGHS (scipy's build; the
    stand-in for the reference's Gurobi call, lp_methods/algorithms.py:50-54)."""
    from scipy.optimize import linprog
    from oracle import lp_path as L         # checker / baseline only
    t0 = time.perf_counter()
    res = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    c_pt, _ = L.perturbed_cost_full(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, inst.x, False, explicit=False)
    sub = L.sub_problem(inst.A, inst.b, c_pt, inst.l, inst.u, inst.sense, res["fix_low"], res["fix_up"], res["fixed_rows"])
    t1 = time.perf_counter()
    lt = np.asarray(sub["sense"]) == "<"
    A = sub["A"].tocsr()
    ref = linprog(sub["c"], A_ub=A[lt], b_ub=sub["b"][lt], A_eq=A[~lt], b_eq=sub["b"][~lt], bounds=np.c_[sub["l"], sub["u"]],
                  method="highs", options={"time_limit": float(highs_limit_s)})
    t2 = time.perf_counter()
    return {"cpu_kind": "port (numpy/scipy oracle, matrix-free CG) + HiGHS (scipy's bundled build, method 'highs') on the same "
                        "sub-problem; the reference would call Gurobi there",
            "cpu_host_arithmetic_s": t1 - t0, "cpu_resolve_s": t2 - t1, "cpu_total_s": t2 - t0,
            "cpu_resolve_status": "OPTIMAL" if ref.status == 0 else ("TIME_LIMIT" if ref.status == 1 else f"status {ref.status}"),
            "cpu_resolve_iterations": int(getattr(ref, "nit", 0) or 0), "cpu_objective": float(ref.fun) if ref.status == 0 else None,
            "cpu_cores": os.cpu_count(), "cpu_time_limit_s": float(highs_limit_s)}


def _cpu_path_record(name: str, which: dict, limit_s: float, objective: float, gpu_ms: float, gpu_resolve_ms: float):
    """The CPU path that finishes, beside an LP crossover (tools/cpu_lp_path.py: numpy/scipy oracle + oracle/pdlp.py
    first-order stage + HiGHS' simplex warm-started from the basis that point indicates -- the device's own route on
    the host's cores; HiGHS alone does not finish on these sub-problems).  Timed in this run with ``limit_s`` > 0
    (written to gpurun_out/<name>), otherwise the latest record committed under profiles/.  The perturbed LP has ONE
    optimal vertex: the two objectives must agree."""
    rec = None
    if limit_s > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_lp_path                      # (imports oracle/: checker / baseline only)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        # the CPU path gets the host's whole CPU quota for its BLAS / OpenMP pools (the device path cuts them to a quarter
        # for the sake of its launching thread: smart_crossover/hip/host_threads.py)
        from smart_crossover.hip import host_threads
        quota = host_threads.cpu_quota()
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=quota):
                rec = cpu_lp_path.run(dict(which, limit=limit_s, out=os.path.join(ROOT, "gpurun_out", name)))
        except ImportError:
            rec = cpu_lp_path.run(dict(which, limit=limit_s, out=os.path.join(ROOT, "gpurun_out", name)))
        rec["measured"] = "in this run, on this host"
        rec["host_cpu_quota"] = quota
    else:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
            rec = json.load(open(path))
            rec["recorded_in"] = os.path.relpath(path, ROOT) + " (bench.py --lp-cpu-path SECONDS re-measures it)"
            break
    if rec is None:
        return None
    # a ratio is a SPEEDUP only when both sides ran on this host in this run; against a committed record (measured in
    # the build container, other cores) it is named for what it is
    same_host = "measured" in rec and rec["measured"].startswith("in this run")
    key = "speedup" if same_host else "ratio_vs_recorded_cpu_other_host"
    if rec.get("cpu_resolve_status") == "OPTIMAL":
        # (reported, not fatal: a committed record must not be able to take the whole line down)
        rec["objective_matches_device"] = bool(abs(rec["cpu_objective"] - objective) <= 1e-7 * (1 + abs(objective)))
        rec["device_objective"] = objective
        rec[f"{key}_total"] = rec["cpu_total_s"] * 1e3 / gpu_ms
        rec[f"{key}_resolve"] = rec["cpu_resolve_s"] * 1e3 / gpu_resolve_ms
    elif rec.get("cpu_total_s"):   # stopped without a solution: a lower bound, named as such
        rec[f"{key}_total_lower_bound_no_cpu_solution"] = rec["cpu_total_s"] * 1e3 / gpu_ms
    return rec


def _device_lp_crossover(inst, reps: int, what: str):
    """One LP crossover through the drop-in API, host memory to host memory: get_perturb_problem (K1-K6) + the re-solve
    of the perturbed sub-problem (reference lp_methods/algorithms.py:45-61: first-order stage K16p + sparse crossover
    K16s on the bordered band factorisation) + the reference's gap test (:63).  Every call builds a fresh GeneralLP and
    uploads its matrix; the LAST of ``reps`` calls is reported, all are listed (the first also pays the first-use set-up
    of kernels and layouts)."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller import solving
    runs = []
    for rep in range(reps):
        lp = GeneralLP(inst.A, inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
        if rep == reps - 1 and reps >= 3 and "back_to_back" not in os.environ.get("SX_BENCH_EXPERIMENT", ""):
            # the LAST call is timed after a pause, the call before it (index reps - 2) right behind its predecessor: the
            # back-to-back figure.  They differed by ~90 ms (round 3's "in-bench slowdown") until the host's BLAS / OpenMP
            # pools were cut to the CPU quota (smart_crossover/hip/host_threads.py; profiles/r04/in_bench_slowdown.md);
            # both stay in the line.
            time.sleep(0.25)
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
            t1 = time.perf_counter()
            caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
            caller.read_genlp(mgr.lp_sub)
            caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
            caller.run_barrier()
            out = caller.return_output()
            ok = alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x))
        t2 = time.perf_counter()
        if out.status != "OPTIMAL" or not ok:
            raise SystemExit(f"bench: the device crossover of {what} did not reach an optimal vertex")
        runs.append((t2 - t0, t1 - t0, t2 - t1, caller, out, mgr))
        if rep < reps - 1:
            del lp, mgr, caller, out
            runs[-1] = runs[-1][:3] + (None, None, None)
    tot, tgp, trs, caller, out, mgr = runs[-1]
    p = caller.pdlp
    m, n = inst.A.shape
    return {"workload": f"{what}: {m} rows x {n} columns, {inst.A.nnz} entries (staircase + 1 % linking rows); interior point -> "
                        "optimal vertex + basis of the perturbed sub-problem, host memory to host memory",
            "sub_problem_shape": list(mgr.lp_sub.A.shape), "gpu_ms": tot * 1e3, "gpu_ms_first_call": runs[0][0] * 1e3,
            "gpu_ms_calls": [r[0] * 1e3 for r in runs],
            "gpu_ms_back_to_back": runs[-2][0] * 1e3 if len(runs) >= 3 else None,
            "timing_note": "gpu_ms = the last call, started 0.25 s after the call before it returned; "
                           "gpu_ms_back_to_back = the call before, started right behind its predecessor; gpu_ms_first_call also pays "
                           "the first-use set-up of kernels and layouts" if len(runs) >= 3 else "one call",
            "gpu_get_perturb_problem_ms": tgp * 1e3, "gpu_resolve_ms": trs * 1e3,
            "first_order_stage": {"iterations": int(p.iters), "restarts": int(p.restarts), "seconds": caller.pdlp_seconds,
                                  "us_per_iteration": caller.pdlp_seconds / max(int(p.iters), 1) * 1e6,
                                  "primal_residual": p.primal_residual, "dual_residual": p.dual_residual, "gap": p.gap},
            "crossover": {"kind": caller.solved_by, "simplex_iterations": int(out.iter_count)},
            "objective": float(mgr.lp_sub.c @ out.x), "gap_test_passed": True}


def crossover_lp_1e6(lp_highs_s: float, cpu_path_s: float = 0.0):
    """BASELINE metric 'crossover wall-time (ms)' on the configuration it is quoted on: a 1e6-variable netlib-style LP
    (workloads.netlib_lp: 1e5 rows, 8e6 entries, staircase + linking rows).  Warm process: the third of three calls is
    reported.  CPU paths beside it are committed records (no CPU run finishes at this size, profiles/r03/); their ratios
    are named ``ratio_vs_recorded_cpu_other_host*`` -- the speedup both sides of which ran on one host is the
    ``lp_2e4_rows`` leg's."""
    rec = _device_lp_crossover(workloads.netlib_lp(), 3, "netlib_lp (the 1e6-variable LP of the metric)")
    cpu = None
    if lp_highs_s > 0:
        cpu = _cpu_crossover_lp(workloads.netlib_lp(), lp_highs_s)
        cpu["measured"] = "in this run, on this host"
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(cpu, open(os.path.join(ROOT, "gpurun_out", "lp_1e6_highs.json"), "w"), indent=1)
    else:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "lp_1e6_highs.json")), reverse=True):
            cpu = json.load(open(path))
            cpu["recorded_in"] = os.path.relpath(path, ROOT) + " (python bench.py --lp-highs SECONDS re-measures it)"
            break
    if cpu is not None:
        rec["cpu"] = cpu
        same_host = str(cpu.get("measured", "")).startswith("in this run")
        key = "speedup_total" if same_host else "ratio_vs_recorded_cpu_other_host_total"
        if cpu.get("cpu_resolve_status") == "OPTIMAL":
            if abs(cpu["cpu_objective"] - rec["objective"]) > 1e-7 * (1 + abs(rec["objective"])):
                raise SystemExit("bench: the device's optimum of the 1e6-variable LP differs from HiGHS'")
            rec[key] = cpu["cpu_total_s"] * 1e3 / rec["gpu_ms"]
        else:
            rec[key + "_lower_bound_no_cpu_solution"] = cpu["cpu_total_s"] * 1e3 / rec["gpu_ms"]
        rec["target"] = ">= 5x lower crossover wall-time than the CPU path (BASELINE.json); measured on one host at the largest " \
                        "size where a CPU path finishes: crossover.lp_2e4_rows"
    path = _cpu_path_record("lp_1e6_cpu_path.json", {}, cpu_path_s, rec["objective"], rec["gpu_ms"], rec["gpu_resolve_ms"])
    if path is not None:
        rec["cpu_path"] = path
    return rec


def crossover_lp_2e4(cpu_limit_s: float):
    """The same crossover at a fifth of the headline size -- netlib_lp(20000, 200000), the largest of the family on which
    a CPU path FINISHES (HiGHS alone does not: profiles/r03/lp_1e6_highs.json) -- with BOTH sides timed in this run on
    this host: the device through the drop-in API, the CPU by tools/cpu_lp_path.py (numpy/scipy oracle + oracle/pdlp.py +
    HiGHS' simplex warm-started from the indicated basis, all host cores HiGHS chooses to use; the reference would call
    Gurobi).  ``speedup_total`` here is the one measured GPU-over-CPU crossover ratio of the line."""
    inst = workloads.netlib_lp(20_000, 200_000)
    rec = _device_lp_crossover(inst, 2, "netlib_lp(20000, 200000)")
    if cpu_limit_s > 0:
        path = _cpu_path_record("lp_2e4_rows_cpu_path.json", {"m": 20_000, "n": 200_000}, cpu_limit_s, rec["objective"],
                                rec["gpu_ms"], rec["gpu_resolve_ms"])
        rec["cpu_path"] = path
        for k in ("speedup_total", "speedup_resolve", "speedup_total_lower_bound_no_cpu_solution"):
            if path is not None and k in path:
                rec[k] = path[k]
        rec["target"] = ">= 5x lower crossover wall-time than the CPU path (BASELINE.json)"
    return rec


def crossover_lp_c5_end_to_end():
    """The LP crossover at config-5 size (BASELINE configs[4]: 1e6 rows x 1e7 columns, 8e7 entries, netlib_lp of that size):
    get_perturb_problem + first-order stage + sparse crossover on the bordered band factorisation + gap test, host memory
    to host memory.  Two calls, the second reported (`gpu_ms`; the first, which also makes the ~10 GB of driver allocations
    the library's memory pool keeps afterwards, is listed as `gpu_ms_first_call`) -- the headline leg reports its third."""
    inst = workloads.netlib_lp(1_000_000, 10_000_000)
    rec = _device_lp_crossover(inst, 2, "netlib_lp(1e6, 1e7) = config-5 size")
    rec["scoring_on_this_lp"] = _scoring_walks_on(inst)
    return rec


def _scoring_walks_on(inst, reps: int = 5):
    """The three scoring walks (K1 score_columns, K2 score_rows, K10 price) on THIS instance's matrix -- the same LP the
    crossover leg above re-solves -- so that both halves of BASELINE's metric are quoted on one config-5 workload (the
    headline scoring step runs on workloads.lp_shard, a kernel workload whose (x, y) is no consistent pair).  HIP events
    on the library's stream, algorithmic bytes as in SURVEY.md 8(d)."""
    from smart_crossover.hip import default_context
    ctx = default_context()
    m, n = inst.A.shape
    dA = ctx.matrix(inst.A)
    d = {k: ctx.to_device(getattr(inst, k)) for k in ("b", "c", "l", "u", "x", "y")}
    vb = ctx.to_device(np.where(inst.x - inst.l < 1e-6, -1, np.where(inst.u - inst.x < 1e-6, -2, 0)).astype(np.int8))
    s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    price = ctx.empty(24, np.uint8)
    t = {"k_score_columns": [], "k_score_rows": [], "k_price": []}
    for i in range(2 + reps):
        ctx.marker(0)
        ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        ctx.marker(1)
        ctx.score_rows(dA, d["x"], d["b"], d["y"], 1e-3, s_p, flag)
        ctx.marker(2)
        ctx.price(dA, d["y"], d["c"], vb, 1e-6, None, price)
        ctx.marker(3)
        ctx.sync()
        if i >= 2:
            for k, name in enumerate(t):
                t[name].append(ctx.marker_elapsed(k, k + 1))
    nnz = int(inst.A.nnz)
    algo = {"k_score_columns": 12 * nnz + 49 * n + 8 * m, "k_score_rows": 12 * nnz + 8 * n + 33 * m, "k_price": 12 * nnz + 17 * n + 8 * m}
    out = {"workload": f"netlib_lp {m} rows x {n} columns, {nnz} entries (1 % linking rows at the head)",
           "row_layout": "column-blocked" if dA.rowblock() is not None else "plain walk (auto rule)"}
    for name, ms in t.items():
        avg = float(np.mean(ms))
        out[name] = {"avg_kernel_ms": avg, "algorithmic_bytes": int(algo[name]), "achieved_GBps": algo[name] / avg / 1e6,
                     "frac_of_hbm_peak": algo[name] / avg / 1e6 / HBM_PEAK_GBS, "columns_per_s": n / avg * 1e3}
    dA.free()
    return out


def crossover_lp_c5():
    """get_perturb_problem at full config-5 size (1e6 x 1e7, 8e7 entries) through the drop-in API: scipy CSR +
    numpy vectors in host memory -> restricted sub-problem in host memory (uploads, 1000 CG iterations = 2000
    sparse products, compaction, downloads)."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods.algorithms import get_perturb_problem
    sh = workloads.lp_shard(0, 1)
    rng = np.random.default_rng(11)
    sense = np.where(rng.random(sh.m) < 0.5, "<", "=")
    lp = GeneralLP(sh.row_block, sh.b, sh.c, sh.l, sh.u, sense)
    times = []
    mgr = None
    for rep in range(2):
        fresh = GeneralLP(lp.A, lp.b.copy(), lp.c.copy(), lp.l.copy(), lp.u.copy(), lp.sense.copy()) if rep else lp
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            mgr = get_perturb_problem(fresh, sh.x, sh.y, 1e-3, 1e-3, False)
        times.append((time.perf_counter() - t0) * 1e3)
    info = getattr(mgr, "perturb_info", {}) or {}
    return {"workload": "c5: 1e6 x 1e7, 8e7 nnz, get_perturb_problem (is_feas=False), host memory to host memory",
            "gpu_ms_first_call": times[0], "gpu_ms": times[1], "cg_iters": int(info.get("cg_iters", 0)),
            "fixed_columns": int(mgr.get_num_fixed_variables()), "fixed_rows": int(mgr.get_num_fixed_constraints()),
            "sub_problem_shape": list(mgr.lp_sub.A.shape),
            "cpu_note": "the reference cannot run this size (explicit Y Y^T); its matrix-free restatement needs ~0.4 s per "
                        "CG iteration on one core"}


# the sources K1 / K2 / K10 are compiled from: their HBM traffic depends on nothing else
WALK_SOURCES = ("sx_segwalk.h", "sx_runwalk.h", "sx_slabs.h", "sx_slabs.hip", "sx_window.h", "sx_window.hip", "sx_lp_kernels.hip", "sx_rowblock.h", "sx_rowblock.hip",
                "sx_rowblock_build.hip", "sx_tiles.hip", "sx_sort.hip", "sx_internal.h")


def source_hash() -> str:
    """sha256 over the sources of the walk kernels: a PMC traffic figure is only reused for the build it was
    measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in WALK_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "smart-crossover_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def crossover_network():
    """BASELINE metric, part 'crossover wall-time (ms)', network case: the whole ``network_crossover`` call
    (TNET) on config 3 -- OT on the 28 x 28 grid, 784 x 784, 614,656 arcs -- from the inexact plan in host
    memory to the optimal basis in host memory.  Once with every re-solve on the device (solver 'HIP': device
    simplex with a session), once with the re-solves in HiGHS on the host cores (the stand-in for the
    reference's Gurobi/CPLEX).  Warm process (second call); both reach the same optimal cost."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.config3()
    S, D = inst.M.shape
    out = {"workload": f"c3: optimal transport {S} x {D} (n = {S * D}), Manhattan grid cost, method tnet, host memory to host memory"}
    costs = {}
    for solver, key in (("HIP", "gpu_resident_ms"), ("HGS", "host_solver_ms")):
        times = []
        for _ in range(3):
            ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
            t0 = time.perf_counter()
            with redirect_stdout(io.StringIO()):
                res = network_crossover(inst.x, ot=ot, method="tnet", solver=solver, solver_settings=SolverSettings(log_console=0))
            times.append((time.perf_counter() - t0) * 1e3)
        X = res.x.reshape(S, D)
        if max(np.abs(X.sum(axis=1) - inst.s).max(), np.abs(X.sum(axis=0) - inst.d).max()) > 1e-9:
            raise SystemExit(f"bench: network crossover ({solver}) returned an infeasible plan")
        costs[solver] = float((X * inst.M).sum())
        out[key] = float(np.median(times[1:]))
        out[key.replace("_ms", "_first_call_ms")] = times[0]
        out[f"simplex_iterations_{solver}"] = int(res.iter_count)
    if abs(costs["HIP"] - costs["HGS"]) > 1e-9 * (1 + abs(costs["HGS"])):
        raise SystemExit("bench: device and HiGHS re-solves disagree on the optimal transport cost")
    out["optimal_cost"] = costs["HGS"]
    out["speedup"] = out["host_solver_ms"] / out["gpu_resident_ms"]
    return out


def crossover_mcf(V: int = 4096, E: int = 32768, solvers=("HIP", "HGS"), repeats: int = 2):
    """BASELINE metric, part 'crossover wall-time (ms)', min-cost-flow case: the whole ``network_crossover`` call
    (CNET_MCF) on a V-node / E-arc network of config 4's family (V = 2^17, E = 2^20 is config 4 itself), with the
    re-solves on the device (solver 'HIP': every round is a network LP with a warm tree basis -> dual network
    simplex K16d on the whole GPU) and, when asked for, in HiGHS on the host cores.  Host memory to host memory."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    out = {"workload": f"min-cost flow V = {V}, E = {E} (workloads.mcf, seed 3), method cnet_mcf, host memory to host memory"}
    costs = {}
    for solver, key in (("HIP", "gpu_resident_ms"), ("HGS", "host_solver_ms")):
        if solver not in solvers:
            continue
        times = []
        for _ in range(repeats if solver == "HIP" else 1):
            inst = workloads.mcf(V, E, 3)
            mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
            t0 = time.perf_counter()
            with redirect_stdout(io.StringIO()):
                res = network_crossover(inst.x.copy(), mcf=mcf, method="cnet_mcf", solver=solver)
            times.append((time.perf_counter() - t0) * 1e3)
        costs[solver] = float(inst.c @ res.x[:E])
        flow_violation = float(np.abs(inst.A @ res.x[:E] - inst.b).max())
        if flow_violation > 1e-6 * (1 + float(np.abs(inst.b).max())) or res.x[:E].min() < -1e-7 or (res.x[:E] - inst.u).max() > 1e-7:
            raise SystemExit(f"bench: network crossover ({solver}) returned an infeasible flow")
        out[key] = float(times[-1])
        out[f"simplex_iterations_{solver}"] = int(res.iter_count)
    if len(costs) == 2 and abs(costs["HIP"] - costs["HGS"]) > 1e-9 * (1 + abs(costs["HGS"])):
        raise SystemExit("bench: device and HiGHS re-solves disagree on the optimal flow cost")
    out["optimal_cost"] = costs.get("HGS", costs.get("HIP"))
    if len(costs) == 2:
        out["speedup"] = out["host_solver_ms"] / out["gpu_resident_ms"]
    return out


def _heartbeat():
    """One stderr line a minute while the long legs (HiGHS beside the device crossover) run: a driver that takes
    minutes of silence for a hang sees progress.  stdout stays the single JSON line."""
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(60)
            print(f"[bench] running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    threading.Thread(target=beat, daemon=True).start()


def main():
    _heartbeat()
    if os.environ.get("SX_BENCH_DUMP_AFTER"):      # debugging aid: Python stacks of all threads to stderr after N seconds, repeated
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SX_BENCH_DUMP_AFTER"]), repeat=True, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c5", choices=sorted(WORKLOADS))
    ap.add_argument("--structure", default=None, choices=["staircase", "uniform"],
                    help="row structure of the synthetic LP (default: staircase = netlib-style for c5, uniform for c2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-crossover", action="store_true", help="skip the whole-crossover timings")
    ap.add_argument("--no-uniform", action="store_true", help="skip the no-locality (uniform) record")
    ap.add_argument("--cg-iters", type=int, default=0,
                    help="also time this many iterations of the column-sharded projector CG (one m-vector all-reduce per "
                         "iteration over RCCL at N > 1): reported under 'sharded_cg', not part of the step")
    ap.add_argument("--pricing-rounds", type=int, default=50,
                    help="also time this many pricing rounds of a column-sharded simplex pivot (K10 on the rank's block + "
                         "the all-gather of the 24-byte records at N > 1): reported under 'sharded_pricing' (0: skip)")
    ap.add_argument("--resolve-rows", type=int, default=20000,
                    help="N > 1 only: rows of the LP of the column-sharded re-solve leg ('sharded_resolve'; 0: skip)")
    ap.add_argument("--c4-highs", action="store_true",
                    help="time config 4's network crossover with the re-solves in HiGHS too (94 s on the box's host cores; "
                         "without it the line quotes profiles/r02/netdual_c4_highs.jsonl)")
    ap.add_argument("--highs-seconds", type=float, default=0.0,
                    help="time the HiGHS re-solve of config 2's sub-problem in this run, with this time limit (0: quote the "
                         "record under profiles/)")
    ap.add_argument("--lp-highs", type=float, default=0.0,
                    help="time the CPU path of the 1e6-variable LP crossover (oracle + HiGHS, this time limit in seconds) in "
                         "this run and write gpurun_out/lp_1e6_highs.json (0: quote the record under profiles/)")
    ap.add_argument("--lp-2e4-cpu", type=float, default=600.0,
                    help="HiGHS time limit (s) of the CPU path timed IN THIS RUN beside the device crossover of netlib_lp(20000, "
                         "200000) -- the largest size of the family where a CPU path finishes (~45 s on 8 cores); 0: skip the CPU side")
    ap.add_argument("--no-c5-crossover", action="store_true",
                    help="skip the LP crossover at config-5 size (1e6 x 1e7: ~1.5 min of host-side instance generation)")
    ap.add_argument("--lp-cpu-path", type=float, default=0.0,
                    help="time the CPU path that finishes (tools/cpu_lp_path.py: oracle + first-order stage + warm-started HiGHS "
                         "simplex, this HiGHS time limit in seconds) beside the two LP crossovers in this run and write "
                         "gpurun_out/lp_1e6_cpu_path.json / lp_c2_cpu_path.json (0: quote the records under profiles/)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    from smart_crossover.hip import Context   # raises if libsxhip.so is missing: no CPU fallback

    dist = None
    torch = None
    stream = None
    # SX_BENCH_REHEARSAL=gloo: exercise the N > 1 code path on a box with ONE GPU -- every rank uses
    # device 0 and the two collectives go over gloo through CPU staging.  Numbers from it mean nothing.
    rehearse = world > 1 and os.environ.get("SX_BENCH_REHEARSAL") == "gloo"
    # SX_BENCH_REHEARSAL=rccl1: ONE rank takes the N > 1 code path over RCCL itself (process group "nccl", the
    # all-gather of the 48-byte records, barrier, MAX all-reduce of the time, the sharded CG's all-reduce) -- the
    # rehearsal of the RCCL calls that a box with one GPU allows; launch it under torch.distributed.run as well
    use_dist = world > 1 or os.environ.get("SX_BENCH_REHEARSAL") == "rccl1"
    dev_index = 0 if rehearse else local_rank
    if rehearse:
        os.environ["SX_DEVICE"] = "0"       # (the solver backends' default context as well)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        # kernels and RCCL must share a stream; torch's default stream has handle 0, which the
        # library would read as "create your own", so make an explicit one current
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    elif args.cg_iters > 0:
        # the sharded-CG step hands torch tensors to the kernels: torch has to bring its HIP runtime up BEFORE the
        # library does (the other order leaves torch without a device), and both must share one stream
        import torch
        torch.cuda.set_device(dev_index)
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    ctx = Context(dev_index, stream)
    dev_name, cus, hbm = ctx.device_info()

    m, n_block, k, desc = WORKLOADS[args.workload]
    t0 = time.time()
    structure = args.structure or ("staircase" if args.workload == "c5" else "uniform")
    if m % k or k % world:
        raise SystemExit(f"workload {args.workload}: world={world} must divide nnz/col={k}")
    if structure == "staircase":
        # weak scaling: rows and columns of the global LP grow together (8 row regions and m rows per
        # rank), so every rank's column and row block keeps the shape of the single-GPU problem
        m = m * world
        sh = workloads.lp_shard(rank, world, m=m, n_block=n_block, k=k, seed=5, structure=structure,
                                regions=workloads.STAIR_REGIONS * world)
    else:
        sh = workloads.lp_shard(rank, world, m=m, n_block=n_block, k=k, seed=5, structure=structure)
    desc = f"{desc}; row structure: {structure}"
    if rank == 0:
        log(f"[bench] generated shard in {time.time() - t0:.1f}s: col block {sh.col_block.shape} nnz={sh.col_block.nnz}, "
            f"row block {sh.row_block.shape} nnz={sh.row_block.nnz}; device {dev_name} ({cus} CUs)")

    t0 = time.time()
    dC = ctx.column_shard(sh.col_block)
    dR = ctx.row_shard(sh.row_block)
    n_loc, m_loc, n_tot = sh.n_block, sh.row_block.shape[0], sh.row_block.shape[1]
    d_y = ctx.to_device(sh.y)
    d_x = ctx.to_device(sh.x)
    off = rank * n_loc
    d_xloc = ctx.wrap(d_x.ptr + 8 * off, n_loc, np.float64, owner=d_x)
    roff = rank * m_loc
    d_yloc = ctx.wrap(d_y.ptr + 8 * roff, m_loc, np.float64, owner=d_y)
    d_c, d_l, d_u, d_b = (ctx.to_device(v) for v in (sh.c, sh.l, sh.u, sh.b))
    s_d, code = ctx.empty(n_loc, np.float64), ctx.empty(n_loc, np.uint8)
    s_p, flag = ctx.empty(m_loc, np.float64), ctx.empty(m_loc, np.uint8)
    idx_low, idx_up, idx_row = ctx.empty(n_loc, np.int64), ctx.empty(n_loc, np.int64), ctx.empty(m_loc, np.int64)
    vb = ctx.to_device(np.full(n_loc, -1, dtype=np.int8))
    if use_dist:
        # one 48-byte record per rank and step: pricing record (24 B) + the three set sizes (3 x int64),
        # exchanged by ONE all-gather; every rank reduces the gathered records itself
        t_rec = torch.zeros(48, dtype=torch.uint8, device="cuda")
        t_gather = torch.zeros(48 * world, dtype=torch.uint8, device="cuda")
        price = ctx.wrap(t_rec.data_ptr(), 24, np.uint8, owner=t_rec)
        counts = ctx.wrap(t_rec.data_ptr() + 24, 3, np.int64, owner=t_rec)
    else:
        counts = ctx.empty(3, np.int64)
        price = ctx.empty(24, np.uint8)
    c_low = ctx.wrap(counts.ptr, 1, np.int64, owner=counts)
    c_up = ctx.wrap(counts.ptr + 8, 1, np.int64, owner=counts)
    c_row = ctx.wrap(counts.ptr + 16, 1, np.int64, owner=counts)
    ctx.sync()
    if rank == 0:
        log(f"[bench] upload {time.time() - t0:.1f}s")

    gamma = 1e-3
    # marker ids 5*step + {0: before K1, 1: after K1, 2: after K2, 3: before K10, 4: after K10}

    def step(i, timed):
        if timed:
            ctx.marker(5 * i)
        ctx.score_columns(dC, d_y, d_c, d_xloc, d_l, d_u, gamma, s_d, code)
        if timed:
            ctx.marker(5 * i + 1)
        ctx.score_rows(dR, d_x, d_b, d_yloc, gamma, s_p, flag)
        if timed:
            ctx.marker(5 * i + 2)
        ctx.select_indices(code, 1, idx_low, c_low)
        ctx.select_indices(code, 2, idx_up, c_up)
        ctx.select_indices(flag, 0xFF, idx_row, c_row)
        if timed:
            ctx.marker(5 * i + 3)
        ctx.price(dC, d_y, d_c, vb, 1e-6, None, price)
        if timed:
            ctx.marker(5 * i + 4)
        if use_dist and not rehearse:
            dist.all_gather_into_tensor(t_gather, t_rec)
        elif rehearse:
            cpu_gather = torch.empty(48 * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(cpu_gather, t_rec.cpu())
            t_gather.copy_(cpu_gather)

    def fence():
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            ctx.sync_device()

    for i in range(args.warmup):
        step(i, False)
    fence()
    t_start = time.perf_counter()
    n_marked = min(args.steps, 800)              # marker ids are limited to 4096
    for i in range(args.steps):
        step(i, i < n_marked)
    fence()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- pricing rounds of a column-sharded simplex pivot (ShardedLP.simplex_price): K10 over the rank's column block
    # + ONE all-gather of the 24-byte records; the rest of a pivot (FTRAN, ratio test, update) is replicated
    sharded_pricing = None
    if args.pricing_rounds > 0:
        try:
            def pricing_round():
                ctx.price(dC, d_y, d_c, vb, 1e-6, None, price)
                if use_dist and not rehearse:
                    dist.all_gather_into_tensor(t_gather, t_rec)
                elif rehearse:
                    cpu_gather = torch.empty(48 * world, dtype=torch.uint8)
                    dist.all_gather_into_tensor(cpu_gather, t_rec.cpu())
                    t_gather.copy_(cpu_gather)
            for _ in range(3):
                pricing_round()
            fence()
            t_pr = time.perf_counter()
            for _ in range(args.pricing_rounds):
                pricing_round()
            fence()
            pr_elapsed = time.perf_counter() - t_pr
            if use_dist:
                tt = torch.tensor([pr_elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                pr_elapsed = float(tt.item())
            sharded_pricing = {"rounds": args.pricing_rounds, "us_per_round": pr_elapsed / args.pricing_rounds * 1e6,
                               "pivots_per_s_bound": args.pricing_rounds / pr_elapsed,
                               "columns_priced_per_round": int(world * n_loc),
                               "exchange": f"all_gather of {world} x 24-byte pricing records" if world > 1 else "none (one rank)",
                               "note": "the sharded part of a pivot (smart_crossover.distributed.ShardedLP.simplex_price); FTRAN, "
                                       "ratio test and basis update are replicated"}
        except Exception as exc:          # (the matrices were freed for the no-locality record: skip)
            sharded_pricing = {"skipped": type(exc).__name__}

    k1_list = [ctx.marker_elapsed(5 * i, 5 * i + 1) for i in range(n_marked)]
    k2_list = [ctx.marker_elapsed(5 * i + 1, 5 * i + 2) for i in range(n_marked)]
    k10_list = [ctx.marker_elapsed(5 * i + 3, 5 * i + 4) for i in range(n_marked)]
    nnz_loc = sh.col_block.nnz

    def kernel_records(k1, k2, k10, col_nnz, row_nnz):
        """Algorithmic bytes over the HIP-event time of each of the three walks."""
        algo = {
            # entries 12 B, colptr + c, x, l, u in (40 B), s_d + code out (9 B) per column, y once
            "k_score_columns": 12 * col_nnz + 49 * n_loc + 8 * m,
            # entries 12 B, x once, rowptr + b + y in (24 B) and s_p + flag out (9 B) per row
            "k_score_rows": 12 * row_nnz + 8 * n_tot + 33 * m_loc,
            # as the step calls it (rc_out = None): entries, colptr + c + vbasis (17 B per column), y once; no store
            "k_price": 12 * col_nnz + 17 * n_loc + 8 * m,
        }
        recs = {}
        for name, ms in (("k_score_columns", k1), ("k_score_rows", k2), ("k_price", k10)):
            avg = float(np.mean(ms))
            recs[name] = {"avg_kernel_ms": avg, "min_kernel_ms": float(np.min(ms)), "algorithmic_bytes": int(algo[name]),
                          "achieved_GBps": algo[name] / avg / 1e6, "frac_of_hbm_peak": algo[name] / avg / 1e6 / HBM_PEAK_GBS}
        return recs

    kernels = kernel_records(k1_list, k2_list, k10_list, nnz_loc, sh.row_block.nnz)
    dominant = max(kernels, key=lambda k: kernels[k]["avg_kernel_ms"])

    # HBM bytes per launch of the dominant kernel from the PMC counters: they cannot be read from inside this
    # process, so the figure comes from a committed rocprofv3 --pmc pass of this same command
    # (profiles/rNN/kernel_traffic.json, corrected as MI355X_MICROARCH.md prescribes) and is only used when that
    # pass was taken on these very kernel sources (hash) and this workload; null otherwise
    traffic = None
    try:
        import glob
        # the kernels that can stand behind each of the three walks (the one with launches in the PMC pass ran)
        behind = {"k_score_columns": ("k_score_columns_lw", "k_score_columns"), "k_score_rows": ("k_rb_score_rows", "k_score_rows"),
                  "k_price": ("k_price_lw", "k_price")}
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "kernel_traffic.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("workload") == f"{args.workload}/{structure}" and rec.get("source_hash") == source_hash():
                per = rec.get("traffic_bytes_per_launch", {})
                for name, cands in behind.items():
                    got = next((per[c] for c in cands if c in per), None)
                    kernels[name]["traffic_bytes"] = got
                    kernels[name]["traffic_source"] = os.path.relpath(path, ROOT)
                traffic = kernels[dominant].get("traffic_bytes")
                break
    except Exception:
        traffic = None

    # ---- the same three walks on the no-locality variant of the workload (uniformly random rows): the LDS
    # windows of K1 / K10 and the column-blocked rows of K2 find nothing to hold on to there
    uniform = None
    if rank == 0 and world == 1 and args.workload == "c5" and structure != "uniform" and not args.no_uniform:
        for a in (dC, dR):
            a.free()
        shu = workloads.lp_shard(0, 1, m=m, n_block=n_block, k=k, seed=5, structure="uniform")
        uC, uR = ctx.column_shard(shu.col_block), ctx.row_shard(shu.row_block)
        u_y, u_x = ctx.to_device(shu.y), ctx.to_device(shu.x)
        u_c, u_l, u_u, u_b = (ctx.to_device(v) for v in (shu.c, shu.l, shu.u, shu.b))
        # outputs of their own: the results of the timed run above are still to be checked and reported
        o_sd, o_code = ctx.empty(n_loc, np.float64), ctx.empty(n_loc, np.uint8)
        o_sp, o_flag = ctx.empty(m_loc, np.float64), ctx.empty(m_loc, np.uint8)
        o_price = ctx.empty(24, np.uint8)
        t1, t2, t10 = [], [], []
        for i in range(2 + 5):
            ctx.marker(0)
            ctx.score_columns(uC, u_y, u_c, u_x, u_l, u_u, gamma, o_sd, o_code)
            ctx.marker(1)
            ctx.score_rows(uR, u_x, u_b, u_y, gamma, o_sp, o_flag)
            ctx.marker(2)
            ctx.price(uC, u_y, u_c, vb, 1e-6, None, o_price)
            ctx.marker(3)
            ctx.sync()
            if i >= 2:
                t1.append(ctx.marker_elapsed(0, 1))
                t2.append(ctx.marker_elapsed(1, 2))
                t10.append(ctx.marker_elapsed(2, 3))
        ku = kernel_records(t1, t2, t10, shu.col_block.nnz, shu.row_block.nnz)
        dom_u = max(ku, key=lambda kk: ku[kk]["avg_kernel_ms"])
        uniform = {"workload": "c5 with uniformly random rows (8 strata per column): no locality for the gathers",
                   "kernel": dom_u, "bound": "hbm", "achieved": ku[dom_u]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": ku[dom_u]["frac_of_hbm_peak"], "traffic": None, "kernels": ku,
                   "row_layout": "column-blocked" if uR.rowblock() is not None else "plain walk (auto rule)",
                   "operand_slabs": {"column_walk": uC.slabs(1), "row_walk": None if uR.rowblock() is not None else uR.slabs(0),
                                     "note": "csrc/sx_slabs.h: entries cut by operand index into L2-sized slabs, running sums "
                                             "carried from pass to pass; bit-identical to the plain walk"}}
        for a in (uC, uR):
            a.free()

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_loc / (elapsed / args.steps)

    if use_dist:
        from smart_crossover import distributed as D
        raw = t_gather.cpu().numpy().tobytes()
        recs = [D.unpack_price(raw[r * 48:r * 48 + 24]) for r in range(world)]
        mn, am, bad = D.reduce_price_records(recs, [r * n_loc for r in range(world)])
        cnts = np.sum([np.frombuffer(raw[r * 48 + 24:(r + 1) * 48], dtype=np.int64) for r in range(world)], axis=0)
    else:
        mn, am, bad = ctx.read_price(price)
        cnts = counts.download()

    # ---- optional: iterations of the column-sharded projector CG on this rank's column block
    sharded_cg = None
    if args.cg_iters > 0:
        from smart_crossover import distributed as D
        ops = D.HipOps(ctx, torch)
        A_loc = ops.matrix(sh.col_block.tocsr())                  # both layouts of the m x n_block column block
        rng = np.random.default_rng(7)
        xa = ops.vec(rng.uniform(0.1, 1.0, n_loc))
        xs = ops.vec(np.where(rng.random(m) < 0.5, rng.uniform(0.1, 1.0, m), 0.0))
        st = ops.cg_open(A_loc, xa, xs, ops.vec(sh.c), 1e-30)
        if use_dist and not rehearse:
            dist.all_reduce(st["q"])
        ops.cg_start(st)
        for k in range(3):                                         # warm-up
            ops.cg_local(st)
            if use_dist and not rehearse:
                dist.all_reduce(st["q"])
            ops.cg_update(st, k)
        fence()
        t_cg = time.perf_counter()
        for k in range(args.cg_iters):
            ops.cg_local(st)
            if use_dist and not rehearse:
                dist.all_reduce(st["q"])
            ops.cg_update(st, k + 3)
        fence()
        cg_elapsed = time.perf_counter() - t_cg
        if use_dist:
            tt = torch.tensor([cg_elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cg_elapsed = float(tt.item())
        try:
            ops.cg_finish(st)
        finally:
            ops.cg_close(st)               # the shard handle's device block (~10 m + n vectors)
            free_A = getattr(A_loc, "free", None)
            if free_A:
                free_A()
        cg_bytes = 2 * 12 * nnz_loc + 8 * (3 * m + 2 * n_loc)      # K4 per iteration, this rank's block
        sharded_cg = {"iterations": args.cg_iters, "ms_per_iteration": cg_elapsed / args.cg_iters * 1e3,
                      "allreduce_bytes_per_iteration": 8 * m if world > 1 else 0,
                      "algorithmic_GBps_per_gpu": cg_bytes / (cg_elapsed / args.cg_iters) / 1e9}

    # ---- optional at N > 1: the column-sharded LP re-solve (smart_crossover.distributed.ShardedLP.restricted_resolve):
    # restricted LP replicated and re-solved from its own basis on every rank's GPU (K16s), pricing of the columns
    # outside it rank-local (K1 walk over the rank's block), one all-gather of the records and one of the entering
    # columns per round.  Never fatal: a failure is recorded and the line goes out without it
    sharded_resolve = None
    if use_dist and args.resolve_rows > 0:
        try:
            from smart_crossover import distributed as D
            from smart_crossover.formats import GeneralLP
            rr = args.resolve_rows
            inst2 = workloads.netlib_lp(rr, 10 * rr, seed=17)
            rng2 = np.random.default_rng(18)
            lp2 = GeneralLP(inst2.A, inst2.b, inst2.c + 0.02 * rng2.standard_normal(10 * rr), inst2.l,
                            np.where(np.isinf(inst2.u), 30.0, inst2.u), inst2.sense)
            sh2 = D.ShardedLP(lp2, dist, D.HipOps(ctx, torch))
            tr2 = []
            fence()
            t_rs = time.perf_counter()
            x_R, y_R, R_R, basis_R, status_R, rounds_R = sh2.restricted_resolve(np.flatnonzero(inst2.x > 1e-6), solver="HIP", x_start=inst2.x, y_start=inst2.y,
                                                                                first_method="barrier", batch=2048, opt_tol=1e-6, trace=tr2, max_seconds=45.0)
            fence()
            rs_elapsed = time.perf_counter() - t_rs
            tt = torch.tensor([rs_elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sharded_resolve = {"workload": f"netlib_lp({rr}, {10 * rr}) with another cost vector (c + 0.02 N(0, 1): 6 rounds, 5.7 s in one process; with 0.3 N(0, 1) "
                                           "21 rounds of ~35,000 simplex iterations, 83 s -- tools/sharded_resolve_probe.py); restricted LP = the point's interior columns; "
                                           "stops after 45 s (status TIME_LIMIT)",
                               "status": status_R, "rounds": int(rounds_R), "columns_added": [len(t) for t in tr2],
                               "seconds": float(tt.item()), "objective": float(lp2.c[R_R] @ x_R) if status_R == "OPTIMAL" else None,
                               "exchange": f"per round: all_gather of <= 2048 (|rc|, column) records per rank + all_gather of the entering columns, {world} ranks",
                               "note": "factorisation and tableau replicated (every rank makes the same pivots); pricing sharded"}
        except Exception as exc:
            import traceback
            log(f"[bench] rank {rank}: sharded_resolve failed:\n{traceback.format_exc()}")
            sharded_resolve = {"failed": f"{type(exc).__name__}: {exc}"[:300]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import lp_path as L     # checker / baseline only; never on the product path
        A = sh.row_block                    # world == 1: the whole matrix in CSR
        reps, spent = 0, 0.0
        ref = None
        L.scoring_pass(A, sh.b, sh.c, sh.l, sh.u, sh.x, sh.y)     # warm-up (scipy builds the CSC view lazily)
        while reps < 5 and (spent < args.cpu_seconds or reps == 0):
            t1 = time.perf_counter()
            ref = L.scoring_pass(A, sh.b, sh.c, sh.l, sh.u, sh.x, sh.y)
            spent += time.perf_counter() - t1
            reps += 1
        # the timed GPU pass must agree with the CPU pass it is compared with
        same = (np.array_equal(ref["code"], code.download()) and np.array_equal(ref["rowflag"], flag.download())
                and [ref["fix_low"].size, ref["fix_up"].size, ref["fixed_rows"].size] == [int(v) for v in cnts])
        if not same:
            raise SystemExit("bench: GPU scoring pass disagrees with the CPU oracle")
        cpu = {"value": n_loc / (spent / reps), "unit": "columns/s", "cores": 1, "kind": "port",
               "sample": f"{reps} full scoring passes (K1+K2+np.where x3) of the same {args.workload} workload with the "
                         f"numpy/scipy oracle, {spent / reps * 1e3:.0f} ms each; parity with the GPU pass checked",
               "ms_per_step": spent / reps * 1e3}

    crossover = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_crossover:
        if uniform is None:             # free the resident c5 shard before the crossovers allocate theirs
            for a in (dC, dR):
                a.free()
        exp = os.environ.get("SX_BENCH_EXPERIMENT", "")   # measurement aids for the in-bench slowdown of get_perturb_problem
        if "close_ctx" in exp:                             # ... the scoring context (its stream, 4,096 marker events, arrays) gone
            for name in ("price", "counts", "code", "flag", "s_d", "s_p"):
                v = locals().get(name)
                if hasattr(v, "free"):
                    v.free()
            ctx.close()
        if "gc" in exp:                                    # ... the host arrays of the 1e7-column shard released
            import gc
            sh = None
            gc.collect()
        if "one_thread" in exp:                            # ... numpy's / scipy's worker threads told to stay out
            os.environ["OMP_NUM_THREADS"] = "1"
        if os.environ.get("SX_BENCH_ONLY_LP_1E6"):     # profiling aid (tools/gpu/prof_inbench.sh): the headline leg alone, in bench.py's process state
            rec = _device_lp_crossover(workloads.netlib_lp(), 3, "netlib_lp (the 1e6-variable LP of the metric)")
            print(json.dumps({"lp_1e6_end_to_end": rec}), flush=True)
            return
        if os.environ.get("SX_BENCH_ONLY_LP_C5"):      # ... the config-5-size leg alone, N calls
            rec = _device_lp_crossover(workloads.netlib_lp(1_000_000, 10_000_000), int(os.environ["SX_BENCH_ONLY_LP_C5"]),
                                       "netlib_lp(1e6, 1e7) = config-5 size")
            print(json.dumps({"lp_c5_end_to_end": rec}), flush=True)
            return
        net_c3, net_mcf = crossover_network(), crossover_mcf()
        net_c4 = crossover_mcf(2 ** 17, 2 ** 20, solvers=("HIP", "HGS") if args.c4_highs else ("HIP",), repeats=1)
        if not args.c4_highs:   # the HiGHS leg as recorded on one of the pool's boxes (same instance, same code path)
            rec_path = os.path.join(ROOT, "profiles", "r02", "netdual_c4_highs.jsonl")
            if os.path.exists(rec_path):
                rec = json.loads(open(rec_path).readline())
                net_c4["host_solver_ms_recorded"] = rec["wall_ms"]
                net_c4["simplex_iterations_HGS_recorded"] = rec["simplex_iterations"]
                net_c4["recorded_in"] = "profiles/r02/netdual_c4_highs.jsonl (python bench.py --c4-highs re-measures it)"
                if abs(rec["cost"] - net_c4["optimal_cost"]) > 1e-9 * (1 + abs(rec["cost"])):
                    raise SystemExit("bench: the device's config-4 optimum differs from the recorded HiGHS optimum")
        crossover = {"lp_1e6_end_to_end": crossover_lp_1e6(args.lp_highs, args.lp_cpu_path),
                     "lp_2e4_rows": crossover_lp_2e4(args.lp_2e4_cpu),
                     "lp_c2_host_path": crossover_host_path(args.cpu_seconds),
                     "lp_c2_end_to_end": crossover_lp_end_to_end(args.highs_seconds, args.lp_cpu_path),
                     "lp_c5_get_perturb_problem": crossover_lp_c5(),
                     "network_c3": net_c3,
                     "network_mcf_4096": net_mcf,
                     "network_c4": net_c4}
        if not args.no_c5_crossover:
            crossover["lp_c5_end_to_end"] = crossover_lp_c5_end_to_end()

    if rank == 0:
        # the metric's other half where the driver's record keeps it: top level AND inside config (the driver's parsed
        # copy keeps the contract keys whole and only the names of the others)
        wall = {}
        if crossover is not None:
            h = crossover["lp_1e6_end_to_end"]
            wall = {"crossover_wall_ms": h["gpu_ms"], "crossover_wall_ms_back_to_back": h.get("gpu_ms_back_to_back"),
                    "crossover_config": "netlib_lp 1e5 rows x 1e6 columns (the 1e6-variable LP of BASELINE's metric), interior point "
                                        "in host memory -> optimal vertex + basis in host memory, 1 GPU, third call of a warm process (started 0.25 s "
                                        "after the second returned; the second, right behind the first: crossover_wall_ms_back_to_back)"}
            if "lp_c5_end_to_end" in crossover:
                wall["crossover_wall_ms_c5_size"] = crossover["lp_c5_end_to_end"]["gpu_ms"]
            s2 = crossover["lp_2e4_rows"]
            if "speedup_total" in s2:
                wall["crossover_speedup_vs_cpu_same_host"] = s2["speedup_total"]
                wall["crossover_speedup_config"] = ("netlib_lp 2e4 rows x 2e5 columns: device %.0f ms, CPU path %.1f s on this host's "
                                                    "cores in this run (oracle + numpy first-order stage + warm-started HiGHS simplex)"
                                                    % (s2["gpu_ms"], s2["cpu_path"]["cpu_total_s"]))
        out = {
            "metric": "columns_scored_per_sec", "value": value, "unit": "columns/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "rows": m, "cols_per_gpu": n_loc, "cols_total": n_tot,
                       "nnz_per_gpu": int(nnz_loc), "step": "K1 score_columns + K2 score_rows + 3x select_indices + K10 price"
                                                            + (" + all_gather(48 B: pricing record + 3 set sizes)" if world > 1 else ""),
                       "parallelism": f"column/row blocks over {world} GPU(s)", **wall},
            **wall,
            # the walk with the largest share of the step; all three under "kernels" (HIP events, same run)
            "roofline": {"kernel": dominant, "bound": "hbm", "achieved": kernels[dominant]["achieved_GBps"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernels[dominant]["frac_of_hbm_peak"],
                         "traffic": traffic, "algorithmic_bytes": kernels[dominant]["algorithmic_bytes"],
                         "avg_kernel_ms": kernels[dominant]["avg_kernel_ms"],
                         "min_kernel_ms": kernels[dominant]["min_kernel_ms"]},
            "kernels": kernels,
            "roofline_uniform": uniform,
            "cpu_baseline": cpu,
            "sharded_cg": sharded_cg,
            "sharded_pricing": sharded_pricing,
            "sharded_resolve": sharded_resolve,
            "crossover": crossover,
            "result": {"fix_low": int(cnts[0]), "fix_up": int(cnts[1]), "fixed_rows": int(cnts[2]),
                       "min_rc": mn, "argmin": am, "n_violating": bad},
            "device": dev_name,
        }
        print(json.dumps(out), flush=True)
        if wall:       # the tail of the driver's record ends here: the numbers of the metric's first half once more
            log("[bench] " + json.dumps(wall))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
