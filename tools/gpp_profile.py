#!/usr/bin/env python3
"""Where the time of get_perturb_problem goes when it follows a re-solve (bench.py's order): [gpp, resolve] x 2 on the
headline LP, host-side profile of the second gpp.  Development tool."""
import cProfile
import io
import os
import pstats
import sys
import time
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))
import workloads  # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from smart_crossover.lp_methods import algorithms as alg  # noqa: E402
from smart_crossover.solver_caller.caller import SolverSettings  # noqa: E402
from smart_crossover.solver_caller import solving  # noqa: E402

keep = []
if "torch" in sys.argv:     # bench.py brings torch's HIP runtime up first (sharded CG / pricing legs)
    import torch
    torch.cuda.set_device(0)
    keep.append(torch.cuda.Stream())
if "dist" in sys.argv:      # ... and imports the sharded driver
    from smart_crossover import distributed  # noqa: F401
if "prelude" in sys.argv:   # bench.py's state: a second context with the config-5 shards resident and one scoring step done
    import numpy as np
    from smart_crossover.hip import Context
    sh = workloads.lp_shard(0, 1)
    c5 = Context(0)
    dC, dR = c5.column_shard(sh.col_block), c5.row_shard(sh.row_block)
    d = {k: c5.to_device(getattr(sh, k)) for k in ("y", "x", "c", "l", "u", "b")}
    s_d, code = c5.empty(sh.col_block.shape[1], np.float64), c5.empty(sh.col_block.shape[1], np.uint8)
    c5.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
    c5.sync()
    keep = [sh, c5, dC, dR, d, s_d, code]
    if "k2" in sys.argv:     # the row walk too: its first call builds the column-blocked layout (GBs of sort temporaries)
        s_p, flag = c5.empty(sh.row_block.shape[0], np.float64), c5.empty(sh.row_block.shape[0], np.uint8)
        c5.score_rows(dR, d["x"], d["b"], d["y"], 1e-3, s_p, flag)
        c5.sync()
        keep += [s_p, flag]
    if "k10" in sys.argv:    # pricing rounds
        vb = c5.to_device(np.where(np.arange(sh.col_block.shape[1]) % 7 == 0, -2, -1).astype(np.int8))
        res = None
        for _ in range(50):
            res = c5.price(dC, d["y"], d["c"], vb, 1e-6, None, res)
        c5.sync()
        keep += [vb, res]
    if "oracle" in sys.argv: # the numpy / scipy baseline leg
        from oracle import lp_path as L
        for _ in range(3):
            L.scoring_pass(sh.row_block, sh.b, sh.c, sh.l, sh.u, sh.x, sh.y[:sh.row_block.shape[0]])
    if "load" in sys.argv:   # half a minute of the scoring walks first, as bench.py's timed and profiled loops do
        t_end = time.perf_counter() + 30.0
        while time.perf_counter() < t_end:
            for _ in range(200):
                c5.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
            c5.sync()
    if "free" in sys.argv:
        dC.free(); dR.free()
inst = workloads.netlib_lp()
for rep in range(3):
    lp = GeneralLP(inst.A, inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        pr.enable()
        mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        pr.disable()
    t1 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
        caller.read_genlp(mgr.lp_sub)
        caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
        caller.run_barrier()
        out = caller.return_output()
    t2 = time.perf_counter()
    print(f"rep {rep}: get_perturb_problem {1e3 * (t1 - t0):.1f} ms, resolve {1e3 * (t2 - t1):.1f} ms", flush=True)
    if rep == 2:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
        print(s.getvalue())
