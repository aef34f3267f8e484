"""Host BLAS / OpenMP thread pools sized to the CPU quota of the process.

The device path is driven by ONE host thread that launches kernels and polls events (the projector CG replays 24
iterations, polls, replays).  numpy's OpenBLAS and torch's OpenMP size their pools by the CPUs they SEE -- 256 on an
MI355X host -- while a container usually OWNS far fewer (cgroup ``cpu.max``: 16 on the measured boxes).  After any threaded
BLAS call (``c @ x`` over 1e6 variables in the reference's gap test, lp_methods/algorithms.py:63) the pool's workers spin
for ~0.1 s looking for more work; 64 spinning threads on a 16-CPU quota get the whole process throttled by the scheduler,
the launching thread included: the next crossover's projector CG lost ~90 ms to launch gaps
(profiles/r04/in_bench_slowdown.md -- with one BLAS thread 403 ms back to back instead of 470).  The reference leaves
threading to Gurobi; here the pools are cut to what the quota can run beside the launching thread.

``SX_BLAS_THREADS``: unset / ``auto`` -> min(current, max(1, quota // 4)); ``0`` -> leave the pools alone; N -> N threads.
"""
from __future__ import annotations

import math
import os
from typing import Optional

_applied: Optional[dict] = None


def cpu_quota() -> int:
    """CPUs this process may use: the smaller of its affinity mask and its cgroup quota (v2 ``cpu.max``, v1 cfs files)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - not Linux
        cpus = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                cpus = min(cpus, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return cpus


def fit_to_quota(force: bool = False) -> dict:
    """Cut the BLAS / OpenMP pools loaded in this process (once; ``force`` to do it again after more libraries were
    loaded).  Returns {"quota": CPUs owned, "limit": threads set or None, "pools": [(library, before, after)]}."""
    global _applied
    if _applied is not None and not force:
        return _applied
    quota = cpu_quota()
    want = os.environ.get("SX_BLAS_THREADS", "auto").strip().lower()
    rec = {"quota": quota, "limit": None, "pools": []}
    if want == "0":
        _applied = rec
        return rec
    try:
        from threadpoolctl import ThreadpoolController
    except ImportError:  # (nothing to do it with: the environment variables of the pools still work)
        _applied = rec
        return rec
    limit = int(want) if want.isdigit() else max(1, quota // 4)
    for lib in ThreadpoolController().lib_controllers:
        before = lib.num_threads
        # auto only ever cuts: a pool the user already set smaller stays as it is
        if before and (want.isdigit() or before > limit) and before != limit:
            lib.set_num_threads(limit)
            rec["limit"] = limit
        rec["pools"].append((os.path.basename(lib.filepath or "?"), before, lib.num_threads))
    _applied = rec
    return rec
