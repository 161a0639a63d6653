"""How many CPUs this process may really use, and a way to keep the numeric libraries' thread pools inside that share.

A container usually sees every CPU of its host (256 on the MI355X boxes) while its cgroup grants a fraction (16 per GPU there).  Pools sized
by the visible count — OpenBLAS / OpenMP under numpy, torch's intra-op pool, a worker per "core" — burn the quota of a scheduling period in a
few milliseconds of spinning, and the kernel then stops the WHOLE process until the next period: measured in round 4 as 15-25 ms stalls at
whatever call came next (a hipStreamSynchronize, a D2H copy) after any multi-threaded host stage of a job.  No numpy import here: call
``limit_thread_pools()`` before numpy is first imported for the environment variables to count."""
from __future__ import annotations

import os


def cpu_share() -> int:
    """CPUs granted to this process: the cgroup's quota / period (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``), the affinity mask, the visible count
    — the smallest of them, at least 1."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = int(fq.read()), int(fp.read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def limit_thread_pools(n: int | None = None) -> int:
    """Cap the BLAS / OpenMP pools (environment defaults, so only before numpy's first import) and torch's intra-op pool (if torch is already
    loaded) at ``n`` (default: ``cpu_share()``).  Values the user has set are left alone.  Returns the cap."""
    import sys
    n = int(n or cpu_share())
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(var, str(n))
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.get_num_threads() > n:
                torch.set_num_threads(n)
        except Exception:
            pass
    return n
