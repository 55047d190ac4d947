"""Host-side thread budget.

A GPU box hands a job a CPU *quota* (cgroup ``cpu.max``, e.g. 16 cores' worth of a 256-core host) while ``os.cpu_count()``
still reports every core of the host.  numpy's BLAS, torch's intra-op pool and OpenMP then start one spinning worker per
reported core; their spin-waits burn the quota and the kernel throttles the WHOLE process group until the next 100 ms
scheduler period -- including the thread that enqueues kernels and the one that waits for them.  Measured on an MI355X box
(round 2, tools/bench_warm_diag.py): the GPU finished 20 rollout steps in 38 ms by its own clock, the host saw it at 80-100 ms
in 6 of 10 processes; with the pools capped, 0 of 6 (cpu.stat nr_throttled stops moving).  The reference has the same
hazard in mirror image (one OS process per env, subproc_vec_env.py:35-58) and sidesteps it by running few envs.

``apply()`` caps the pools at the quota.  It must run before numpy / torch are imported to catch the pools at creation;
pools that already exist are narrowed through threadpoolctl / torch.set_num_threads.  SUMO_HOST_THREADS=<n> overrides the
count, SUMO_HOST_THREADS=0 leaves everything alone.
"""
import os
import sys

_VARS = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS")
_applied = None


def cpu_quota():
    """Cores this process may use: min(affinity mask, cgroup v2/v1 CPU quota).  ``None`` parts are skipped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                        # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:                                                    # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def throttle_stats():
    """(nr_throttled, throttled_usec) of this cgroup, or None: lets a benchmark say whether the scheduler held it back."""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None


def apply(threads=None):
    """Cap the host thread pools; returns the thread count in force (0 = disabled).  Idempotent."""
    global _applied
    if _applied is not None and threads is None:
        return _applied
    env = os.environ.get("SUMO_HOST_THREADS")
    if threads is None and env is not None:
        threads = int(env)
        if threads == 0:
            _applied = 0
            return 0
    if threads is None:
        # half the quota for pooled workers: the enqueueing thread, the HIP runtime's helper threads and the waiters need the rest;
        # the ranks of one node (torchrun: LOCAL_WORLD_SIZE) share the job's cgroup, so each takes its share of that half
        try:
            ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            ranks = 1
        threads = max(1, cpu_quota() // (2 * ranks))
    for v in _VARS:
        os.environ.setdefault(v, str(threads))
    if "numpy" in sys.modules:                                  # pools created before us: narrow them in place
        try:
            import threadpoolctl
            threadpoolctl.threadpool_limits(limits=threads)
        except Exception:
            pass
    if "torch" in sys.modules:
        try:
            import torch
            if torch.get_num_threads() > threads:
                torch.set_num_threads(threads)
        except Exception:
            pass
    _applied = threads
    return threads


_captures_open = 0      # HIP graph captures in progress in this process (gc_paused blocks)


def capturing():
    return _captures_open > 0


def drop_graphs(graphs):
    """The ONE place captured HIP graphs are released (``graphs``: a dict of capture records).  Destroying a graph while a stream of
    the process is capturing aborts the process; the cyclic collector is held off by ``gc_paused``, and this guard covers the other
    route -- a reference count reaching zero inside a capture block."""
    if _captures_open:
        raise RuntimeError("a captured HIP graph would be destroyed while a graph capture is open (hostcfg.gc_paused): "
                           "release graphs before entering the capture block")
    graphs.clear()


class gc_paused(object):
    """Context manager for HIP graph capture: collect garbage now, then keep the cyclic collector off until the block ends.  A
    collection that happens to run inside a capture may finalise device objects of an earlier model (its captured graphs, streams,
    events); destroying those while a stream is capturing aborts the process.  While the block is open ``capturing()`` is true and
    ``drop_graphs`` refuses to release graphs."""

    def __enter__(self):
        import gc
        global _captures_open
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        _captures_open += 1
        return self

    def __exit__(self, *exc):
        import gc
        global _captures_open
        _captures_open -= 1
        if self._was:
            gc.enable()
        return False
