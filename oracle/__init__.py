"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the JSPSR hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The shipped path is ``jspsr_amd`` (HIP kernels behind a C ABI); it never imports this
package and raises if its HIP library is missing.

Parity pin: the reference has no tests or golden vectors of its own, so
the oracle is pinned by fixtures generated in the build container by importing the
reference's own modules (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``), with
``torchvision.ops.deform_conv2d`` (third-party, torchvision 0.16, not vendored and not
installed) supplied by an independent ``grid_sample``-based stand-in.
"""


def host_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup's CPU quota.  On the GPU boxes of
    this pool 256 CPUs are visible and torch starts 128 intra-op threads, while the container's quota is 16 CPUs
    (/sys/fs/cgroup/cpu.max "1600000 100000"): oversubscribed eight-fold, the oracle runs 7.7 x SLOWER than on 16 threads
    (tools/cpu_probe.py: 2.79 s vs 0.36 s per fp64 step).  The tests and bench.py's cpu_baseline leg size torch's CPU
    thread pool with this."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                                          # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n
