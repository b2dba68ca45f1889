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
