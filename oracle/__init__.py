"""CPU oracle for the LFPSQP hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
reported baseline.  The product package (``lfpsqp.jl_amd``) never imports it.

Parity status: the reference (ksil/LFPSQP.jl) is Julia-only and no ``julia``
binary exists in the build image, so the restatement cannot be pinned against
outputs of the reference itself.  It is pinned against
  * the only golden numbers the reference holds (README.md:31-36, Rosenbrock),
  * seeded re-creations of every property the reference's own tests assert
    (test/test_cg.jl, test_retractions.jl, test_inequalities.jl,
    test_linesearch.jl),
see tests/test_oracle_*.py.  For configs 2-5 trajectory parity against the
Julia reference itself is therefore "parity unpinned" (FINDINGS.md §3).
"""
