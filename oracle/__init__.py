"""CPU oracle for the MCD hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32/fp64, CPU) restatement of the reference's
algorithm for the path SURVEY.md section 8 names: DRN / MFNet encoder-decoder,
the two losses and the three-step MCD update.  It exists so that the HIP
product path can be checked against something that follows the reference line
by line.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package must never do so.

Parity pin: ``tests/golden/make_golden.py`` imports the *real* reference from
``/root/reference`` (in the build container only, through an in-memory 2to3
shim), checks every function here against it and writes the golden vectors in
``tests/golden/*.npz|json``; ``tests/test_oracle_golden.py`` re-checks the
oracle against those vectors wherever the test-suite runs.
"""
