"""CPU oracle for the robust-aggregation hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it, and only as the checker / the timed CPU baseline -- never as a compute path.
The product (``attacking_federate_learning_amd``) raises if its HIP library is
missing; it never falls back to anything in here.

Two tiers:

* ``oracle.faithful`` -- restates ``/root/reference/defences.py`` and
  ``malicious.py`` operation by operation in numpy fp32 (same numpy calls, same
  evaluation order, same tie rules).  Pinned bit-for-bit against the reference
  itself (``tests/test_oracle_vs_reference.py``, runs where ``/root/reference``
  exists) and against golden vectors minted from the reference
  (``tests/golden/*.npz``, made by ``tests/golden/make_golden.py``).
* ``oracle.ideal`` -- the same selection rules evaluated in fp64 with vectorised
  numpy (dgemm Gram, ``np.sort``).  It is what large GPU runs are compared with,
  together with the decision margins of SURVEY.md section 8(d).

The reference ships no tests, fixtures or golden vectors of its own (SURVEY.md
section 4), so the pin is "outputs of the reference run in this image"
(numpy 2.2.6 + OpenBLAS 0.3.29).
"""
