"""CPU oracle for the haloop hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``haloop_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning: every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were generated in the build
container by importing the reference itself (``tests/golden/make_golden.py``).
"""
