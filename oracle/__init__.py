"""CPU oracle for the MADiff hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` may be imported by the product (``madiff_amd``/``diffuser``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / the timed CPU baseline.
"""
