"""CPU oracle for the CDML triplet-embedding hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``cdml_amd``) never imports this package and has no CPU fallback.
"""
