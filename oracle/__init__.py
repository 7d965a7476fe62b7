"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MaxSim rerank path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported baseline -- never as a fallback for the HIP path.
"""
