"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MaxSim rerank path (torch/numpy restatement in
maxsim_oracle.py, plain-C restatement in maxsim_oracle.c + c_oracle.py).

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported baseline -- never as a fallback for the HIP path.
"""
