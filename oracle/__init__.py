"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the zopt hot path.

Nothing under ``zopt_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
there only as the checker / the timed CPU baseline, never as the product.
"""
