"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's hot path.

Nothing under ``oracle/`` is imported by the product package ``unet_zoo_amd``; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker
/ the CPU baseline, never as the thing measured or shipped.
"""
