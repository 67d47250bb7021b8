"""CPU checkers for the pixel merger.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; ``mergenet_amd`` never does.
"""
