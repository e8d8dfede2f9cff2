"""Parity oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this package, and only as the checker / reported baseline.  The renderer
(raytrace_amd) never imports it.
"""
