"""CPU oracle of the hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py, never by the
product package."""
