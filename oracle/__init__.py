"""Parity oracle for the MOIHGP hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  PARITY UNPINNED (see oracle/README.md).
"""
