"""ORACLE — CPU restatements of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import anything
from this package; the product (spheremanopt_amd/) never does.  See oracle/README.md.
"""
