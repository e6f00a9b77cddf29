"""CPU oracle for the malva-geno hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product under malva_amd/ never does.
"""
