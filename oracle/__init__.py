"""CPU oracle for the ivid sampling hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline; the product (ivid_amd) never does.
"""
