"""oracle/ -- CPU restatement of the reference's image<->mesh projection path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(geograypher_amd/) never does.  See oracle/oracle_raster.c (rasterization rule-set, parity status) and
oracle/oracle_np.py (numpy stages, pinned against the real reference through tests/golden/).
"""
