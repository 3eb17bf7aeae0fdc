"""CPU oracle for the LSNet hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(as the checker / the timed CPU baseline).  Nothing under lsnet_amd/ imports it.
"""
