"""CPU oracle for the HoVer-Net hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (see hvn_oracle.c header).  The product (hover_net_amd/) never does.
"""
