"""TEST INFRASTRUCTURE — CPU restatement of the AfterQC hot path (see oracle/aqc_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
