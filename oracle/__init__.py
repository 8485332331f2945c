"""CPU restatement of the reference's algorithm: TEST INFRASTRUCTURE ONLY (parity unpinned, see mvsnet_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
