"""oracle/ -- TEST INFRASTRUCTURE (CPU restatement of the reference hot path + reference build recipe).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product under so-net_amd/ never does and fails loudly without its HIP library.
"""
