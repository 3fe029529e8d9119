"""nuts333_amd -- CPU-only baseline harness for ToKe79/nuts333 (NUTS 3.3.3 telnet talker).

The reference has no data-parallel numeric hot path (BASELINE.json
``north_star``): this package therefore contains no HIP kernels and no RCCL code.  It holds
what the north star asks for -- a scratch-tree provisioner, a talker launcher, the
closed-loop load generator and the five BASELINE configurations.
"""
__all__ = ["provision", "talker", "workloads"]
