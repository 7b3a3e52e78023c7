"""Parts of bench.py (the driver-facing CLI stays there): roofline helpers, synthetic workloads, boundary timing, the other
single-GPU configurations of the `configs` array, the N > 1 scaling entries."""
