"""MI355X-native implementation of the gradient-guided `--attack` loop of Facico/style-transfer.

Python host (flags, data, checkpoints, log format) over a C-ABI HIP library (include/st_kernels.h).
"""
__all__ = ["_lib", "weights", "ops", "data", "attack", "logfmt"]
