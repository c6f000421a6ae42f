"""meryl_amd -- MI355X-native `meryl count` engine.

The product is the C-ABI shared library built from meryl_amd/csrc (see
include/meryl_gpu_count.h).  This package is only the Python plumbing the
tests and bench.py use to reach that ABI: `capi` (ctypes binding) and `count`
(single- and multi-GPU orchestration over torch device buffers /
torch.distributed).  Nothing in here computes k-mers on the CPU.
"""
__version__ = "0.1"
