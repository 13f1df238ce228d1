"""zdw_amd — MI355X (gfx950) implementation of adobe/zdw's TSV -> ZDW block encode.

The product is native: zdw_amd/lib/libzdw_hip.so (hand-written HIP kernels behind the C ABI of
include/zdw_hip.h) and zdw_amd/lib/convertDWfile (the reference's command line on top of it).
This package is thin plumbing for tests, the benchmark and the multi-GPU driver:

    zdw_amd.capi    ctypes binding of include/zdw_hip.h (class Encoder)
    zdw_amd.cli     the convertDWfile command line, in-process
    zdw_amd.build   compiles the native code in-tree
    zdw_amd.sharded one-block-over-N-GPUs driver on torch.distributed (RCCL)
    zdw_amd.synth   synthetic TSV generators of the benchmark shapes
"""
__all__ = ["capi", "cli", "build", "sharded", "synth"]
