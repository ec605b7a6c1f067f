"""graphmat_amd: MI355X-native generalized-SpMV engine behind GraphMat's vertex-program surface.

Python here is host-side plumbing only (ctypes binding of the C-ABI library,
edge-list I/O, synthetic graph generators, multi-GPU sharding glue over
torch.distributed).  All compute runs in csrc/*.hip through libgraphmat_hip.so.
"""
__all__ = ["mtx", "generators"]
