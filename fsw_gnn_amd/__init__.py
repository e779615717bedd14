"""fsw_gnn_amd: MI355X-native (gfx950) hot path of the Fourier Sliced-Wasserstein GNN layer.

Public surface mirrors the reference modules fsw_embedding.py / fsw_conv.py:
    FSW_embedding, segcumsum          (fsw_gnn_amd.fsw_embedding)
    FSW_conv, FSW_readout             (fsw_gnn_amd.fsw_conv)
All computation runs in hand-written HIP kernels (csrc/, libfsw_hip.so, C ABI in include/fsw_hip.h).
"""
from .fsw_embedding import FSW_embedding, segcumsum  # noqa: F401
from .fsw_conv import FSW_conv, FSW_readout  # noqa: F401
from .graph import CSRGraph, build_csr  # noqa: F401
