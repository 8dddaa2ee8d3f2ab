"""gnndelete_amd - MI355X (gfx950) native engine for the GNNDelete unlearning hot path.

Layout (only what the path needs):
  csrc/        hand-written HIP kernels + the C ABI declared in include/gnndelete_hip.h
  lib/         libgnndelete_hip.so (built in-tree by __graft_entry__.build())
  _lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
  graph.py     CSR construction from the reference's edge_index layout
  ops.py       torch.autograd.Function wrappers around the C entries
  nn.py        GCNConv / GATConv / GINConv / RGCNConv with torch_geometric's parameter names
  framework/   host-side mirror of the reference's `framework` package (same names/signatures)
"""
__version__ = '0.1.0'
