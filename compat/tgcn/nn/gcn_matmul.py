"""Import-path shim for tgcn.nn.gcn_matmul (same three dense-L classes; accepts torch-sparse L like the original)."""
from tgcn_amd.nn import GCNCheb, TGCNCheb, TGCNCheb_H, gcn_pool, gcn_pool_4, uniform  # noqa: F401
