"""Import-path shim: `from tgcn.nn.gcn import TGCNCheb, ...` resolves to the MI355X-native modules.

Put <repo>/compat ahead of the reference on PYTHONPATH; model code that imports
tgcn.nn.gcn then runs on the HIP path unchanged."""
from tgcn_amd.nn import (ChebConv, ChebTimeConv, GCNCheb, TGCNCheb, TGCNCheb_H, gcn_pool, gcn_pool_4, spmm,  # noqa: F401
                         spmm_batch_2, spmm_batch_3, uniform)
