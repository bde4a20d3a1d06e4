"""tgcn_amd -- MI355X-native Chebyshev (time-)graph convolution behind the API of cassianobecker/tgcn's tgcn.nn."""
from . import functional  # noqa: F401
from .graph import GraphOperand  # noqa: F401
from .nn import (ChebConv, ChebTimeConv, GCNCheb, TGCNCheb, TGCNCheb_H, cheb_relu_pool, gcn_pool, gcn_pool_4, spmm,  # noqa: F401
                 spmm_batch_2, spmm_batch_3, uniform)
