"""Shadow of the reference's `gcn` package: graph.chebyshev and coarsening.* are native; `gcn.models`, `gcn.utils` and the
remaining names of `gcn.graph` come from the reference checkout behind compat/ on sys.path (compat/_tgcn_amd_delegate.py)."""
from _tgcn_amd_delegate import extend_package_path as _extend

_extend(__path__, "gcn")
