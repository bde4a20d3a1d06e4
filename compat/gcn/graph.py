"""Import-path shim for the numpy scripts: `gcn.graph.chebyshev(L, X, K)` on the HIP path.

Only the hot-path function is provided; graph construction helpers (grid, adjacency, laplacian, ...) stay with
the reference -- import those from the reference's own gcn.graph."""
from tgcn_amd.numpy_api import chebyshev  # noqa: F401
