"""Import-path shim for `import gcn.graph as graph` (reference: gcn/graph.py).

`chebyshev(L, X, K)` (gcn/graph.py:241-283) is the hot-path function and runs on the HIP path
(tgcn_amd/numpy_api.py).  Every other name -- grid, distance_sklearn_metrics, adjacency, laplacian, rescale_L,
fourier, lanczos ... (gcn/graph.py:10-238, out of scope: SURVEY.md section 2 #4) -- resolves to the reference's own
gcn/graph.py found behind compat/ on sys.path, so scripts that mix both (examples/tgcn_mnist.py:52-54,173-175,194)
run unchanged.  Without a reference checkout only `chebyshev` is available and the others raise an ImportError
that says so."""
from tgcn_amd.numpy_api import chebyshev  # noqa: F401

from _tgcn_amd_delegate import install as _install

_install(globals(), "gcn", "graph", native=("chebyshev",))
