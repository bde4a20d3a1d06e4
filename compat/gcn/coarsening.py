"""Import-path shim: `from gcn import coarsening` / `from gcn.coarsening import coarsen, perm_data` (reference: gcn/coarsening.py).

All six public functions of the reference's module are provided natively (tgcn_amd/coarsening.py); any other name falls
through to the reference's file behind compat/ on sys.path."""
from tgcn_amd.coarsening import coarsen, compute_perm, metis, perm_adjacency, perm_data, perm_data_device, graclus_match as metis_one_level  # noqa: F401

from _tgcn_amd_delegate import install as _install

_install(globals(), "gcn", "coarsening", native=("coarsen", "compute_perm", "metis", "perm_adjacency", "perm_data", "perm_data_device",
                                                "metis_one_level"))
