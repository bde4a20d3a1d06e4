"""Import-path shim: `from gcn import coarsening` / `from gcn.coarsening import coarsen, perm_data` (reference: gcn/coarsening.py)."""
from tgcn_amd.coarsening import coarsen, compute_perm, metis, perm_adjacency, perm_data, perm_data_device, graclus_match as metis_one_level  # noqa: F401
