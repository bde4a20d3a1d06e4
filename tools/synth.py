"""Seeded synthetic graphs of the BASELINE.json shapes, generated on the device with torch (bench / test support;
not part of the product path).  All return (n, row, col, val) COO of L-hat = -D^-1/2 A D^-1/2 (zero diagonal)."""
import torch

RMAT_ABCD = (0.57, 0.19, 0.19, 0.05)


def _rmat_pairs(m, scale, gen, device, abcd=RMAT_ABCD):
    a, b, c, _ = abcd
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, device=device, generator=gen)
        bs = (r >= a + b)
        bd = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        src = (src << 1) | bs.long()
        dst = (dst << 1) | bd.long()
    return src, dst


def normalized_laplacian_coo(n, u, v):
    """Undirected unit-weight pairs (u < v) -> symmetric COO of -D^-1/2 A D^-1/2."""
    row = torch.cat([u, v])
    col = torch.cat([v, u])
    deg = torch.bincount(row, minlength=n).to(torch.float32)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    val = -(dis[row] * dis[col])
    return row, col, val


def relabel(n, row, col, labeling, gen):
    """'natural' keeps generator ids; 'random' applies a random permutation (Graph500 style, worst case for
    locality); 'degree' sorts vertices by decreasing degree (hubs first)."""
    if labeling == "natural":
        return row, col
    if labeling == "random":
        perm = torch.randperm(n, device=row.device, generator=gen)
    elif labeling == "degree":
        deg = torch.bincount(row, minlength=n)
        order = torch.argsort(deg, descending=True, stable=True)
        perm = torch.empty_like(order)
        perm[order] = torch.arange(n, device=row.device)
    else:
        raise ValueError(labeling)
    return perm[row], perm[col]


def rmat(n, nnz, seed=12345, labeling="random", device="cuda", scale=None):
    """R-MAT (0.57,0.19,0.19,0.05): endpoints >= n rejected, self loops dropped, symmetrised, de-duplicated,
    exactly nnz stored entries (nnz/2 undirected pairs)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    if scale is None:
        scale = max(1, (n - 1).bit_length())
    need = nnz // 2
    keys = torch.zeros(0, dtype=torch.int64, device=device)
    while keys.numel() < need:
        m = min(int((need - keys.numel()) * 1.8) + 1024, 64_000_000)
        s, d = _rmat_pairs(m, scale, gen, device)
        ok = (s < n) & (d < n) & (s != d)
        s, d = s[ok], d[ok]
        k = torch.minimum(s, d) * n + torch.maximum(s, d)
        del s, d, ok
        keys = torch.unique(torch.cat([keys, k]))
    if keys.numel() > need:
        keys = keys[torch.randperm(keys.numel(), device=device, generator=gen)[:need]]
    u, v = keys // n, keys % n
    del keys
    u, v = relabel(n, u, v, labeling, gen)
    row, col, val = normalized_laplacian_coo(n, u, v)
    return n, row, col, val


def sheet_mesh(side=300, long_range=0.10, seed=12345, device="cuda"):
    """HCP-style cortical-sheet stand-in: side x side triangulated sheet, 10 in-plane neighbours
    ((+-1,0),(0,+-1),(+1,+1),(-1,-1),(+-2,0),(0,+-2)) + `long_range` * n random long-range pairs."""
    gen = torch.Generator(device=device).manual_seed(seed)
    n = side * side
    i, j = torch.meshgrid(torch.arange(side, device=device), torch.arange(side, device=device), indexing="ij")
    i, j = i.flatten(), j.flatten()
    us, vs = [], []
    for di, dj in ((1, 0), (0, 1), (1, 1), (2, 0), (0, 2)):
        ok = (i + di < side) & (j + dj < side)
        us.append((i * side + j)[ok])
        vs.append(((i + di) * side + (j + dj))[ok])
    m = int(long_range * n)
    a = torch.randint(0, n, (m,), device=device, generator=gen)
    b = torch.randint(0, n, (m,), device=device, generator=gen)
    ok = a != b
    us.append(torch.minimum(a, b)[ok])
    vs.append(torch.maximum(a, b)[ok])
    keys = torch.unique(torch.cat(us) * n + torch.cat(vs))
    u, v = keys // n, keys % n
    row, col, val = normalized_laplacian_coo(n, u, v)
    return n, row, col, val


def grid_knn(side=28, device="cuda"):
    """MNIST-style grid: 8-neighbour king-move graph on side x side pixels (n=784 at side 28).  The reference's
    8-NN construction (gcn/graph.py:22-82) on a regular grid selects exactly these neighbours away from the border."""
    n = side * side
    i, j = torch.meshgrid(torch.arange(side, device=device), torch.arange(side, device=device), indexing="ij")
    i, j = i.flatten(), j.flatten()
    us, vs = [], []
    for di, dj in ((1, 0), (0, 1), (1, 1), (1, -1)):
        ok = (i + di < side) & (j + dj < side) & (j + dj >= 0)
        us.append((i * side + j)[ok])
        vs.append(((i + di) * side + (j + dj))[ok])
    u, v = torch.cat(us), torch.cat(vs)
    row, col, val = normalized_laplacian_coo(n, torch.minimum(u, v), torch.maximum(u, v))
    return n, row, col, val


def banded(n, nnz, width=4096, seed=12345, device="cuda"):
    """Graph with the sizes of the R-MAT workload but perfect locality: every vertex has nnz/n neighbours drawn within +-width of
    its own index (symmetrised, de-duplicated).  Upper bound for the hop kernel: the gathers of a row block hit in L2."""
    gen = torch.Generator(device=device).manual_seed(seed)
    m = nnz // 2
    u = torch.randint(0, n, (int(m * 1.05),), device=device, generator=gen)
    off = torch.randint(1, width + 1, (u.numel(),), device=device, generator=gen)
    v = u + off
    ok = v < n
    keys = torch.unique(u[ok] * n + v[ok])[:m]
    u, v = keys // n, keys % n
    row, col, val = normalized_laplacian_coo(n, u, v)
    return n, row, col, val
