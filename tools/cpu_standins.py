"""numpy / scipy stand-ins for the compute calls of tgcn_amd/dist.py -- TEST INFRASTRUCTURE, never a product path.

tgcn_amd/dist.py keeps its communication logic (partition, halo lists, exchanges, overlap, the transposed shard, the gradient
all-reduce) apart from the arithmetic behind one small interface (`HipOps` there: operand / hop / project / project_first / pack /
wgrad / fold / weight_layout).  The product binds that interface to libtgcn_hip.so and nothing else; the gloo tests on the CPU
(tests/test_dist_gloo.py, tests/test_sharded_modules.py) and `bench.py --rehearsal-cpu` inject THIS object instead, so that the
world-size > 1 control flow can be checked against the oracle where there is no GPU.  Nothing measured with it is a result.
"""
import numpy as np
import torch


class ScipyOperand:
    def __init__(self, n_rows, n_cols, row, col, val):
        import scipy.sparse as sp
        self.n, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(row.numel())
        self.L = sp.coo_matrix((val.cpu().numpy().astype(np.float32), (row.cpu().numpy(), col.cpu().numpy())), shape=(n_rows, n_cols)).tocsr()


class CpuOps:
    name = "cpu stand-in (scipy / numpy): control flow only"

    def operand(self, n_rows, n_cols, row, col, val, device):
        return ScipyOperand(n_rows, n_cols, row, col, val)

    def edge_coo(self, edge_index, edge_weight, n, device):
        """the edge-list classes' operand in numpy (tgcn/nn/gcn.py:398-413): self loops removed, unweighted source degree, inf -> 0"""
        row, col = edge_index[0].numpy(), edge_index[1].numpy()
        keep = row != col
        row, col = row[keep], col[keep]
        w = np.ones(row.shape[0], np.float32) if edge_weight is None else edge_weight.detach().numpy().astype(np.float32).reshape(-1)[keep]
        deg = np.bincount(row, minlength=n).astype(np.float32)
        with np.errstate(divide="ignore"):
            dis = deg ** np.float32(-0.5)
        dis[np.isinf(dis)] = 0
        return torch.from_numpy(row.astype(np.int64)), torch.from_numpy(col.astype(np.int64)), torch.from_numpy((-dis[row] * w * dis[col]).astype(np.float32))

    def hop(self, op, x, z, alpha, beta, out, z2=None, gamma=0.0):
        y = np.stack([op.L.dot(x[b].numpy()) for b in range(x.shape[0])]).astype(np.float32)
        y = np.float32(alpha) * y
        if z is not None:
            y = y + np.float32(beta) * z.numpy()
        if z2 is not None:
            y = y + np.float32(gamma) * z2.numpy()
        out.copy_(torch.from_numpy(np.ascontiguousarray(y.astype(np.float32))))
        return out

    def project(self, terms, W, bias, bias_kind, n_vertices, rowmap=None, out=None):
        """out[r(m)] = sum_t terms[t][m] @ W[t] + bias[r(m)]; r = rowmap (int32) or the identity"""
        acc = sum(t.numpy().astype(np.float64) @ W[k].numpy().astype(np.float64) for k, t in enumerate(terms))
        M, N = acc.shape
        r = np.arange(M) if rowmap is None else rowmap.numpy().astype(np.int64)
        if bias_kind == 1:
            acc = acc + bias.numpy().reshape(1, N)
        elif bias_kind == 2:
            acc = acc + bias.numpy().reshape(-1, N)[r % n_vertices]
        res = np.empty((M, N), np.float32)
        res[r] = acc.astype(np.float32)
        res = torch.from_numpy(res)
        return res if out is None else out.copy_(res.reshape(out.shape))

    def project_first(self, x3, Wcat, bias, bias_kind, K, N, rowmap=None):
        q, rows, C = x3.shape
        Z = (x3.numpy().astype(np.float64).reshape(q * rows, C) @ Wcat.numpy().astype(np.float64)).reshape(q, rows, K * N)
        res = np.empty_like(Z)
        r = np.arange(rows) if rowmap is None else rowmap.numpy().astype(np.int64)
        res[:, r] = Z
        if bias_kind == 1:
            res[:, :, :N] += bias.numpy().reshape(1, 1, N)
        elif bias_kind == 2:
            res[:, :, :N] += bias.numpy().reshape(1, rows, N)
        return torch.from_numpy(res.astype(np.float32))

    def pack(self, src, idx, out):
        return out.copy_(src.index_select(0, idx))

    def wgrad(self, terms, g2d):
        g = g2d.numpy().astype(np.float64)
        return torch.from_numpy(np.stack([t.numpy().astype(np.float64).T @ g for t in terms]).astype(np.float32))

    def fold(self, W, transpose=False):
        K = W.shape[0]
        if K <= 2:
            return W
        from tgcn_amd.functional import _power_fold_matrix
        c = _power_fold_matrix(K, dtype=torch.float64).numpy()
        m = c if transpose else c.T                       # W'[j] = sum_k c[k, j] W[k]; transpose: sum_k c[j, k] W[k]
        return torch.from_numpy(np.einsum("jk,kcn->jcn", m, W.numpy().astype(np.float64)).astype(np.float32))

    def weight_layout(self, W, kind):
        K, C, N = W.shape
        if kind == 0:
            return W.permute(1, 0, 2).reshape(C, K * N).contiguous()
        if kind == 1:
            return W.permute(0, 2, 1).contiguous()
        return W.permute(2, 0, 1).reshape(N, K * C).contiguous()
