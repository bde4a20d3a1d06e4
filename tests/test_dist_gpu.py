"""N = 2 on ONE GPU (gloo rendezvous, both ranks on cuda:0): the vertex-sharded layer with its real HIP compute
(the CPU suite runs the same communication code with the oracle injected, tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cheb_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, exchange, banded, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        torch.cuda.set_device(0)
        rng = np.random.default_rng(3)
        n, q, C, N, K = 3000, 2, 8, 12, 4
        if banded:
            row = np.repeat(np.arange(n), 6)
            col = np.clip(row + rng.integers(-9, 10, row.shape[0]), 0, n - 1)
        else:
            row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
        row = np.concatenate([row, np.full(500, 7)])
        col = np.concatenate([col, rng.integers(0, n, 500)])
        val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
        x = rng.standard_normal((q, n, C)).astype(np.float32)
        W = (rng.standard_normal((K, C, N)) / 4).astype(np.float32)
        bias = rng.standard_normal((n, N)).astype(np.float32)
        dev = torch.device("cuda:0")
        sh = VertexShardedCheb(n, torch.as_tensor(row), torch.as_tensor(col), torch.as_tensor(val), device=dev, exchange=exchange)
        args = (torch.as_tensor(x[:, sh.lo:sh.hi]).to(dev), torch.as_tensor(W).to(dev), torch.as_tensor(bias[sh.lo:sh.hi]).to(dev), 2, 1)
        out = sh.forward(*args)                                   # overlapped: HIP pack kernel, in-place receives, interior / boundary hops
        assert torch.equal(out, sh.forward(*args, overlap=False)), "overlapped and plain forms differ"
        if exchange == "halo":
            assert 0 < sh.n_int < sh.owned
        L = O.coo_to_csr(row, col, val, n)
        ref = np.einsum("kqnc,kcg->qng", O.stack_chebyshev(L, x, K).astype(np.float64), W.astype(np.float64)) + bias
        err = np.abs(out.cpu().numpy() - ref[:, sh.lo:sh.hi]).max() / np.abs(ref).max()
        ret[rank] = (float(err), sh.exchange, sh.owned)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,banded", [("halo", True), ("allgather", False)])
def test_vertex_sharded_hip_two_ranks_one_gpu(exchange, banded, gpu_device):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), exchange, banded, ret), nprocs=world, join=True)
    assert len(ret) == world and sum(ret[r][2] for r in range(world)) == 3000
    for r in range(world):
        assert ret[r][0] <= 1e-5 and ret[r][1] == exchange, ret[r]
