"""The boundary follows the DATA's device, like the reference's torch ops (tgcn/nn/gcn.py:141,147; nn.DataParallel in
examples/pytorch_based/pytorch_hcp_tgcn.py:270-273): every functional entry makes the device of its arguments current before it takes
that device's stream and calls the library; arguments on two devices are refused; the C ABI checks the pointer's device against
hipGetDevice.  The two-device cases need two GPUs and skip on the one-GPU box; the argument check runs on the CPU."""
import types

import numpy as np
import pytest
import torch

import tgcn_amd
from tgcn_amd import _lib, functional as F


def test_arguments_on_two_devices_are_refused_before_any_launch():
    a = types.SimpleNamespace(device=torch.device("cuda", 0))          # stands for a GraphOperand on cuda:0
    b = types.SimpleNamespace(device=torch.device("cuda", 1))
    called = []

    @F._on_device
    def entry(op, plan):
        called.append(1)

    with pytest.raises(_lib.TgcnError, match="two devices"):
        entry(a, b)
    assert not called
    # every entry that takes a stream and calls the library carries the guard
    for name in ("csr_hop", "cheb_project", "cheb_forward_raw", "cheb_forward_compact", "cheb_forward_small", "cheb_forward_pf", "fold_weight",
                 "cheb_wgrad", "csr_sddmm", "pack_rows", "relayout_qnc_to_nqc", "project_mapped", "cheb_forward_pool", "cheb_basis_small"):
        assert getattr(F, name).__wrapped__ is not None, name
    for fn in (F.ChebLayerFn, F.ChebReluPoolFn, F.ReluPoolFn, F.PoolMaxFn, F.SpmmFn, F.ChebWindowsFn):
        assert hasattr(fn.forward, "__wrapped__") and hasattr(fn.backward, "__wrapped__"), fn


def test_cpu_tensors_still_fail_loudly():
    x = torch.zeros(1, 4, 4)
    with pytest.raises(_lib.TgcnError, match="no CPU fallback"):
        F.relayout_qnc_to_nqc(torch.zeros(2, 4, 4))
    del x


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def _layer_and_input(dev, seed=0):
    rng = np.random.default_rng(seed)
    n, q, f, g, K = 3000, 3, 8, 16, 4
    row, col = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
    val = (rng.standard_normal(row.shape[0]) / 3).astype(np.float32)
    import scipy.sparse as sp
    L = sp.coo_matrix((val, (row, col)), shape=(n, n)).tocsr()
    torch.manual_seed(1)
    layer = tgcn_amd.TGCNCheb(L, f, g, K).to(dev)
    x = torch.as_tensor(rng.standard_normal((q, n, f)).astype(np.float32)).to(dev)
    return layer, x


@pytest.mark.gpu
@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_module_on_cuda1_while_cuda0_is_current():
    l0, x0 = _layer_and_input(torch.device("cuda:0"))
    l1, x1 = _layer_and_input(torch.device("cuda:1"))
    torch.cuda.set_device(0)
    x1.requires_grad_(True)
    x0.requires_grad_(True)
    y0, y1 = l0(x0), l1(x1)                        # current device 0 for both calls
    assert y1.device == torch.device("cuda:1") and torch.cuda.current_device() == 0
    assert torch.equal(y0.cpu(), y1.cpu())
    y0.square().sum().backward()
    y1.square().sum().backward()
    assert torch.equal(x0.grad.cpu(), x1.grad.cpu()) and torch.equal(l0.weight.grad.cpu(), l1.weight.grad.cpu())
    # the bare C ABI refuses a pointer of the other device
    op = l1._operand(torch.device("cuda:1"))
    with pytest.raises(_lib.TgcnError, match="current device"):
        F.csr_hop.__wrapped__(op, x1.detach())


@pytest.mark.gpu
@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_data_parallel_over_two_devices_equals_one_device():
    layer, x = _layer_and_input(torch.device("cuda:0"))
    x = torch.cat([x, x.flip(0)])                 # 6 samples: 3 per replica
    want = layer(x)
    got = torch.nn.DataParallel(layer, device_ids=[0, 1])(x)
    assert torch.equal(want, got)
