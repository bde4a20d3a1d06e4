#!/usr/bin/env python3
"""relu + pool inside the projection epilogue vs layer + separate pass, for the WRITE_SIZE counter (developer tool):

    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pool_pmc -- python3 tools/pool_pmc.py

Runs cheb_relu_pool(GCNCheb(L, 32, 64, 5), x, pool=4) on the 59,536-vertex sheet mesh with q = 8 (fused epilogue:
tgcn_cheb_forward_pool_f32), then the same layer followed by tgcn_relu_pool_f32.  Expected bytes written by the projection
kernel: fused q*n/4*64*4 = 30.5 MB, unfused q*n*64*4 = 121.9 MB (+ 30.5 MB by the pool pass)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def main():
    import tgcn_amd
    from tgcn_amd import functional as F, _lib
    from tools import synth
    n, row, col, val = synth.sheet_mesh(244)
    op = tgcn_amd.GraphOperand.from_coo(n, row, col, val)
    torch.manual_seed(0)
    layer = tgcn_amd.GCNCheb(op, 32, 64, 5).cuda()
    q, pool = 8, 4
    x = torch.randn(q, n, 32, device="cuda")
    assert F.pool_epilogue_is_fused(op, q, 32, 64, 5, pool)
    with torch.no_grad():
        for _ in range(3):
            z = tgcn_amd.cheb_relu_pool(layer, x, pool=pool)
        torch.cuda.synchronize()
        for _ in range(3):
            y = layer(x)
            z2 = torch.empty_like(z)
            idx = torch.empty(z.shape, dtype=torch.uint8, device="cuda")
            _lib.check(_lib.lib().tgcn_relu_pool_f32(_lib.stream_ptr(), _lib.ptr(y), _lib.ptr(z2), _lib.ptr(idx), q, n, 64, pool))
        torch.cuda.synchronize()
    assert torch.equal(z, z2)
    print("fused output %d bytes, layer output %d bytes" % (z.numel() * 4, y.numel() * 4))


if __name__ == "__main__":
    main()
