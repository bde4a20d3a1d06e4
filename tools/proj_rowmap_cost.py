#!/usr/bin/env python3
"""The Z projection of the project-first form (cfg4: 90 k x 1200 x 160) with and without an OUTPUT row map: what the vertex shards' [interior |
boundary] row order costs the projection (tgcn_cheb_project_first_f32; developer tool)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tgcn_amd import functional as F, _lib
dev = torch.device("cuda:0")
for (rows, C, K, N) in ((90000, 1200, 5, 32), (45000, 1200, 5, 32), (90000, 64, 5, 8)):
    x = torch.randn(1, rows, C, device=dev)
    Wcat = torch.randn(C, K * N, device=dev) / 30
    bias = torch.randn(rows, N, device=dev)
    ident = torch.arange(rows, device=dev, dtype=torch.int32)
    perm = torch.randperm(rows, device=dev).to(torch.int32)
    half = torch.cat([torch.arange(0, rows, 2, device=dev), torch.arange(1, rows, 2, device=dev)]).to(torch.int32)
    res = {}
    for name, rm in (("no_map", None), ("identity_map", ident), ("interleaved_map", half), ("random_map", perm)):
        for _ in range(5): F.project_first(x, Wcat, bias, 2, K, N, rowmap=rm)
        torch.cuda.synchronize()
        _lib.profile_start(256)
        for _ in range(30): F.project_first(x, Wcat, bias, 2, K, N, rowmap=rm)
        torch.cuda.synchronize()
        pr = [ms for k, ms in _lib.profile_stop(256) if k == 2]
        res[name] = round(sum(pr) / len(pr), 4)
    print(json.dumps(dict(shape="%d x %d x %d" % (rows, C, K * N), projection_ms=res)), flush=True)
