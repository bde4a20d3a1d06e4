#!/usr/bin/env python3
"""Partition, exchange volumes and construction time of the vertex-sharded layer on a BASELINE workload, at any world size, without the GPUs
of a whole node: W gloo ranks on this machine, each on cuda:0 (HIP operand builders) or on the CPU (--cpu: scipy stand-ins).

    python tools/shard_stats.py --workload cfg4 --world 8 --cpu            # rows / bytes per peer and hop at world 8 (DESIGN.md section 7's table)
    python tools/shard_stats.py --workload cfg5 --world 2                  # constructor + transpose() seconds at cfg5 scale, two ranks on one GPU

One JSON line: per rank what VertexShardedCheb.describe(width) says, the seconds its constructor and its transpose() took (tensor collectives
only: count exchange + all_to_all_single of id / entry tensors), and the largest message.  Developer tool."""
import argparse
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, args, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tgcn_amd.dist import VertexShardedCheb
        from tools import synth
        cpu = args.cpu
        dev = torch.device("cpu") if cpu else torch.device("cuda:0")
        if not cpu:
            torch.cuda.set_device(0)
        ops = None
        if cpu:
            from tools.cpu_standins import CpuOps
            ops = CpuOps()
        if args.workload == "cfg4":
            n, row, col, val = synth.sheet_mesh(300, device=dev)
            width = 32
        else:
            n, row, col, val = synth.rmat(args.vertices, args.entries, seed=12345, labeling="random", device=dev)
            width = 64
        sync = (lambda: None) if cpu else torch.cuda.synchronize
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        sh = VertexShardedCheb(n, row, col, val, device=dev, exchange=args.exchange, ops=ops)
        sync()
        dist.barrier()
        t1 = time.perf_counter()
        del row, col, val
        T = sh.transpose()
        sync()
        dist.barrier()
        t2 = time.perf_counter()
        d = sh.describe(width)
        d.update(constructor_s=round(t1 - t0, 3), transpose_s=round(t2 - t1, 3), transpose_exchange=T.exchange, transpose_halo_rows=T.halo)
        ret[rank] = d
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4", choices=["cfg4", "cfg5"])
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--exchange", default="auto")
    ap.add_argument("--vertices", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=160_000_000)
    args = ap.parse_args()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(args.world, port, args, ret), nprocs=args.world, join=True)
    ranks = [ret[r] for r in range(args.world)]
    line = dict(workload=args.workload, world=args.world, transport="gloo", compute="scipy stand-ins on the CPU" if args.cpu else "HIP operand builders, all ranks on cuda:0",
                vertices=(90000 if args.workload == "cfg4" else args.vertices), exchange=ranks[0]["exchange"], row_floats=ranks[0]["row_floats"],
                constructor_s_max=max(r["constructor_s"] for r in ranks), transpose_s_max=max(r["transpose_s"] for r in ranks),
                halo_rows_max=max(r["halo_rows"] for r in ranks), bytes_in_per_hop_and_time_step_max=max(r["bytes_in_per_hop_and_time_step"] for r in ranks),
                largest_message_bytes=max((max(r["message_bytes_per_peer_in"]) if r.get("message_bytes_per_peer_in") else r["bytes_in_per_hop_and_time_step"]) for r in ranks),
                ranks=ranks)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
