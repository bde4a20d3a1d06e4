#!/bin/bash
# cfg4 (sheet_mesh(300), TGCNCheb_H(L,1,32,5,1200), q = 1) three ways on ONE GPU: the single-GPU driver, the
# vertex-sharded module at world 1 under RCCL (no halo: what the sharded control flow itself costs), and at world 2 with both ranks on the
# card and the gloo transport (host copies + a device drain per exchange: a rehearsal of the exchange, not a scaling number).
# usage: tools/collect_cfg4_sharded.sh <tag>   -> gpurun_out/<tag>_cfg4_{single,vertex_w1_nccl,vertex_w2_gloo}.json
set -u
tag=${1:-r06}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
timeout -k 10 200 python bench.py --workload cfg4 --steps 100 --warmup 20 --no-cpu > gpurun_out/${tag}_cfg4_single.json 2> gpurun_out/${tag}_cfg4_single.err || echo "single failed"
port=$((20000 + RANDOM % 20000))
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --workload cfg4 --shard vertex \
  --backend nccl --steps 100 --warmup 20 --no-cpu > gpurun_out/${tag}_cfg4_vertex_w1_nccl.json 2> gpurun_out/${tag}_cfg4_vertex_w1_nccl.err || echo "w1 failed"
port=$((20000 + RANDOM % 20000))
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --workload cfg4 --shard vertex \
  --backend gloo --steps 50 --warmup 10 --no-cpu > gpurun_out/${tag}_cfg4_vertex_w2_gloo.json 2> gpurun_out/${tag}_cfg4_vertex_w2_gloo.err || echo "w2 failed"
for f in single vertex_w1_nccl vertex_w2_gloo; do
  python - "$tag" "$f" <<'PY'
import json, sys
tag, f = sys.argv[1], sys.argv[2]
try:
    d = [json.loads(l) for l in open("gpurun_out/%s_cfg4_%s.json" % (tag, f)) if l.startswith("{")][-1]
    r = d.get("roofline") or {}
    print(f, "ms_per_step", d["ms_per_step"], "value", d["value"], "sharding:", d["config"]["sharding"], "| hop launches/step", r.get("launches_per_step"), "mean hop ms", r.get("mean_launch_ms"), "proj ms", r.get("project_ms_per_step"))
except Exception as e:
    print(f, "no line:", e)
PY
done
