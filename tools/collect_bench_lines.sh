#!/bin/bash
# Runs bench.py for every workload on the local GPU and writes the JSON lines to gpurun_out/bench_lines/<workload>.json
# (tools/merge_bench_lines.py folds them into profiles/<tag>_bench_lines.json).  Developer tool.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/bench_lines
timeout -k 10 900 python3 bench.py > gpurun_out/bench_lines/cfg5.json 2> gpurun_out/bench_lines/cfg5.err
timeout -k 10 600 python3 bench.py --labeling degree --no-cpu > gpurun_out/bench_lines/cfg5_degree.json 2> /dev/null
timeout -k 10 600 python3 bench.py --workload cfg5n > gpurun_out/bench_lines/cfg5n.json 2> /dev/null
for w in cfg4 cfg3 cfg2 hcp148; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 100 --warmup 20 > gpurun_out/bench_lines/$w.json 2> /dev/null
done
timeout -k 10 300 python3 tools/train_bench.py > gpurun_out/bench_lines/train_dx.txt 2> /dev/null
timeout -k 10 300 python3 tools/train_bench.py --no-dx > gpurun_out/bench_lines/train_nodx.txt 2> /dev/null
echo done
