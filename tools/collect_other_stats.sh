#!/bin/bash
# rocprofv3 kernel-trace stats of the other BASELINE workloads (cfg2 f = 1 / 64, cfg3, cfg4): the summaries that `other_workloads[*].roofline` /
# `kernel_ms_per_step` of the default bench line must agree with.  usage: tools/collect_other_stats.sh <tag>  -> gpurun_out/<tag>/<tag>_<w>_kernel_stats.csv
set -u
tag=${1:-r06}
cd "$(dirname "$0")/.."
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg2w cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -- python3 $root/bench.py --workload $w --steps 100 --warmup 20 --no-cpu > $out/kt_$w.log 2>&1
  f=$(find $out/kt_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_${w}_kernel_stats.csv && head -3 $out/${tag}_${w}_kernel_stats.csv | cut -c1-160
done
