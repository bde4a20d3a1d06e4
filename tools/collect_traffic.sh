#!/bin/bash
# rocprofv3 kernel-trace stats + three --pmc passes of the default bench (cfg5, random labels) on the local GPU; results
# under gpurun_out/<tag>/, summaries copied to profiles/<tag>_*.  Developer tool: tools/collect_traffic.sh r02
# (--no-ceiling: the gather-ceiling launches of the bench line run hop_kernel on a folded operand and would enter the per-kernel means)
set -e
tag=${1:-r02}
cd "$(dirname "$0")/.."
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-ceiling --no-others > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu --no-ceiling --no-others > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu --no-ceiling --no-others > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu --no-ceiling --no-others > $out/pmc_tcc.log 2>&1
cd $root
python3 tools/traffic_json.py $out/traffic_cfg5_random.json $out/pmc_fetch $out/pmc_write $out/pmc_tcc
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/${tag}_cfg5_random_kernel_stats.csv
echo done
