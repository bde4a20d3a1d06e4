#!/bin/bash
# Everything a round's record needs from ONE gpurun call, final binary: the default bench line, the bench lines of every workload, the
# kernel-trace stats + --pmc traffic of the default bench, the limiter counters of the projection and the TCC hit rate of the
# nt-on-cold-gathers experiment.  Results under gpurun_out/<tag>/ (copy what is to be judged into profiles/).  usage: tools/collect_round.sh r05
tag=${1:-r05}
cd "$(dirname "$0")/.."
out=gpurun_out/$tag
mkdir -p $out
echo "== default bench" | tee -a $out/steps.log
timeout -k 10 900 python3 bench.py > $out/bench_cfg5.json 2> $out/bench_cfg5.err || echo "bench rc=$?" | tee -a $out/steps.log
echo "== bench lines" | tee -a $out/steps.log
timeout -k 10 1500 bash tools/collect_bench_lines.sh > $out/bench_lines.log 2>&1 || echo "bench lines rc=$?" | tee -a $out/steps.log
python3 tools/merge_bench_lines.py $tag > $out/merge.log 2>&1
echo "== traffic" | tee -a $out/steps.log
timeout -k 10 1500 bash tools/collect_traffic.sh $tag > $out/traffic.log 2>&1 || echo "traffic rc=$?" | tee -a $out/steps.log
echo "== projection limiter" | tee -a $out/steps.log
timeout -k 10 1500 python3 tools/pmc_limiter.py ${tag}_proj --labelings random --script bench.py --extra "--steps 1 --warmup 1 --no-cpu --no-ceiling --no-others" \
  --kernels hop_fixup_kernel,hop_kernel,project_x3_stream_kernel --only-blocks TCP,TCC,SQ,GRBM --limit 240 > $out/proj_limiter.log 2>&1 || echo "proj limiter rc=$?" | tee -a $out/steps.log
echo "== cold-nt TCC" | tee -a $out/steps.log
for v in 6 5; do
  timeout -k 10 600 python3 tools/pmc_limiter.py ${tag}_coldnt_v$v --labelings random --only-blocks TCC --max-passes 1 --limit 240 \
    --extra "--seg-modes 0 --variants $v --cold-last 524288 --cold-nt" > $out/coldnt_v$v.log 2>&1 || echo "coldnt v$v rc=$?" | tee -a $out/steps.log
done
echo "== done" | tee -a $out/steps.log
