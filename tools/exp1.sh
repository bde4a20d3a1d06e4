mkdir -p gpurun_out/r02/exp1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for hp in 0 1 8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/exp1/f_hp$hp -- python3 $R/tools/hop_bench.py --variants 0 --sweep 1 --rounds 3 --sweep-hot-panels $hp > $R/gpurun_out/r02/exp1/hp$hp.log 2>&1
  tail -1 $R/gpurun_out/r02/exp1/hp$hp.log
  (cd $R && python3 tools/traffic_json.py gpurun_out/r02/exp1/t_hp$hp.json gpurun_out/r02/exp1/f_hp$hp | grep -A3 hop_sweep | grep fetch)
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/exp1/f_deg -- python3 $R/tools/hop_bench.py --variants 0 --sweep 0,1 --rounds 3 --labeling degree > $R/gpurun_out/r02/exp1/deg.log 2>&1
tail -2 $R/gpurun_out/r02/exp1/deg.log
(cd $R && python3 tools/traffic_json.py gpurun_out/r02/exp1/t_deg.json gpurun_out/r02/exp1/f_deg | grep -B1 -A3 '"hop_' | grep -E 'hop_|fetch')
