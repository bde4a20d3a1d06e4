mkdir -p gpurun_out/r02/exp2
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_sweep.py -x -q 2>&1 | tail -4
python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20
cd /tmp && export TMPDIR=/tmp
for cfg in "16 0" "16 16" "64 16" "64 48" "32 32"; do
  set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/exp2/f_h$1_b$2 -- python3 $R/tools/hop_bench.py --variants 0 --sweep 1 --rounds 3 --sweep-hot-panels $1 --sweep-barriers $2 > $R/gpurun_out/r02/exp2/h$1_b$2.log 2>&1
  echo "hot $1 barriers $2: $(tail -1 $R/gpurun_out/r02/exp2/h$1_b$2.log)"
  (cd $R && python3 tools/traffic_json.py gpurun_out/r02/exp2/t_h$1_b$2.json gpurun_out/r02/exp2/f_h$1_b$2 | grep -A3 hop_sweep | grep fetch)
done
