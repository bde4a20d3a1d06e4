mkdir -p gpurun_out/r02/exp3
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_sweep.py -x -q 2>&1 | tail -3
for cfg in "64 16" "64 32" "64 48" "64 64"; do
  set -- $cfg
  python3 tools/hop_bench.py --variants 0 --sweep 1 --rounds 5 --sweep-loads 8,16 --sweep-hot-panels $1 --sweep-barriers $2 > gpurun_out/r02/exp3/h$1_b$2.log 2>&1
  echo "hot $1 barriers $2:"; grep hop_variant gpurun_out/r02/exp3/h$1_b$2.log | cut -c1-120
done
python3 tools/hop_bench.py --variants 0 --sweep 1 --rounds 5 --sweep-hot-panels 64 --sweep-barriers 48 --sweep-thresh 8 > gpurun_out/r02/exp3/t8.log 2>&1
echo "thresh 8:"; grep -E "hop_variant|sweep schedule" gpurun_out/r02/exp3/t8.log | cut -c1-130
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/exp3/f -- python3 $R/tools/hop_bench.py --variants 0 --sweep 1 --rounds 3 --sweep-hot-panels 64 --sweep-barriers 48 > /dev/null 2>&1
(cd $R && python3 tools/traffic_json.py gpurun_out/r02/exp3/t.json gpurun_out/r02/exp3/f | grep -A3 hop_sweep | grep fetch)
