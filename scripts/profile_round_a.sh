# part A of scripts/profile_round.sh (a gpurun call is limited to 20 minutes): GPU tests, smoke, the default bench -> gpurun_out/<tag>_*
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -2 $O/${TAG}_gpu_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err && tail -c 2500 $O/${TAG}_bench.json
