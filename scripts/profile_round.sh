# Runs on the GPU box (gpurun): GPU tests + smoke + the default bench bare, under rocprofv3 kernel trace, and the two PMC
# traffic passes.  Usage: bash scripts/profile_round.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -2 $O/${TAG}_gpu_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err && tail -c 1500 $O/${TAG}_bench.json &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_trace.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_write.log 2>&1
echo done
