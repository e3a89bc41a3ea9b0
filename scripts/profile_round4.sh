# round 4: the profile set of ONE library build, every file tagged with its build id (pe_hip_build_id).
#   part a (one gpurun call):  bash scripts/profile_round4.sh a <tag>   GPU suite, smoke, default bench
#   part b (one gpurun call):  bash scripts/profile_round4.sh b <tag>   rocprofv3 kernel trace + the two separate PMC traffic passes
#   part c (one gpurun call):  bash scripts/profile_round4.sh c <tag>   the three SQ counter passes (scripts/pmc_sq.sh)
# rocprofv3 always gets the program itself after `--` (python3 bench.py ...), never a wrapper.
PART=${1:-a}
TAG=${2:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
ID=$(python3 -c "import pe_load; print(pe_load.load().ffi.build_id())")
echo "build_id $ID" | tee $O/${TAG}_build_id.txt
if [ "$PART" = a ]; then
  timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -2 $O/${TAG}_gpu_tests.log
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout -k 10 500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err && tail -c 1500 $O/${TAG}_bench.json
elif [ "$PART" = b ]; then
  cd /tmp && export TMPDIR=/tmp &&
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_trace.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_fetch.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_write.log 2>&1
  echo done; f=$(ls -t $O/${TAG}_trace/*/*kernel_stats.csv | head -1); (echo "# build_id $ID  (rocprofv3 --kernel-trace --stats of: python3 bench.py --no-cpu-baseline --no-single)"; cat $f) > $O/${TAG}_kernel_stats.csv; cut -c1-150 $O/${TAG}_kernel_stats.csv | head -14
  tail -1 $O/${TAG}_trace.log > $O/${TAG}_trace_bench.json
else
  TAG=$TAG bash scripts/pmc_sq.sh > $O/${TAG}_pmc_sq_run.log 2>&1; tail -5 $O/${TAG}_pmc_sq_run.log
fi
