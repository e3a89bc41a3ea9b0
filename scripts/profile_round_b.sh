# part B of scripts/profile_round.sh: rocprofv3 kernel trace of the bench + the two separate PMC traffic passes -> gpurun_out/<tag>_*
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --no-cpu-baseline --no-single > $O/${TAG}_write.log 2>&1
echo done; cut -c1-150 $O/${TAG}_trace/*/*kernel_stats.csv | head -12
