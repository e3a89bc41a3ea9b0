# Runs on the GPU box: on-box HBM ceiling and FETCH_SIZE / WRITE_SIZE calibration (scripts/hbm_calib.hip) -> gpurun_out/calib_*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
BIN=$R/scripts/_build/hbm_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $R/scripts/hbm_calib.hip
timeout -k 10 120 $BIN 4096 5 > $O/calib_bare.jsonl 2>&1 && cat $O/calib_bare.jsonl &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $BIN 4096 2 > $O/calib_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $BIN 4096 2 > $O/calib_write.log 2>&1
echo calib done
