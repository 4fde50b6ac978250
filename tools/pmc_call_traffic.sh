#!/bin/bash
# HBM traffic of ONE as_cosine_shift call (all its launches): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes (kernel-trace only, MI355X_MICROARCH.md), summarised on the box into gpurun_out/<tag>.md / .json
TAG=${1:-r02_shift_traffic}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out
CMD="python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 6 --only shift"
export AS_KB_NO_FULLBOXES=1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o p4 -- $CMD > $OUT/p4.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o p5 -- $CMD > $OUT/p5.log 2>&1
grep -v "rocprofv3\]" $OUT/p4.log | tail -15; ls $OUT
python $GRAFT_REPO_ROOT/tools/pmc_call_traffic.py $OUT shift_final_sim_kernel "shift_" $GRAFT_REPO_ROOT/gpurun_out/$TAG
