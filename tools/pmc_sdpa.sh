#!/bin/bash
# PMC passes for the SDPA kernel (separate --pmc runs, kernel-trace only; see MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sdpa
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 4 --only sdpa"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS_F32 --kernel-trace -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
ls $OUT
