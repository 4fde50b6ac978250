#!/bin/bash
# PMC passes for one kernel of tools/kernel_bench.py (separate --pmc runs, kernel-trace only; MI355X_MICROARCH.md):
#   bash tools/pmc_kernel.sh <kernel_bench --only selector> <out tag>
SEL=${1:-sdpa}
TAG=${2:-$SEL}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 4 --only $SEL"
run() { timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $P -- $CMD > $OUT/$P.log 2>&1; }
P=p1 run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
P=p2 run SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
P=p3 run GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32
P=p4 run FETCH_SIZE
P=p5 run WRITE_SIZE
ls $OUT
