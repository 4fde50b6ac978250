#!/bin/bash
# PMC passes (3 SQ groups, separate runs, kernel-trace only) of as_sdpa_fwd under AS_SDPA_IMPL=$1 -> gpurun_out/pmc_sdpa_impl$1
IMPL=${1:-auto}
cd /tmp && export TMPDIR=/tmp
if [ "$IMPL" != "auto" ]; then export AS_SDPA_IMPL=$IMPL; fi
TAG=${PMC_TAG:-impl$IMPL}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sdpa_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD=${PMC_CMD:-"python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 4 --only sdpa ${KB_ARGS}"}
run() { timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $P -- $CMD > $OUT/$P.log 2>&1; }
P=p1 run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
P=p2 run SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
P=p3 run GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32
if [ -n "$PMC_TRAFFIC" ]; then P=p4 run FETCH_SIZE; P=p5 run WRITE_SIZE; fi
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/pmc_sdpa_$TAG ${PMC_FILTER:-sdpa_fwd} gpurun_out/pmc_sdpa_$TAG.md "as_sdpa_fwd AS_SDPA_IMPL=$IMPL ($TAG)" | tail -40
rm -rf $OUT/*.db $OUT/*/*.db 2>/dev/null
