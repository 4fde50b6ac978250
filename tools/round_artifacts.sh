#!/bin/bash
# GPU-box: the measurements a round's profiles/ entries come from, in one call.  usage: tools/round_artifacts.sh r03
# Writes gpurun_out/<R>_*: full default bench line (+ wall time), rocprofv3 kernel stats of the headline leg, kernel_bench,
# PMC summaries of the shift / roll-out / GEMM kernels, the per-call HBM traffic of as_cosine_shift, and the training
# step's kernel stats, phase timing and host-sync list.
R=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.log
echo "default bench.py wall: $(( $(date +%s) - t0 )) s" | tee gpurun_out/${R}_bench_wall.txt
AS_BENCH_EVENTS=0 AS_BENCH_OTHER_RNG=0 AS_BENCH_MIL=0 AS_BENCH_FP32=0 PROF_LINES=8 tools/prof_cmd.sh ${R}_bench_kernel_stats_final python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --other-configs "" > /dev/null 2>&1
timeout 300 python tools/kernel_bench.py --reps 20 > gpurun_out/${R}_kernel_bench.jsonl 2>&1
bash tools/pmc_call_traffic.sh ${R}_shift_traffic > gpurun_out/${R}_shift_traffic.log 2>&1
for k in shift_sim shift_assign shift_aggregate shift_final_sim; do
  PMC_TAG=${R}_$k PMC_FILTER=${k}_kernel PMC_CMD="python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 6 --only shift" bash tools/pmc_sdpa_impl.sh auto > /dev/null 2>&1
done
PMC_TAG=${R}_rollout_step4 PMC_FILTER=rollout_step4 PMC_CMD="python $GRAFT_REPO_ROOT/tools/kernel_bench.py --reps 8 --only rollout" bash tools/pmc_sdpa_impl.sh auto > /dev/null 2>&1
PMC_TRAFFIC=1 PMC_TAG=${R}_gemm_fc1 PMC_FILTER=gemm_pp PMC_CMD="python $GRAFT_REPO_ROOT/tools/experiments/gemm_variant_bench.py _one --variants base --shapes 8394x3072x768" bash tools/pmc_sdpa_impl.sh auto > /dev/null 2>&1
PMC_TRAFFIC=1 PMC_TAG=${R}_gemm_fc2 PMC_FILTER=gemm_pp PMC_CMD="python $GRAFT_REPO_ROOT/tools/experiments/gemm_variant_bench.py _one --variants base --shapes 8394x768x3072" bash tools/pmc_sdpa_impl.sh auto > /dev/null 2>&1
PROF_LINES=8 tools/prof_cmd.sh ${R}_train_step_kernel_stats python $GRAFT_REPO_ROOT/tools/experiments/train_steps.py 6 > /dev/null 2>&1
timeout 300 python tools/experiments/train_phases.py 8 2>&1 | tail -9 > gpurun_out/${R}_train_phases.txt
timeout 300 python tools/experiments/train_syncs.py 2>&1 | tail -8 > gpurun_out/${R}_train_syncs.txt
# round 4: SDPA counters incl. HBM traffic of the shipped kernel, the dependent-chain floor, the power / clock probe
PMC_TRAFFIC=1 PMC_TAG=${R}_sdpa bash tools/pmc_sdpa_impl.sh auto > gpurun_out/${R}_sdpa_pmc.log 2>&1
cp gpurun_out/pmc_sdpa_${R}_sdpa.md gpurun_out/${R}_sdpa_pmc.md 2>/dev/null
( hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_floor tools/experiments/chain_floor.hip > /dev/null 2>&1 && /tmp/chain_floor ) > gpurun_out/${R}_chain_floor.json 2>&1
timeout 120 python tools/experiments/power_probe.py 2>&1 | grep kernel > gpurun_out/${R}_power_probe.jsonl
# round 5: in-kernel timeline of the mean-shift kernels (instrumented build under tools/experiments/_build/), the glue census,
# and -- when a round-4 tree has been unpacked into _r04/ (git archive b73500e | tar -x -C _r04; built there) -- the
# same-box A/B of the headline leg against it
timeout 200 python tools/experiments/shift_timeline.py run --md gpurun_out/${R}_shift_timeline.md > gpurun_out/${R}_shift_timeline.log 2>&1
timeout 300 python tools/experiments/glue_sites.py --steps 3 --rows 200 > gpurun_out/${R}_glue_sites.log 2>&1
[ -d _r04 ] && bash tools/experiments/r05_ab.sh > gpurun_out/${R}_ab_vs_r04.log 2>&1
timeout 300 python tools/experiments/step_trace.py > gpurun_out/${R}_step_trace.log 2>&1 && cp gpurun_out/step_timeline.txt gpurun_out/${R}_step_timeline.txt
ls gpurun_out | grep ${R}_
cat gpurun_out/${R}_bench_wall.txt
