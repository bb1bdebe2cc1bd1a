#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG [stats|traffic|sq ...]
#   stats    rocprofv3 --kernel-trace --stats of bench.py --single-stream (3 warm-up + 5 timed steps) -> gpurun_out/TAG_kernel_stats.md
#   stats_default  the same of the default command (recorded step, side streams) -> gpurun_out/TAG_kernel_stats_default.md
#   traffic  two --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace domain) -> gpurun_out/TAG_step_hbm_traffic.md
#   dominant FETCH_SIZE / WRITE_SIZE of the layer bench.py's roofline object names -> gpurun_out/TAG_pmc_traffic.json
#   sq       one --pmc pass of SQ occupancy / MFMA-busy counters -> gpurun_out/TAG_sq_counters.md
TAG=$1; shift
WHAT="${*:-stats}"
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
# --single-stream: a kernel's duration (and its counters) are then properties of the kernel, not of what ran beside it
BENCH="python $REPO/bench.py --single-stream --no-cpu-baseline --no-roofline --no-gradsync --no-recipe"
for w in $WHAT; do
  case $w in
    stats)
      rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$TAG -o r -- $BENCH --steps 5 --warmup 3 > $REPO/gpurun_out/prof_$TAG.log 2>&1
      DB=$(find $REPO/gpurun_out/prof_$TAG -name '*results.db' | head -1)
      python $REPO/tools/rocpd_summary.py $DB $REPO/gpurun_out/${TAG}_kernel_stats.md "$TAG: bench.py --steps 5 --warmup 3 (8 steps profiled)"
      ;;
    stats_default)   # the DEFAULT command (recorded step replayed from the launch tape, side streams): the same kernels, counted
      rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/profd_$TAG -o r -- python $REPO/bench.py --no-cpu-baseline --no-roofline --no-gradsync --no-recipe --steps 5 --warmup 3 > $REPO/gpurun_out/profd_$TAG.log 2>&1
      DB=$(find $REPO/gpurun_out/profd_$TAG -name '*results.db' | head -1)
      python $REPO/tools/rocpd_summary.py $DB $REPO/gpurun_out/${TAG}_kernel_stats_default.md "$TAG: bench.py --steps 5 --warmup 3, DEFAULT mode = 3 eager warm-up steps + 1 + 5 replays of the recorded step + 1 + 5 eager steps (eager_step leg): 15 executed steps; kernels share the CUs across 4 streams, so durations are not the kernels' own"
      ;;
    traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c -d $REPO/gpurun_out/pmc_${TAG}_$c -o p --output-format csv -- $BENCH --steps 2 --warmup 1 > $REPO/gpurun_out/pmc_${TAG}_$c.log 2>&1
      done
      python $REPO/tools/pmc_traffic_step.py $REPO/gpurun_out/pmc_${TAG}_FETCH_SIZE $REPO/gpurun_out/pmc_${TAG}_WRITE_SIZE 3 $REPO/gpurun_out/${TAG}_step_class_traffic.json > $REPO/gpurun_out/${TAG}_step_hbm_traffic.md
      ;;
    dominant)
      for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c -d $REPO/gpurun_out/pmcd_${TAG}_$c -o p --output-format csv -- python $REPO/tools/conv_bench.py --only C512-K1024-3x3-H14 --kinds fprop,dgrad,wgrad --iters 2 > $REPO/gpurun_out/pmcd_${TAG}_$c.log 2>&1
      done
      python $REPO/tools/pmc_dominant.py $REPO/gpurun_out/pmcd_${TAG}_FETCH_SIZE $REPO/gpurun_out/pmcd_${TAG}_WRITE_SIZE $REPO/gpurun_out/${TAG}_pmc_traffic.json > /dev/null
      ;;
    sq)
      rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
        -d $REPO/gpurun_out/pmc_${TAG}_sq -o p --output-format csv -- $BENCH --steps 2 --warmup 1 > $REPO/gpurun_out/pmc_${TAG}_sq.log 2>&1
      python $REPO/tools/pmc_summary.py $REPO/gpurun_out/pmc_${TAG}_sq --md "$TAG: bench.py --steps 2 --warmup 1" > $REPO/gpurun_out/${TAG}_sq_counters.md
      ;;
  esac
done
cd $REPO
