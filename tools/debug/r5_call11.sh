#!/bin/bash
O=gpurun_out/r5_11
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dp_rccl.py -q > $O/tests_dp.log 2>&1; echo "dp tests rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "tape or graph or records" > $O/tests_tape.log 2>&1; echo "tape tests rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/tests_dp.log $O/tests_tape.log | tail -4
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
tail -2 $O/smoke.log | cut -c1-400
bash tools/profile_round.sh round5_a stats stats_default traffic sq > $O/profile.log 2>&1; echo "profile rc=$?" >> $O/summary.txt
ls gpurun_out/round5_a_* 2>/dev/null
timeout 900 python bench.py --dump-convs gpurun_out/round5_a_per_conv_in_situ.md > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -1 $O/bench.log | cut -c1-300
cat $O/summary.txt
