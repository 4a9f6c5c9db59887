#!/bin/bash
# Copies the judged evidence of one run_round4.sh output directory into profiles/ under the round's prefix and rebuilds the traffic JSON:
#   bash profiles/promote4.sh gpurun_out/r04_final r04
set -eu
F=$1; R=$2; P=$(dirname "$0")
cp $F/bench.json $P/${R}_bench.json
cp $F/kernel_stats.md $P/${R}_kernel_stats.md
for k in fetch write mfma l2; do cp $F/pmc_$k.md $P/${R}_pmc_$k.md; done
for k in c3 c2 c1 137_100; do cp $F/steps_$k.md $P/${R}_steps_$k.md; done
for k in n2 n8; do cp $F/bench_${k}_gloo.json $P/${R}_bench_${k}_one_gpu_gloo.json; done
cp $F/dense.log $P/${R}_dense.log
for e in c3_reference_loss c5_n1 c2_256 c1_64 c3_recursive c3_trivial c3_f2; do cp $F/extra_$e.json $P/${R}_extra_$e.json; done
grep -v amdgpu.ids $F/loop.log > $P/${R}_single_view_loop.log
cp $F/batch_round.log $P/${R}_batch_round.log
# the timed GPU test run: summary line, slowest tests, wall clock
( grep -E "passed|failed" $F/pytest_gpu.log | tail -1; grep -E "^real" $F/pytest_gpu.log; echo; grep -E "^[0-9.]+s (call|setup)" $F/pytest_gpu.log ) > $P/${R}_pytest_gpu_durations.log
python $P/make_traffic.py $P/$R > /dev/null
ls -la $P | grep " ${R}_" | wc -l
