#!/bin/bash
# Copies the judged evidence of one run_round5.sh output directory into profiles/ under the round's prefix and rebuilds the traffic JSON:
#   bash profiles/promote5.sh gpurun_out/r05_final r05
set -eu
F=$1; R=$2; P=$(dirname "$0")
cp $F/bench.json $P/${R}_bench.json
cp $F/kernel_stats.md $P/${R}_kernel_stats.md
for k in fetch write mfma l2; do cp $F/pmc_$k.md $P/${R}_pmc_$k.md; done
for k in c3 c2 c1 137_100; do cp $F/steps_$k.md $P/${R}_steps_$k.md; done
for k in n2 n8; do grep '^{' $F/bench_${k}_gloo.json > $P/${R}_bench_${k}_one_gpu_gloo.json; done     # (gloo's own stdout chatter dropped)
cp $F/dense.log $P/${R}_dense.log
for e in c3_reference_loss c5_n1 c2_256 c1_64 c3_recursive c3_trivial c3_f2; do cp $F/extra_$e.json $P/${R}_extra_$e.json; done
grep -v amdgpu.ids $F/loop.log > $P/${R}_single_view_loop.log
cp $F/batch_round.log $P/${R}_batch_round.log
cp $F/cluster_phases.log $P/${R}_cluster_phases.log
cp $F/ubench_mfma_chain_agpr.log $P/ubench/mfma_chain_agpr.log; cp $F/ubench_cluster_exchange2.log $P/ubench/cluster_exchange2.log
grep '^{' $F/bench_c5_n2_gloo.json > $P/${R}_bench_c5_n2_one_gpu_gloo.json
for k in 2 4 8; do cp $F/r05_plan_check_c5_n$k.md $P/${R}_plan_check_c5_n$k.md; done
# the timed GPU test run: summary line, slowest tests, wall clock
( grep -E "passed|failed" $F/pytest_gpu.log | tail -1; grep -E "^real" $F/pytest_gpu.log; echo; grep -E "^[0-9.]+s (call|setup)" $F/pytest_gpu.log ) > $P/${R}_pytest_gpu_durations.log
( echo "# Soak run on one MI355X, final tree of round 5 (not part of the timed pytest -m gpu; profiles/run_round5.sh):"
  echo "#   DISTR_TEST_RANDOM_CONFIGS=96 DISTR_TEST_STRESS_ITERS=400 python -m pytest tests/test_gpu_parity.py -q -k \"random_configs or oversubscription or many_streams or cluster_fallback\""
  grep -E "passed|failed|^real" $F/soak.log
  echo "#   DISTR_XCHG_SC1=1 DISTR_TEST_STRESS_ITERS=200 ... -k \"oversubscription or cluster_tiles_bit\"   (write-through slice stores forced: the mixed-XCD path)"
  grep -E "passed|failed|^real" $F/soak_sc1.log ) > $P/${R}_soak.log
cp $F/steps_137_100_nosave.md $P/${R}_steps_137_100_nosave.md
python $P/make_traffic.py $P/$R > /dev/null
ls -la $P | grep " ${R}_" | wc -l
