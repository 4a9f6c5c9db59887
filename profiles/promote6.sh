#!/bin/bash
# Copies the judged evidence of one run_round6.sh output directory into profiles/ under the round's prefix and rebuilds the traffic JSON:
#   bash profiles/promote6.sh gpurun_out/r06 r06
set -eu
F=$1; R=$2; P=$(dirname "$0")
cp $F/bench.json $P/${R}_bench.json
cp $F/kernel_stats.md $P/${R}_kernel_stats.md
for k in fetch write mfma l2 ea dram; do cp $F/pmc_$k.md $P/${R}_pmc_$k.md; done
grep -i -E "^Counter_Name|^Description" $F/rocprof_counters.txt | grep -i -B1 -E "TCC_EA0|MALL|HBM|UMC|DRAM" | grep -v "^--" > $P/${R}_rocprof_counters.txt || true
for k in c3 c2 c1 137_100 137_100_nosave 137_100_notail 64_100_recursive; do cp $F/steps_$k.md $P/${R}_steps_$k.md; done
cp $F/tail_steps.md $P/${R}_tail_steps.md
for k in n2 n8; do grep '^{' $F/bench_${k}_gloo.json > $P/${R}_bench_${k}_one_gpu_gloo.json; done     # (gloo's own stdout chatter dropped)
grep '^{' $F/bench_c5_n2_gloo.json > $P/${R}_bench_c5_n2_one_gpu_gloo.json
cp $F/dense.log $P/${R}_dense.log
for e in c3_reference_loss c5_n1 c2_256 c1_64 137_100 c3_recursive c3_trivial c3_f2 c3_notail; do cp $F/extra_$e.json $P/${R}_extra_$e.json; done
( echo "# single-view iteration (render + fused losses + backward + Adam, 100 march steps), default policy (tail launch from the previous iteration's hint):"
  grep -v amdgpu.ids $F/loop.log
  echo "# the same with DISTR_TAIL=0 (one launch per march step to the end, rounds 1-5), same box, same call:"
  grep -v amdgpu.ids $F/loop_notail.log ) > $P/${R}_single_view_loop.log
cp $F/batch_round.log $P/${R}_batch_round.log
cp $F/cluster_phases.log $P/${R}_cluster_phases.log
( echo "# Soak run on one MI355X, round 6 tree (not part of the timed pytest -m gpu; profiles/run_round6.sh part c):"
  echo "#   DISTR_TEST_RANDOM_CONFIGS=96 DISTR_TEST_STRESS_ITERS=400 python -m pytest tests/test_gpu_parity.py -q -k \"random_configs or oversubscription or many_streams or cluster_fallback\""
  grep -E "passed|failed|^real" $F/soak.log
  echo "#   DISTR_XCHG_SC1=1 DISTR_TEST_STRESS_ITERS=200 ... -k \"oversubscription or cluster_tiles_bit\"   (write-through slice stores forced)"
  grep -E "passed|failed|^real" $F/soak_sc1.log
  echo "#   DISTR_CLUSTER_SPREAD=1 DISTR_TEST_STRESS_ITERS=200 ... -k \"oversubscription or cluster_tiles_bit\"   (members of every cluster on different XCDs: the real mixed-XCD exchange)"
  grep -E "passed|failed|^real" $F/soak_spread.log
  echo "#   six times: python -m pytest tests/test_gpu_tail.py -q -x   (tail launch: bit identity, hint, absent workgroups, batch, oracle, two streams, member drop-out, XCD spread)"
  grep -E "passed|failed|^real" $F/soak_tail.log
  if [ -f $F/soak_pyramids.log ]; then
    echo "#   DISTR_TEST_RANDOM_PYRAMIDS=96 python -m pytest tests/test_gpu_parity.py -q -k \"random_pyramids\"   (96 seeded pyramids, 2..4 levels, ratios 2..8: HIP vs oracle, zero mask flips)"
    grep -E "passed|failed|^real" $F/soak_pyramids.log
  fi ) > $P/${R}_soak.log
# the timed GPU test run: summary line, slowest tests, wall clock
if [ -f $F/pytest_gpu.log ]; then ( grep -E "passed|failed" $F/pytest_gpu.log | tail -1; grep -E "^real" $F/pytest_gpu.log; echo; grep -E "^[0-9.]+s (call|setup)" $F/pytest_gpu.log ) > $P/${R}_pytest_gpu_durations.log; fi
python $P/make_traffic.py $P/$R > /dev/null
ls -la $P | grep " ${R}_" | wc -l
