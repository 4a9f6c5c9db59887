#!/bin/bash
# Copies the judged evidence of one run_round.sh output directory into profiles/ under the round's prefix and rebuilds the traffic
# JSON:   bash profiles/promote.sh gpurun_out/r02_final2 r02
set -eu
F=$1; R=$2; P=$(dirname "$0")
cp $F/bench.json $P/${R}_bench.json
cp $F/kernel_stats.md $P/${R}_kernel_stats.md
for k in fetch write mfma l2; do cp $F/pmc_$k.md $P/${R}_pmc_$k.md; done
cp $F/steps_c3.md $P/${R}_steps_c3.md
for k in c2 c1 137_100 137_100_nosticky c3_bf16x6 c3_f16x3; do [ -f $F/steps_$k.md ] && cp $F/steps_$k.md $P/${R}_steps_$k.md; done
for k in c3_bf16x6 c3_f16x3; do [ -f $F/extra_$k.json ] && cp $F/extra_$k.json $P/${R}_extra_$k.json; done
[ -f $F/kernel_stats_f16x3.md ] && cp $F/kernel_stats_f16x3.md $P/${R}_kernel_stats_f16x3.md
for U in split_bf16_bounds split_f16_layer; do [ -f $F/ubench_$U.log ] && cp $F/ubench_$U.log $P/ubench/$U.log; done
[ -f $F/batch_round.log ] && cp $F/batch_round.log $P/${R}_batch_round.log
[ -f $F/extra_c3_f2.json ] && cp $F/extra_c3_f2.json $P/${R}_extra_c3_f2.json
cp $F/dense.log $P/${R}_dense.log
for e in c3_reference_loss c5_n1 c2_256 c1_64 c3_recursive c3_trivial; do cp $F/extra_$e.json $P/${R}_extra_$e.json; done
grep -v amdgpu.ids $F/loop.log > $P/${R}_single_view_loop.log
grep "^[0-9]*x[0-9]*, 8 view pairs" $F/multiview.log > $P/${R}_multiview_round.log || true
cp $F/grid256.log $P/${R}_grid256.log
cp $F/view_balance.md $P/${R}_view_balance.md
[ -f $F/plan_check.md ] && cp $F/plan_check.md $P/${R}_plan_check_n8.md
if [ -f $F/cluster_phases_c8.log ]; then
  (echo "== CL = 8"; grep -v amdgpu $F/cluster_phases_c8.log; echo "== CL = 4 (DISTR_CLUSTER=4)"; grep -v amdgpu $F/cluster_phases_c4.log) > $P/${R}_cluster_phases.log
fi
python $P/make_traffic.py $P/$R > /dev/null
ls -la $P | grep " ${R}_" | wc -l
