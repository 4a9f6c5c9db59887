#!/bin/bash
# Round 6 evidence (run from the repo root THROUGH gpurun; everything lands in <out>/, which gpurun merges back; promote6.sh copies what
# is kept into profiles/). Round 6: the persistent tail launch (k_tail), no scratch in the march kernels, bench.py's extra.small_renders.
# Same set as round 5 plus: the tail-launch table (tests/gpu_diag_tail.py), the 64 x 64 / 100-step table, a PMC pass with the L2's
# memory-side read counters split by destination (where the fetched bytes come from), and the list of counters this rocprofv3 knows.
#   bash profiles/run_round6.sh gpurun_out/r06 [part]      part: a (bench + traces + PMC), b (tables + loops + extras), c (soak); default all
set -u
OUT=${1:-gpurun_out/r06}
PART=${2:-abc}
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$(pwd)
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass --no-small-renders --no-live-traffic"      # the headline (exact f32) kernels only
if [[ $PART == *a* ]]; then
  python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  ( cd /tmp && rocprofv3 -L > "$R/$OUT/rocprof_counters.txt" 2>&1 )
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_kt.log" 2>&1 )
  python profiles/summarize.py /tmp/prof_kt "$OUT/kernel_stats.md" > /dev/null 2>&1
  for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l2:TCC_HIT_sum TCC_MISS_sum" \
           "ea:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "gmi:TCC_EA0_RDREQ_GMI_sum TCC_EA0_RDREQ_IO_sum"; do
    N=${P%%:*}; C=${P#*:}
    ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$N -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_$N.log" 2>&1 )
    python profiles/summarize.py /tmp/prof_$N "$OUT/pmc_$N.md" --pmc > /dev/null 2>&1
  done
  DISTR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/bench_n2_gloo.json" 2> "$OUT/bench_n2_gloo.err"
  DISTR_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 3 --warmup 1 > "$OUT/bench_n8_gloo.json" 2> "$OUT/bench_n8_gloo.err"
  DISTR_DIST_BACKEND=gloo python bench.py --workload c5 --gpus 2 --steps 2 --warmup 1 > "$OUT/bench_c5_n2_gloo.json" 2> "$OUT/bench_c5_n2_gloo.err"
fi
if [[ $PART == *b* ]]; then
  python tests/gpu_diag_steps.py --out "$OUT/steps_c3.md" > /dev/null 2>&1
  python tests/gpu_diag_steps.py --size 256 --march-step 50 --out "$OUT/steps_c2.md" > /dev/null 2>&1
  python tests/gpu_diag_steps.py --size 64 --march-step 20 --out "$OUT/steps_c1.md" > /dev/null 2>&1
  python tests/gpu_diag_steps.py --size 137 --march-step 100 --out "$OUT/steps_137_100.md" > /dev/null 2>&1
  python tests/gpu_diag_steps.py --size 137 --march-step 100 --no-save --out "$OUT/steps_137_100_nosave.md" > /dev/null 2>&1
  python tests/gpu_diag_steps.py --size 64 --march-step 100 --marcher recursive --out "$OUT/steps_64_100_recursive.md" > /dev/null 2>&1
  DISTR_TAIL=0 python tests/gpu_diag_steps.py --size 137 --march-step 100 --out "$OUT/steps_137_100_notail.md" > /dev/null 2>&1
  python tests/gpu_diag_tail.py --out "$OUT/tail_steps.md" > "$OUT/tail_steps.log" 2>&1
  python tests/gpu_diag_dense.py 2>&1 | grep -v amdgpu.ids > "$OUT/dense.log"
  python tests/gpu_diag_loop.py 64 137 224 > "$OUT/loop.log" 2>&1
  DISTR_TAIL=0 python tests/gpu_diag_loop.py 64 137 224 > "$OUT/loop_notail.log" 2>&1
  python tests/gpu_diag_cluster.py 64 99 2>&1 | grep -v amdgpu.ids > "$OUT/cluster_phases.log"
  python tests/gpu_diag_batch.py 137 8 recursive 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|Consider\|return Variable" > "$OUT/batch_round.log"
  python bench.py --loss reference --no-cpu-baseline --no-split-bf16-pass --no-small-renders > "$OUT/extra_c3_reference_loss.json" 2>/dev/null
  python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c5_n1.json" 2>/dev/null
  python bench.py --size 256 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c2_256.json" 2>/dev/null
  python bench.py --size 64 --march-step 20 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c1_64.json" 2>/dev/null
  python bench.py --size 137 --march-step 100 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_137_100.json" 2>/dev/null
  python bench.py --marcher recursive --steps 5 --warmup 2 --no-cpu-baseline --no-split-bf16-pass --no-small-renders > "$OUT/extra_c3_recursive.json" 2>/dev/null
  python bench.py --marcher trivial --steps 3 --warmup 1 --no-cpu-baseline --no-split-bf16-pass --no-small-renders > "$OUT/extra_c3_trivial.json" 2>/dev/null
  python bench.py --fixture f2 --no-cpu-baseline --no-split-bf16-pass --no-small-renders > "$OUT/extra_c3_f2.json" 2>/dev/null
  DISTR_TAIL=0 python bench.py --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c3_notail.json" 2>/dev/null
fi
if [[ $PART == *c* ]]; then
  # soak (not part of the timed pytest -m gpu): 96 seeded random renderer configurations HIP vs oracle, the 8-stream oversubscription stress at 400
  # iterations (tail launches of eight streams competing for the compute units), many streams, forced cluster fallback; the stress with
  # write-through slice stores forced and with the members of every cluster spread over the XCDs; the tail tests with absent workgroups
  ( time DISTR_TEST_RANDOM_CONFIGS=96 DISTR_TEST_STRESS_ITERS=400 python -m pytest tests/test_gpu_parity.py -q -k "random_configs or oversubscription or many_streams or cluster_fallback" ) > "$OUT/soak.log" 2>&1
  ( time DISTR_XCHG_SC1=1 DISTR_TEST_STRESS_ITERS=200 python -m pytest tests/test_gpu_parity.py -q -k "oversubscription or cluster_tiles_bit" ) > "$OUT/soak_sc1.log" 2>&1
  ( time DISTR_CLUSTER_SPREAD=1 DISTR_TEST_STRESS_ITERS=200 python -m pytest tests/test_gpu_parity.py -q -k "oversubscription or cluster_tiles_bit" ) > "$OUT/soak_spread.log" 2>&1
  ( time bash -c 'for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_tail.py -q -x || exit 1; done' ) > "$OUT/soak_tail.log" 2>&1
  ( time DISTR_TEST_RANDOM_PYRAMIDS=96 python -m pytest tests/test_gpu_parity.py -q -k "random_pyramids" ) > "$OUT/soak_pyramids.log" 2>&1
fi
ls -la "$OUT" | tail -60
