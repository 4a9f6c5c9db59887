#!/bin/bash
# Round 5 evidence (run from the repo root THROUGH gpurun; everything lands in <out>/, which gpurun merges back; promote5.sh copies what
# is kept into profiles/). Round 5 changed the cluster tiles of the exact-f32 march (pipelined granule hand-off): the same set as round 4
# (timed pytest -m gpu, bench lines, rocprofv3 kernel trace + the four separate PMC passes, per-launch step tables, single-view loops)
# plus the cluster phase stamps and the C5 partition emulation.
#   pytest_gpu.log       python -m pytest tests -m gpu -q --durations=30 (wall time of the driver's step; budget <= 360 s)
#   bench.json           default `python bench.py` (C3, 20 steps, cpu baselines)
#   bench_n2_gloo.json, bench_n8_gloo.json   the N > 1 protocol on this ONE GPU (ranks time-share it over gloo: NOT a scaling measurement,
#                        the lines say so): serial check, both timings, per-rank breakdown
#   kernel_stats.md      rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass`
#   pmc_{fetch,write,mfma,l2}.md   separate rocprofv3 --pmc passes of the same command (never combined with tracing domains)
#   steps_{c3,c2,c1,137_100}.md    per-launch table of one forward
#   extra_*.json         other configurations through the same bench.py
# Usage: bash profiles/run_round5.sh gpurun_out/r05_final
set -u
OUT=${1:-gpurun_out/r05_round}
mkdir -p "$OUT"
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q --durations=30 ) > "$OUT/pytest_gpu.log" 2>&1
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
DISTR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/bench_n2_gloo.json" 2> "$OUT/bench_n2_gloo.err"
DISTR_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 3 --warmup 1 > "$OUT/bench_n8_gloo.json" 2> "$OUT/bench_n8_gloo.err"
python tests/gpu_diag_steps.py --out "$OUT/steps_c3.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 256 --march-step 50 --out "$OUT/steps_c2.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 64 --march-step 20 --out "$OUT/steps_c1.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 137 --march-step 100 --out "$OUT/steps_137_100.md" > /dev/null 2>&1
python tests/gpu_diag_dense.py 2>&1 | grep -v amdgpu.ids > "$OUT/dense.log"
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass"      # the headline (exact f32) kernels only
R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_kt.log" 2>&1 )
python profiles/summarize.py /tmp/prof_kt "$OUT/kernel_stats.md" > /dev/null 2>&1
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  N=${P%%:*}; C=${P#*:}
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$N -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_$N.log" 2>&1 )
  python profiles/summarize.py /tmp/prof_$N "$OUT/pmc_$N.md" --pmc > /dev/null 2>&1
done
python bench.py --loss reference --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c3_reference_loss.json" 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c5_n1.json" 2>/dev/null
python bench.py --size 256 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c2_256.json" 2>/dev/null
python bench.py --size 64 --march-step 20 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c1_64.json" 2>/dev/null
python bench.py --marcher recursive --steps 5 --warmup 2 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c3_recursive.json" 2>/dev/null
python bench.py --marcher trivial --steps 3 --warmup 1 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c3_trivial.json" 2>/dev/null
python bench.py --fixture f2 --no-cpu-baseline --no-split-bf16-pass > "$OUT/extra_c3_f2.json" 2>/dev/null
python tests/gpu_diag_loop.py 64 137 224 > "$OUT/loop.log" 2>&1
python tests/gpu_diag_cluster.py 64 99 2>&1 | grep -v amdgpu.ids > "$OUT/cluster_phases.log"
for b in cluster_exchange2 mfma_chain_agpr; do [ -x profiles/ubench/$b ] && timeout 120 profiles/ubench/$b > "$OUT/ubench_$b.log" 2>&1; done
DISTR_DIST_BACKEND=gloo python bench.py --workload c5 --gpus 2 --steps 2 --warmup 1 > "$OUT/bench_c5_n2_gloo.json" 2> "$OUT/bench_c5_n2_gloo.err"
python profiles/plan_check_c5.py "$OUT" > "$OUT/plan_check_c5.log" 2>&1
python tests/gpu_diag_batch.py 137 8 recursive 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|Consider\|return Variable" > "$OUT/batch_round.log"
# soak (not part of the timed pytest -m gpu): 96 seeded random renderer configurations HIP vs oracle, the 8-stream oversubscription stress at 400
# iterations, many streams, forced cluster fallback; then the stress with write-through slice stores forced (the mixed-XCD path)
( time DISTR_TEST_RANDOM_CONFIGS=96 DISTR_TEST_STRESS_ITERS=400 python -m pytest tests/test_gpu_parity.py -q -k "random_configs or oversubscription or many_streams or cluster_fallback" ) > "$OUT/soak.log" 2>&1
( time DISTR_XCHG_SC1=1 DISTR_TEST_STRESS_ITERS=200 python -m pytest tests/test_gpu_parity.py -q -k "oversubscription or cluster_tiles_bit" ) > "$OUT/soak_sc1.log" 2>&1
python tests/gpu_diag_steps.py --size 137 --march-step 100 --no-save --out "$OUT/steps_137_100_nosave.md" > /dev/null 2>&1
ls -la "$OUT" | tail -40
