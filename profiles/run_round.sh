#!/bin/bash
# Collects the judged evidence of a round on the GPU box (run from the repo root THROUGH gpurun; everything lands in <out>/, which
# gpurun merges back; copy what is kept into profiles/ afterwards):
#   bench.json           default `python bench.py` (C3, 20 steps, cpu baselines)
#   kernel_stats.md      rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass`, condensed by summarize.py
#   pmc_fetch / pmc_write / pmc_mfma .md   separate rocprofv3 --pmc passes of the same command (never combined with tracing domains)
#   steps_c3.md (+ steps_c2 / _c1 / _137_100[_nosticky])   per-launch table of one forward (tests/gpu_diag_steps.py)
#   batch_round.log      phases, evaluations and march roofline of one batched multi-view round (tests/gpu_diag_batch.py)
#   dense.log            dense decoder rate (exact f32, bf16x6, f16x3) + in-kernel phase stamps (tests/gpu_diag_dense.py)
#   steps_c3_bf16x6 / _f16x3 .md   the C3 per-launch table in the opt-in arithmetic modes
#   ubench_*.log         profiles/ubench/split_bf16_bounds.hip, split_f16_layer.hip
#   kernel_stats_f16x3.md   rocprofv3 --kernel-trace --stats of the bench in the split-f16 mode
#   extra_*.json         other configurations through the same bench.py
# Usage: bash profiles/run_round.sh gpurun_out/r02_final
set -u
OUT=${1:-gpurun_out/round}
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python tests/gpu_diag_steps.py --out "$OUT/steps_c3.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 256 --march-step 50 --out "$OUT/steps_c2.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 64 --march-step 20 --out "$OUT/steps_c1.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --size 137 --march-step 100 --out "$OUT/steps_137_100.md" > /dev/null 2>&1
DISTR_STICKY=0 python tests/gpu_diag_steps.py --size 137 --march-step 100 --out "$OUT/steps_137_100_nosticky.md" > /dev/null 2>&1
python tests/gpu_diag_dense.py --stamps 2>&1 | grep -v amdgpu.ids > "$OUT/dense.log"
# the opt-in arithmetic modes, same per-launch table
python tests/gpu_diag_steps.py --arith bf16x6 --out "$OUT/steps_c3_bf16x6.md" > /dev/null 2>&1
python tests/gpu_diag_steps.py --arith f16x3 --out "$OUT/steps_c3_f16x3.md" > /dev/null 2>&1
for U in split_bf16_bounds split_f16_layer; do
  ( cd profiles/ubench && hipcc -O3 --offload-arch=gfx950 $U.hip -o /tmp/$U 2>/dev/null && /tmp/$U ) > "$OUT/ubench_$U.log" 2>&1
done
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass"      # the headline (exact f32) kernels only
R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_kt.log" 2>&1 )
python profiles/summarize.py /tmp/prof_kt "$OUT/kernel_stats.md" > /dev/null 2>&1
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  N=${P%%:*}; C=${P#*:}
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$N -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_$N.log" 2>&1 )
  python profiles/summarize.py /tmp/prof_$N "$OUT/pmc_$N.md" --pmc > /dev/null 2>&1
done
python bench.py --loss reference --no-cpu-baseline > "$OUT/extra_c3_reference_loss.json" 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/extra_c5_n1.json" 2>/dev/null
python bench.py --size 256 --no-cpu-baseline > "$OUT/extra_c2_256.json" 2>/dev/null
python bench.py --size 64 --march-step 20 --no-cpu-baseline > "$OUT/extra_c1_64.json" 2>/dev/null
python bench.py --marcher recursive --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/extra_c3_recursive.json" 2>/dev/null
python bench.py --marcher trivial --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/extra_c3_trivial.json" 2>/dev/null
python bench.py --fixture f2 --no-cpu-baseline > "$OUT/extra_c3_f2.json" 2>/dev/null
python bench.py --arith bf16x6 --no-cpu-baseline > "$OUT/extra_c3_bf16x6.json" 2>/dev/null
python bench.py --arith f16x3 --no-cpu-baseline > "$OUT/extra_c3_f16x3.json" 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt_h3 -- bash -c "cd $R && python bench.py --arith f16x3 --steps 5 --warmup 1 --no-cpu-baseline" > "$R/$OUT/rocprof_kt_f16x3.log" 2>&1 )
python profiles/summarize.py /tmp/prof_kt_h3 "$OUT/kernel_stats_f16x3.md" > /dev/null 2>&1
python tests/gpu_diag_loop.py 64 137 224 > "$OUT/loop.log" 2>&1
python tests/gpu_diag_multiview.py > "$OUT/multiview.log" 2>&1
python tests/gpu_diag_batch.py 137 8 recursive 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|Consider\|return Variable" > "$OUT/batch_round.log"
python tests/gpu_diag_grid.py > "$OUT/grid256.log" 2>&1
# cost of each of the eight C4 views on one GPU -> the row-band plan bench.py --gpus 8 would derive from them (view_balance.md)
for V in 0 1 2 3 4 5 6 7; do
  python bench.py --view-offset $V --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/view_$V.json" 2>/dev/null
done
python profiles/view_balance.py "$OUT" > "$OUT/view_balance.md" 2>&1
# the plan with the row cost profiles of the rendered masks, every rank's items timed on this GPU (bench.py --items)
python profiles/plan_check.py "$OUT" 2>&1 | grep -v amdgpu.ids > "$OUT/plan_check.md"
ls -la "$OUT" | tail -30
