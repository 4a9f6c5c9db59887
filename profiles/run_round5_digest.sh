#!/bin/bash
# The digest-bound part of profiles/run_round5.sh alone (after a source edit that does not change the generated code, e.g. comments:
# distr.binding.source_digest() covers every byte of csrc/): the default bench line, the rocprofv3 kernel trace and the four separate PMC
# passes of the same command, and a second bench line once the traffic JSON of THIS run exists.
#   bash profiles/run_round5_digest.sh gpurun_out/r05_digest     (through gpurun; then: bash profiles/promote5_digest.sh gpurun_out/r05_digest r05)
set -u
OUT=${1:-gpurun_out/r05_digest}
mkdir -p "$OUT"
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fixture_f2.py -q -x -k "cluster or sticky or oversubscription or f2_hip_matches_reference_goldens or render_matches_reference_goldens or random_configs" ) > "$OUT/pytest_subset.log" 2>&1
python bench.py > "$OUT/bench_first.json" 2> "$OUT/bench.err"
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass"
R=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_kt.log" 2>&1 )
python profiles/summarize.py /tmp/prof_kt "$OUT/kernel_stats.md" > /dev/null 2>&1
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  N=${P%%:*}; C=${P#*:}
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$N -- bash -c "cd $R && $CMD" > "$R/$OUT/rocprof_$N.log" 2>&1 )
  python profiles/summarize.py /tmp/prof_$N "$OUT/pmc_$N.md" --pmc > /dev/null 2>&1
done
# the traffic JSON of this run (digest from bench_first.json), then the bench line that quotes it
mkdir -p "$OUT/p"; cp "$OUT/bench_first.json" "$OUT/p/r05_bench.json"; for k in fetch write mfma l2; do cp "$OUT/pmc_$k.md" "$OUT/p/r05_pmc_$k.md"; done
python profiles/make_traffic.py "$OUT/p/r05" > /dev/null && cp "$OUT/p/r05_traffic.json" profiles/r05_traffic.json && cp "$OUT/p/r05_traffic.json" "$OUT/traffic.json"
python bench.py > "$OUT/bench.json" 2>> "$OUT/bench.err"
tail -3 "$OUT/pytest_subset.log"
python -c "
import json; j=json.load(open('$OUT/bench.json')); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['traffic'], r['traffic_csrc_sha256']==r['csrc_sha256'])"
