#!/bin/bash
# Builds and runs the microbenchmarks on the GPU box and keeps their OUTPUT (profiles/ubench/*.log are the judged evidence;
# binaries are not tracked). rocm-smi samples clocks / power every 0.5 s beside the sustained MFMA stream.
# Usage (from the repo root, on the box): bash profiles/ubench/run_ubench.sh <out_dir>
set -u
OUT=${1:-gpurun_out/ubench}
mkdir -p "$OUT"
cd "$(dirname "$0")"
for b in mfma_sustain mfma_rate mfma_chain dual_pipe cluster_exchange split_bf16_layer; do
  [ -f $b.hip ] && /opt/rocm/bin/hipcc -O3 -w --offload-arch=gfx950 $b.hip -o /tmp/$b 2> "$OLDPWD/$OUT/$b.build.log"
done
cd "$OLDPWD"
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" ; sleep 0.5; done ) > "$OUT/rocm_smi_during_mfma_sustain.log" 2>&1 &
SMI=$!
/tmp/mfma_sustain 4 > "$OUT/mfma_sustain.log" 2>&1
kill $SMI 2>/dev/null
/tmp/mfma_rate > "$OUT/mfma_rate.log" 2>&1
/tmp/mfma_chain > "$OUT/mfma_chain.log" 2>&1
/tmp/dual_pipe > "$OUT/dual_pipe.log" 2>&1
/tmp/cluster_exchange > "$OUT/cluster_exchange.log" 2>&1
/tmp/split_bf16_layer > "$OUT/split_bf16_layer.log" 2>&1
rocm-smi --showclocks --showpower --showmaxpower > "$OUT/rocm_smi_idle.log" 2>&1
tail -n 40 "$OUT/mfma_sustain.log"
