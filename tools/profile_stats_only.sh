#!/bin/bash
# Only the four rocprofv3 --kernel-trace --stats runs of tools/profile_round.sh (per-kernel averages of the two forms of the step).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for WL in h36m panoptic; do
  T1=0; T2=0; [ $WL = h36m ] && T1=0x510 && T2=0x310     # (0x510: five passes, plain stores -- SKS_NO_NT_STORES is 0x10 -- the tuner's usual pick for the one-call step; 0x310 for the two-call step)
  rm -rf "$OUT/${WL}_stats" "$OUT/${WL}2_stats"
  SKS_BENCH_AUTOTUNE=0 SKS_FWD_TUNE=$T1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${WL}_stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --form one --steps 100 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/${WL}_bench_under_rocprof.json" 2> "$OUT/${WL}_stats.log"
  SKS_BENCH_AUTOTUNE=0 SKS_FWD_TUNE=$T2 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${WL}2_stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --form two --steps 100 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/${WL}_bench_two_calls_under_rocprof.json" 2> "$OUT/${WL}2_stats.log"
done
cat "$OUT/h36m_bench_under_rocprof.json" | head -c 1500
