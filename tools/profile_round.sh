#!/bin/bash
# Collects the evidence kept under profiles/: per-kernel durations (rocprofv3 --kernel-trace --stats) and HBM
# traffic (PMC WRITE_SIZE / FETCH_SIZE, each in its own pass, never combined with tracing domains other than
# --kernel-trace).  Run on the GPU box from the repo root:  bash tools/profile_round.sh r01
# Outputs land in gpurun_out/prof/; tools/make_profiles.py turns them into profiles/<round>_*.
set -u
R=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SKS_BENCH_TRAFFIC=0     # (bench.py's own in-run PMC child passes stay out of a bench.py that is itself being profiled)
for WL in h36m panoptic; do
  # per-kernel averages of the HEADLINE form of the step (--form two: sks_forward, then sks_backward: the forward alone on the chip)
  # and of the one-call form (sks_forward_backward, the backward beside the forward): two runs, so that neither average is a mixture
  # (no tuning inside these runs -- its candidates' launches would be averaged in: the fill configuration Workspace.tune picks on
  # this pool -- the library's default for the two-call step, three non-temporal passes per fill block for the H36M one-call step --
  # is set by hand)
  T1=0; [ $WL = h36m ] && T1=0x300
  rm -rf "$OUT/${WL}_stats" "$OUT/${WL}1_stats"
  SKS_BENCH_AUTOTUNE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${WL}_stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --form two --steps 100 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/${WL}_bench_under_rocprof.json" 2> "$OUT/${WL}_stats.log"
  SKS_BENCH_AUTOTUNE=0 SKS_FWD_TUNE=$T1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${WL}1_stats" -o stats -- python3 "$ROOT/bench.py" --workload $WL --form one --steps 100 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/${WL}_bench_one_call_under_rocprof.json" 2> "$OUT/${WL}1_stats.log"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${WL}_w" -o w -- python3 "$ROOT/bench.py" --workload $WL --form two --steps 20 --warmup 3 --no-cpu-baseline --no-prof --no-extras > /dev/null 2> "$OUT/${WL}_w.log"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${WL}_f" -o f -- python3 "$ROOT/bench.py" --workload $WL --form two --steps 20 --warmup 3 --no-cpu-baseline --no-prof --no-extras > /dev/null 2> "$OUT/${WL}_f.log"
done
unset SKS_BENCH_TRAFFIC
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stress_stats" -o stats -- python3 "$ROOT/tools/bench_stress.py" > "$OUT/stress.log" 2>&1
# HBM traffic of the binned path's kernels (separate PMC passes, --kernel-trace only) and the timeline of one forward call
bash "$ROOT/tools/pmc_traffic_stress.sh" "$OUT" > "$OUT/stress_traffic.txt" 2>&1
bash "$ROOT/tools/trace_stress.sh" > "$OUT/stress_timeline.txt" 2>&1
cd /tmp
# the frame-batched loop: 16 H36M frames per launch (kernel stats of that run alone), then the frames/s table
ONLY_BATCH=1 STREAMS= ITERS=200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/frames_stats" -o stats -- python3 "$ROOT/tools/bench_frames.py" 16 > "$OUT/frames_under_rocprof.log" 2>&1
cd "$ROOT"
python3 tools/bench_ssim.py > "$OUT/bench_ssim.txt" 2>/dev/null
bash tools/pmc_ssim.sh fwd 4,17,1000,1000 > "$OUT/ssim_pmc_fwd.txt" 2>&1
bash tools/pmc_ssim.sh train 4,17,1000,1000 > "$OUT/ssim_pmc_train.txt" 2>&1
cd "$ROOT"
python3 bench.py > "$OUT/h36m_bench.json" 2> "$OUT/h36m_bench.log"
python3 bench.py --workload panoptic --steps 50 --warmup 5 > "$OUT/panoptic_bench.json" 2> "$OUT/panoptic_bench.log"
# the sharded path at world size 1 (RCCL all_gather included, the group also replayed as a hipGraph)
SKS_BENCH_FORCE_DIST=1 SKS_GRAPH_COLLECTIVES=1 python3 bench.py --steps 50 --warmup 5 > "$OUT/sharded_world1_bench.json" 2> "$OUT/sharded_world1_bench.log"
python3 tools/width_sweep2.py 1000,1002,1024,1920 0 > "$OUT/width_sweep.txt" 2>&1
python3 tools/bench_frames.py 1 2 4 8 16 2>/dev/null | grep "frames/s" > "$OUT/frames.txt"
FACTORED=0 ONLY_BATCH=1 python3 tools/bench_frames.py 16 2>/dev/null | grep "frames/s" | sed 's/^/planes (factored=False): /' >> "$OUT/frames.txt"
# the literal drop-in iteration (render -> criterion -> loss.backward(), one view at a time): kernel averages + one iteration's timeline,
# with the fused criterion and with the reference's tensor-op criterion (whose own ops are ~560 us of GPU time per iteration)
bash tools/dropin_trace.sh "$OUT/dropin_trace_fused.txt" > /dev/null 2>&1
bash tools/dropin_trace.sh --tensor "$OUT/dropin_trace_tensor.txt" > /dev/null 2>&1
# the one-call step's kernel timeline (two queues), SQ counter tables of the backward kernels and of fused SSIM, the
# calibration of the fuzz sweep's rounding allowance, passes per fill block in both forms
bash tools/trace_one_call.sh h36m > "$OUT/one_call_timeline.txt" 2>&1
bash tools/trace_one_call.sh panoptic4 > "$OUT/one_call_timeline_rank_step.txt" 2>&1
bash tools/trace_loop.sh > "$OUT/loop_timeline.txt" 2>&1
bash tools/counters.sh $R > /dev/null 2>&1
for f in bwd_tile_stress bwd_wave_h36m bwd_wave_panoptic ssim_fwd ssim_train; do cp "$ROOT/gpurun_out/${R}_pmc_$f.txt" "$OUT/pmc_$f.txt"; done
python3 tools/fuzz_bound_calib.py 4000 10000 > "$OUT/fuzz_bound_calib.txt" 2>&1
bash tools/ab_split.sh 2>&1 | grep -v amdgpu.ids > "$OUT/fill_passes_sweep.txt"
# round 6: the forward over one output set vs eight in turn (what roofline.frac / frac_same_buffer are made of); the stress step as
# two calls vs one call with view groups on two streams (the A/B that closes the binned overlap); the drop-in iteration's host side
python3 tools/probe_rotating.py 2>&1 | grep -v amdgpu.ids > "$OUT/probe_rotating.txt"
for G in 1 2 4; do SKS_BIN_GROUPS=$G python3 tools/bench_stress_forms.py 5 2>&1 | grep -v amdgpu.ids | tail -2; done > "$OUT/stress_forms.txt"
python3 tools/profile_dropin_host.py 2>&1 | grep -v amdgpu.ids | head -60 > "$OUT/dropin_host.txt"
find "$OUT" -name "*.csv" | head -40
