#!/bin/bash
# the GPU test tier and the driver's bench line, as the driver runs them at round end (GPU box):  bash tools/driver_tier.sh
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
( time python bench.py ) > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench.json"))
def show(d, keys):
    return {k: d.get(k) for k in keys}
print(show(r, ["value", "ms_per_step"]), r["config"]["form"], r["config"]["autotuned"]["fill_role"])
ro = r["roofline"]
print("roofline", show(ro, ["frac", "frac_same_buffer", "avg_launch_us", "stores", "whole_step_frac", "traffic_over_algorithmic", "hbm_write_rate_by_zero_fill_GBps", "frac_of_zero_fill_fresh_memory"]))
print("one_call", show(r.get("one_call_step", {}), ["ms_per_step", "views_per_s"]))
print("plain", r.get("same_buffer_plain_stores"))
for k in ("grad_step_ms", "grad_step_ms_hipgraph", "dropin_iteration_ms", "dropin_iteration_fused_loss_ms", "dropin_iteration_fused_loss_one_launch_adam_ms", "loop_error", "dropin_error"):
    print(k, r.get(k))
for k in ("panoptic", "stress", "h36m_mixed_1002"):
    print(k, json.dumps(r.get(k))[:1500])
rs = r.get("rank_step_8gpu", {})
print("rank_step", {k: rs.get(k) for k in ("error", "predicted_8gpu_speedup", "predicted_8gpu_speedup_at_30us_wire", "predicted_8gpu_speedup_one_call", "predicted_8gpu_speedup_one_call_at_30us_wire", "repetitions")})
for f in ("two_calls", "one_call"):
    b = rs.get(f, {})
    print(f, {k: (b[k]["rank_step_ms"] if isinstance(b.get(k), dict) else b.get(k)) for k in ("no_exchange", "exchange_0us", "exchange_10us", "exchange_20us", "exchange_30us", "three_view_rank_exchange_0us_ms", "consistent", "inversions", "extra_sampling_rounds")})
print("cpu", r.get("cpu_baseline"))
PY
