#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_round.sh (gpurun_out/prof/) into the files kept under profiles/:

  <round>_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary, H36M workload, the headline (two-call) form of the step
  <round>_kernel_stats_one_call.csv   same, the one-call form (sks_forward_backward)
  <round>_kernel_stats_panoptic.csv   same, Panoptic 31-view workload
  <round>_kernel_stats_stress.csv     same, tools/bench_stress.py (P = 4352, 8 views, 2048^2, binned path)
  <round>_bench.json / <round>_bench_panoptic.json    the bench.py lines of the same box
  <round>_kernel_stats_frames.csv     same, tools/bench_frames.py 16 (frame-batched loop); <round>_frames.txt: frames/s table
  traffic.json                        HBM bytes per launch per kernel from the PMC passes (bench.py reads this); the stress
                                      workload's passes run tools/bench_stress.py (tools/pmc_traffic_stress.sh)
  <round>_stress_traffic.txt / _stress_timeline.txt   the same figures as text; start / end of every kernel of one binned forward call

PMC units and corrections as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): WRITE_SIZE and FETCH_SIZE
are reported in KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, so it is doubled.
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
NAMES = {"h36m": "h36m_4view_1000x1000_P17_C17", "panoptic": "panoptic_31view_1920x1080_P19_C19",
         "stress": "stress_256skeletons_8view_2048x2048_P4352_C17"}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()


def pmc(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


os.makedirs(DST, exist_ok=True)
traffic = {"round": rnd,
           "source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) "
                     "-- python3 bench.py --workload W --steps 20 --warmup 3 --no-cpu-baseline --no-prof",
           "note": "KiB units; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md (HBM section)"}
for wl, tag in (("h36m", ""), ("panoptic", "_panoptic"), ("stress", "_stress")):
    st = glob.glob(os.path.join(SRC, f"{wl}_stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(DST, f"{rnd}_kernel_stats{tag}.csv"))
    st1 = glob.glob(os.path.join(SRC, f"{wl}1_stats", "**", "*kernel_stats.csv"), recursive=True)
    if st1:   # the one-call form of the step (bench.py --form one): the backward beside the forward (the headline is the two calls)
        shutil.copy(st1[0], os.path.join(DST, f"{rnd}_kernel_stats{tag}_one_call.csv"))
    b = os.path.join(SRC, f"{wl}_bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(DST, f"{rnd}_bench{tag}.json"))
    w = glob.glob(os.path.join(SRC, f"{wl}_w", "**", "*counter_collection.csv"), recursive=True)
    f = glob.glob(os.path.join(SRC, f"{wl}_f", "**", "*counter_collection.csv"), recursive=True)
    if not (w and f):
        continue
    W, nW = pmc(w[0], "WRITE_SIZE")
    F, nF = pmc(f[0], "FETCH_SIZE")
    ks = {}
    for k in sorted(set(W) | set(F)):
        if not k.startswith("k_"):
            continue
        wk, fk = W.get(k, 0.0), F.get(k, 0.0)
        ks[k] = {"launches": nW.get(k, 0), "WRITE_SIZE_KiB_per_launch": wk, "FETCH_SIZE_KiB_per_launch": fk,
                 "hbm_bytes_per_launch": 1024.0 * (wk + 2.0 * fk)}
    entry = {"kernels": ks}
    fwd = [k for k in ks if k.startswith("k_render_fwd_sparse") or k.startswith("k_render_fwd_binned")]
    if fwd:
        entry["fwd_bytes_per_launch"] = ks[fwd[0]]["hbm_bytes_per_launch"]
    traffic[NAMES[wl]] = entry
    # bench.py read the traffic.json committed BEFORE this run; the copied bench line gets this run's own PMC figure
    bj = os.path.join(DST, f"{rnd}_bench{tag}.json")
    if fwd and os.path.exists(bj):
        line = json.load(open(bj))
        if "roofline" in line:
            line["roofline"]["traffic"] = entry["fwd_bytes_per_launch"]
            json.dump(line, open(bj, "w"))
st = glob.glob(os.path.join(SRC, "frames_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:   # tools/bench_frames.py 16: the frame-batched loop (16 H36M frames per launch)
    shutil.copy(st[0], os.path.join(DST, f"{rnd}_kernel_stats_frames.csv"))
for name in ("bench_ssim.txt", "ssim_pmc_fwd.txt", "ssim_pmc_train.txt", "sharded_world1_bench.json", "width_sweep.txt",
             "frames.txt", "stress_traffic.txt", "stress_timeline.txt", "stress.log",
             "dropin_trace_fused.txt", "dropin_trace_tensor.txt", "one_call_timeline.txt", "one_call_timeline_rank_step.txt",
             "pmc_bwd_tile_stress.txt", "pmc_bwd_wave_h36m.txt", "pmc_bwd_wave_panoptic.txt", "pmc_ssim_fwd.txt", "pmc_ssim_train.txt",
             "fuzz_bound_calib.txt", "fill_passes_sweep.txt", "loop_timeline.txt", "probe_rotating.txt", "stress_forms.txt",
             "dropin_host.txt"):   # fused-SSIM timings and SQ counter passes, the sharded path at world 1, sweeps
    src = os.path.join(SRC, name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, f"{rnd}_{name}"))
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(open(os.path.join(DST, "traffic.json")).read()[:3000])
