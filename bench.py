#!/usr/bin/env python3
"""bench.py -- rendered+backpropagated views/s of the skeletal-Gaussian rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W     (N > 1: one rank per GPU under torch.distributed.run; a plain
                                                        `python bench.py --gpus N` starts that launcher itself)

A "step" is one accumulation group of the reference loop (train.py:130-222): the V views that share the Gaussian
parameters are rendered (forward) and back-propagated (backward) through the C ABI with the upstream gradient
dL/d(render) already resident in HBM, then the per-view joint gradients are averaged (train.py:215-217).

N = 1 (default): BASELINE.json configs[1] -- H36M, 17 joints, 4 views @ 1000x1000 (synthetic skeleton + cameras).
    Extras on the same line: configs[2] (Panoptic 31 views @ 1920x1080), configs[4] (stress, binned path), the real H36M
    sensor mix (1002- and 1000-wide views in one group), the full loop step (render + masked-L2 + backward + Adam).
N > 1: BASELINE.json configs[3] -- the SAME 31 Panoptic views split over the ranks (view v -> rank v % N, strong
    scaling), each rank renders and back-propagates its views, ONE all_gather_into_tensor (RCCL) of the per-view joint
    gradients per step rebuilds the V slots on every rank.  `value` = 31 views x steps / time; `strong_scaling` holds the
    same step run by one GPU alone (measured in the same run, all ranks side by side without communication), the speed-up,
    and the same pair for the loop's dense step and its sparse fused step; frame sharding (every rank its own frames, no
    communication) is reported as frames/s.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (forward fill + compositor): algorithmic bytes per
launch = 4*H*W*(C+1)*V_local (the dense colour + inverse-depth planes it must write) / its average launch
duration measured with hipEvents on the launch stream.  `cpu_baseline` times the pure-PyTorch restatement
(oracle/torch_ref.py) on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROOF_MIN_LAUNCHES = 32  # forward launches whose duration goes into `roofline`, at least
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves

WORKLOADS = {
    "h36m": dict(dataset="h36m", V=4, name="h36m_4view_1000x1000_P17_C17"),
    "panoptic": dict(dataset="panoptic", V=31, name="panoptic_31view_1920x1080_P19_C19"),
}
METRIC = "rendered+backpropagated views/s (differentiable skeletal-Gaussian rasterizer fwd+bwd)"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, params, n_views):
    """Pure-PyTorch CPU rasterization (fwd + autograd bwd) of the first `n_views` views of the same scene."""
    import math
    import torch
    from oracle import torch_ref
    threads = min(32, os.cpu_count() or 1)   # small ops: more threads only add sync overhead
    torch.set_num_threads(threads)
    means, feat, opac, scales, quats = [p.detach().cpu() for p in params]
    W, H = scene.W, scene.H
    g = torch.Generator().manual_seed(0)

    def one(cam):
        m = means.clone().requires_grad_(True)
        s = scales.clone().requires_grad_(True)
        q = quats.clone().requires_grad_(True)
        o = opac.clone().requires_grad_(True)
        col, _, inv = torch_ref.rasterize(m, None, feat, o, s, q, None, cam.world_view_transform.cpu(),
                                          cam.full_proj_transform.cpu(), W, H, math.tan(cam.FoVx * 0.5),
                                          math.tan(cam.FoVy * 0.5))
        dL = torch.randn(col.shape, generator=g)
        (col * dL).sum().backward()
        return m.grad

    one(scene.cameras[0])  # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < n_views or time.perf_counter() - t0 < 12.0:   # a bounded sample: ~12 s of CPU work (cap 25 s)
        one(scene.cameras[done % len(scene.cameras)])
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    res = dict(value=done / dt, unit="views/s", cores=torch.get_num_threads(), kind="port",
               host_cpus=os.cpu_count(), cpu_model=cpu_model(), threads=torch.get_num_threads(),
               sample=f"{done} views fwd+bwd of the same scene, oracle/torch_ref.py (pure PyTorch, fp32), {dt:.1f}s")
    # ... and once with every host thread (BASELINE.md section 3 planned os.cpu_count(); these are small ops: more threads
    # mostly add synchronisation, which is why the figure above uses 32)
    allc = os.cpu_count() or threads
    if allc != threads:
        torch.set_num_threads(allc)
        t1, d2 = time.perf_counter(), 0
        while d2 < 1 or (time.perf_counter() - t1 < 3.0 and d2 < 8):
            one(scene.cameras[d2 % len(scene.cameras)])
            d2 += 1
        res["value_all_host_threads"] = d2 / (time.perf_counter() - t1)
        res["all_host_threads"] = allc
        torch.set_num_threads(threads)
    return res


class ApiStep:
    """One step through the C ABI: sks_forward + sks_backward of this process's views (dL resident) and the mean of the
    per-view joint gradients over the V views (train.py:175, 215-217) -- on one GPU formed by sks_backward itself
    (dL_dmeans3D_mean).  Sharded over ranks, the mean needs ONE collective per step, in one of two forms:
      "all_reduce": every rank's sks_backward forms the mean over ITS views in the geometry backward's own launch, and one
                    RCCL all-reduce with a pre-multiplied sum (weight V_local / V) makes it the mean over all V views
                    (BASELINE configs[3]: "RCCL all-reduce on joint gradients"); needs the library's own communicator
                    (skelsplat_amd/rccl_direct.py) and is checked against the other form on the running system first;
      "all_gather": the per-view gradients land in the rank's rows of a shard, all_gather_into_tensor, then sks_mean_views sums
                    the V rows in VIEW order: bit-identical to one GPU (what MultiViewLoop does, which needs the rows)."""

    def __init__(self, views, params, dL, V_total=None, exchange=None, one_call=None):
        import torch
        from skelsplat_amd import rasterizer as R
        self.R, self.views, self.params, self.dL = R, views, params, dL
        self.exchange = exchange          # None, or (world, rank, group)
        self.mode = None
        # one_call: forward + backward through sks_forward_backward -- dL is resident, so the backward (which reads the forward's
        # geometry records, not its image) runs on a second stream beside the dense forward; False: sks_forward, then sks_backward
        self.one_call = (os.environ.get("SKS_BENCH_ONE_CALL") == "1") if one_call is None else bool(one_call)
        if exchange is not None:
            world, rank, _ = exchange
            dev, P = params[0].device, params[0].shape[0]
            vmax = (V_total + world - 1) // world
            self.shard = torch.zeros((vmax, P, 3), device=dev)      # pad rows stay zero
            self.allg = torch.zeros((world * vmax, P, 3), device=dev)
            self.allg_dst = self.allg     # (rank_step_8gpu: the rows a communicator of ONE rank fills, allg[:vmax])
            self.mean_out = torch.empty((P, 3), device=dev)
            self.local_mean = torch.zeros((P, 3), device=dev)      # (a rank without views contributes zeros with weight 0)
            self.V_total = V_total
            self.weight = (views.V if views is not None else 0) / V_total
            # RCCL's C API on the launch stream through a communicator of our own (skelsplat_amd/rccl_direct.py);
            # None (gloo test mode, SKS_RCCL_DIRECT=0, or any rank failing to build it): torch.distributed's all_gather
            from skelsplat_amd.rccl_direct import DirectGather
            self.direct = DirectGather.create(dev, exchange[2])
            self.mode = "all_gather"
        self.ws = R.Workspace()

    def choose_mode(self):
        """Collective.  The default exchange is the loop's: all_gather of the per-view rows through torch.distributed.  Switches
        to "all_reduce" only if the library's own communicator exists (opt-in: SKS_RCCL_DIRECT=1), SKS_BENCH_EXCHANGE does not
        say all_gather, and the all-reduce form reproduces the all_gather form's mean on every rank of THIS system (rtol 1e-5)."""
        import torch
        import torch.distributed as dist
        if self.exchange is None or self.direct is None or os.environ.get("SKS_BENCH_EXCHANGE", "all_reduce") != "all_reduce":
            return self.mode
        want = self().clone()
        self.mode = "all_reduce"
        # (no try / except around the collective: a rank that left it through an exception while its peers are inside
        # ncclAllReduce would turn an error into a hang; an error here ends the run on every rank instead)
        got = self()
        torch.cuda.synchronize()
        ok = bool(torch.allclose(got, want, rtol=1e-5, atol=1e-6 * float(want.abs().max())))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=want.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.exchange[2])
        if int(flag.item()) != 1:
            self.mode = "all_gather"
        return self.mode

    def fwd_bwd(self, join=True, **kw):
        """This process's views forward + backward; the gradients dict.  join=False (one call only): the gradients are ordered on
        the workspace's second stream, not yet on the current one (see __call__)."""
        R = self.R
        if self.one_call:
            return R.forward_backward_views(self.views, *self.params, None, self.dL, workspace=self.ws, join=join, **kw)[4]
        color, inv, radii, st = R.forward_views(self.views, *self.params, None, workspace=self.ws)
        return R.backward_views(st, *self.params, None, self.dL, workspace=self.ws, **kw)

    def autotune(self):
        """The forward's fill configuration for THIS step on THIS box: Workspace.tune, the library's own tuner (what a training loop
        calls once before it starts; ~100 untimed steps).  Non-temporal stores only: the cache-independent kind (rasterizer.
        TUNE_CANDIDATES; plain stores are measured separately, `same_buffer_plain_stores`).  Sharded: every rank issues the same
        number of steps whatever it tunes, so the collectives stay matched."""
        if os.environ.get("SKS_BENCH_AUTOTUNE", "1") == "0":
            return None
        cands, reps, rounds = self.cands or self.R.TUNE_CANDIDATES, 16, 3
        self()          # (the first call records the argument lists the tuner times)
        self()
        for _ in range(200):     # a fresh process: clocks and Python paths warm before anything is compared (the tuner's measurement is
            self()               # host wall time over device synchronisations)
        if self.views is None or self.ws._plans.get("fwd") is None:
            for _ in range(self.R.TUNE_WARM + len(cands) * rounds * (2 + reps)):
                self()
            return None
        # (sharded: no conditional confirmation round -- the step holds a collective, every rank issues the same number of steps)
        best, med = self.ws.tune(self, cands, reps, rounds, confirm=self.exchange is None)
        self.tuned = {"by": "Workspace.tune (skelsplat_amd/rasterizer.py)", "fill_role": self.R.tune_name(best),
                      "fill_passes_per_block": (best & 0xff) or "default (2)",
                      "stores": "plain" if best & self.R.PLAIN_STORES else "non-temporal",
                      "us_by_candidate": {self.R.tune_name(k): round(v, 2) for k, v in med.items()},
                      "rule": "each candidate's least disturbed of 3 interleaved rounds; the default stays unless another beats it by > 2 % "
                              "twice (a confirmation measurement of the two)"}
        return self.tuned

    cands = None        # tuner candidates (None: the library's default list)
    tuned = None
    wire_us = 0.0       # rank_step_8gpu only: a one-wavefront idle kernel of this length in front of the collective stands in for
                        # the xGMI hop a communicator of ONE rank does not make (sks_prof_spin)
    no_collective = False   # rank_step_8gpu only: the step without its exchange (forward + backward into the shard)

    def __call__(self):
        import torch
        import torch.distributed as dist
        R = self.R
        if self.exchange is None:
            return self.fwd_bwd(want_mean=True)["means3D_mean"]     # the mean over the views comes out of the backward's own last launch
        # Sharded.  One call: the joint gradients are ready ~40 us into the step, on the second stream, while the dense forward
        # streams for another ~70 us on the first -- the step's ONE collective (and the mean behind it) is enqueued behind the
        # backward on that second stream, so its latency hides under the forward as well; the first stream joins at the end.
        hide = self.one_call and self.views is not None
        dev = self.params[0].device
        aux = self.ws.aux_stream(dev.index) if hide else None
        reduce_mode = self.mode == "all_reduce"
        if self.views is not None:       # this rank's views; their joint gradients land in its rows of the shard
            if reduce_mode:
                local = self.fwd_bwd(join=not hide, want_mean=True)["means3D_mean"]
            else:
                self.fwd_bwd(join=not hide, out_means3D=self.shard[:self.views.V])
        else:
            local = self.local_mean
        if self.no_collective:
            if hide:
                self.ws.join(dev.index)
            return self.shard
        with torch.cuda.stream(aux) if hide else _nullcontext():
            if self.wire_us:
                from skelsplat_amd import _lib
                _lib.check(_lib.load().sks_prof_spin(float(self.wire_us), torch.cuda.current_stream(dev).cuda_stream), "sks_prof_spin")
            if reduce_mode:
                self.direct.all_reduce_weighted(self.mean_out, local, self.weight)
            else:
                if self.direct is not None:
                    self.direct.all_gather_into_tensor(self.allg_dst, self.shard)
                else:
                    dist.all_gather_into_tensor(self.allg_dst, self.shard, group=self.exchange[2])
                R.mean_views(self.allg, self.V_total, self.exchange[0], out=self.mean_out)   # reads the gathered rows in place
        if hide:
            self.ws.join(dev.index)
        return self.mean_out


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def timed(fn, steps, warmup, sync):
    """`warmup` untimed + `steps` timed calls, bracketed by `sync` (barrier + device synchronise); seconds."""
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    sync()
    return time.perf_counter() - t0, out


def prof_stride(steps):
    samples = max(5, min(25, steps // 8))
    return max(1, steps // samples)


def measure_traffic(wl_key, kernel_prefix="k_render_fwd_sparse", steps=12):
    """HBM bytes per launch of the forward kernel, MEASURED IN THIS RUN: two child processes under rocprofv3 (--pmc WRITE_SIZE, then
    --pmc FETCH_SIZE: the two do not fit one pass; --kernel-trace only beside them), each running the two-call form of the step a
    dozen times (tools/one_call_step.py).  Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters in KiB,
    FETCH_SIZE doubled.  None when rocprofv3 is not on PATH or a pass fails (the caller then quotes profiles/traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None or os.environ.get("SKS_BENCH_TRAFFIC", "1") == "0":
        return None
    # bench.py itself under a profiler (rocprofv3 -- python3 bench.py ...): the children must not inherit its preloaded tool library
    # -- a second profiler inside the first, and every hop of the inner launcher an exec from a GPU-initialised process.  No
    # measurement then (the caller quotes profiles/traffic.json); profile_round.sh passes --no-extras / SKS_BENCH_TRAFFIC=0 anyway.
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="sks_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", WL=wl_key, ONE_CALL="0", STEPS=str(steps))
    vals = {}
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--",
                   "python3", os.path.join(ROOT, "tools", "one_call_step.py")]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            if r.returncode != 0:
                return None
            per = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and kernel_prefix in row["Kernel_Name"]:
                        per.append(float(row["Counter_Value"]))
            if len(per) < 4:
                return None
            per = per[len(per) // 3:]          # (the first launches allocate and warm up)
            vals[counter] = sum(per) / len(per)
        return 1024.0 * (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"])
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_entry(alg_bytes, prof_fwd, wl_name, launches, kernel="k_render_fwd_sparse (forward fill + sparse compositor)",
                   traffic_scale=1.0):
    """`traffic_scale`: a sharded rank launches the kernel for its share of the workload's views; the PMC figure of the
    whole-workload launch scales with the views (the kernel writes every plane once)."""
    fwd_ms, fwd_n, fwd_q = prof_fwd
    avg_s = fwd_ms * 1e-3 / fwd_n
    traffic, src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(wl_name, {}).get("fwd_bytes_per_launch")
            src = "profiles/traffic.json (rocprofv3 --pmc passes of this command, collected separately: not measured in this run)"
            if traffic is not None and traffic_scale != 1.0:
                traffic *= traffic_scale
                src += f"; the whole workload's launch x {traffic_scale:.4f} (this rank's share of the views)"
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src if traffic else None,
            "frac_median": alg_bytes / (fwd_q[1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "avg_launch_us": avg_s * 1e6, "launch_us_p10_p50_p90": [round(1e3 * x, 2) for x in fwd_q],
            "launches_timed": fwd_n, "launches": launches, "algorithmic_bytes_per_launch": alg_bytes}


def zero_fill_us(torch, n_floats, dev, sync, rotate=1):
    """The box's own ceiling for a forward's bytes, in the same run: median duration of tensor.zero_() of a buffer that size.
    rotate > 1: over that many buffers in turn -- the same buffer again and again lets the 256 MB Infinity Cache hold back part of
    the stores (288 MB: 41 us); on memory the kernel has not just written (8 buffers = 2.3 GB) the same call takes 45-46 us: the
    rate HBM itself takes."""
    try:
        zbufs = [torch.empty(int(n_floats), device=dev) for _ in range(rotate)]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(12, 3 * rotate))]
        for i, (a_, b_) in enumerate(evs):
            a_.record()
            zbufs[i % rotate].zero_()
            b_.record()
        sync()
        zt = sorted(a_.elapsed_time(b_) for a_, b_ in evs[max(2, rotate):])
        return 1e3 * zt[len(zt) // 2]
    except Exception:
        return None


def make_scene(torch, wl, dev, seed=0):
    """Synthetic scene + Gaussians + activated parameter tuple for the C ABI."""
    from skelsplat_amd.scene import SyntheticScene, GaussianModel
    scene = SyntheticScene(wl["dataset"], n_views=wl["V"], seed=seed, device=dev)
    gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints,
                                            scene_type=wl["dataset"], device=dev)
    P, C = scene.n_points, scene.n_joints
    with torch.no_grad():
        params = (gm.get_xyz.detach().clone(), gm.get_features.reshape(P, C).contiguous(), gm.get_opacity.detach().clone(),
                  gm.get_scaling.detach().clone(), gm.get_rotation.detach().clone())
    return scene, gm, params


def fresh_model(scene, dataset, dev):
    from skelsplat_amd.scene import GaussianModel
    gm = GaussianModel().create_from_points(scene.pose_3d_init, scene.spatial_lr_scale, scene.n_joints, scene_type=dataset, device=dev)
    gm.training_setup()
    return gm


def loop_ms(torch, loop, n, sync):
    """ms per accumulation group of a loop that does nothing but step: consecutive groups chained as loop.run() chains them (the
    fused step's tail leaves the next group's geometry behind; a step_group() on its own would recompute it from the parameters)"""
    kw = {"parameters_untouched": True} if "parameters_untouched" in loop.step_group.__code__.co_varnames else {}
    loop.step_group()
    for _ in range(2):
        loop.step_group(**kw)
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        loop.step_group(**kw)
    sync()
    return 1e3 * (time.perf_counter() - t0) / n


# ------------------------------------------------------------------------------------------------------------
# N = 1: BASELINE configs[1] + extras
# ------------------------------------------------------------------------------------------------------------
def run_single(args, torch, dev, wl):
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.scene import SyntheticScene

    def sync():
        torch.cuda.synchronize()

    V = wl["V"]
    scene, gm, params = make_scene(torch, wl, dev)
    W, H, P, C = scene.W, scene.H, scene.n_points, scene.n_joints
    views = R.ViewBatch.from_cameras(scene.cameras)
    dL = torch.randn((V, C, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    # Two forms of the same step.  "two" (the headline): sks_forward, then sks_backward on the same stream -- what a caller whose
    # upstream gradient is a function of the image just rendered can use (train.py:141-161: render -> loss -> backward).  "one":
    # sks_forward_backward, the backward on a second stream BESIDE the forward -- legal only because this benchmark's dL is a
    # resident tensor that does not depend on the image (SURVEY section 8d's synthetic dense dL); reported beside the headline,
    # labelled.  --form one / two restricts the run to one of them (a rocprofv3 --kernel-trace --stats run per form gives
    # per-kernel averages that are not a mixture); the default measures both.
    forms = ("two", "one") if args.form == "both" else (args.form,)
    prof = not args.no_prof
    stride = int(os.environ.get("SKS_PROF_EVERY", "0")) or prof_stride(args.steps)

    def measure(one_call):
        """W warm-up + K timed steps of one form; the forward compositor launches of every n-th TIMED step carry a hipEvent pair
        (hipExtLaunchKernelGGL: stamped from the kernel's own dispatch, on the launch stream -- a bracketed step costs ~13 us of
        queue time, so the sample is 5 .. 25 of the K launches), topped up to ROOF_MIN_LAUNCHES behind the timed region by
        untimed steps of the SAME form with every launch bracketed."""
        st = ApiStep(views, params, dL, one_call=one_call)
        st.autotune()                                      # (untimed: Workspace.tune -- the fill configuration for this form on this box)
        for _ in range(args.warmup):
            st()
        sync()
        if prof:
            _lib.prof_enable(True, every=stride, kinds=(0,))
            _lib.prof_read(0), _lib.prof_read(1)
        dt_, out_ = timed(st, args.steps, 0, sync)         # THE timed region: exactly K steps
        pf_ = None
        if prof:
            have = _lib.prof_count(0)
            if have < ROOF_MIN_LAUNCHES:
                _lib.prof_enable(True, every=1, kinds=(0,))
                for _ in range(ROOF_MIN_LAUNCHES - have):
                    st()
                sync()
            pf_ = _lib.prof_read_quantiles(0)
            _lib.prof_enable(False)
        return st, dt_, out_, pf_

    def rotating(n_sets=8, cycles=5):
        """The forward kernel over `n_sets` OUTPUT SETS in turn (8 x 288 MB = 2.3 GB: memory the kernel has not just written, many
        times the 256 MB Infinity Cache): the two-call step through n_sets workspaces, every forward launch bracketed.  This is the
        HBM figure: nothing of a launch's output can still sit in the cache from the launch before."""
        sts = [ApiStep(views, params, dL, one_call=False) for _ in range(n_sets)]
        cfg_flags = step.ws._plans["fwd"][2][16]
        for st in sts:          # the headline step's own fill configuration (what its Workspace.tune picked), flag for flag
            st(), st()
            st.ws._plans["fwd"][2][16] = cfg_flags
        sync()
        _lib.prof_enable(True, every=1, kinds=(0,))
        _lib.prof_read(0)
        for _ in range(cycles):
            for st in sts:
                st()
        sync()
        pf_ = _lib.prof_read_quantiles(0)
        _lib.prof_enable(False)
        cfg = cfg_flags
        del sts
        torch.cuda.empty_cache()
        return pf_, cfg

    step, dt, out, pf = measure(forms[0] == "one")
    step2 = dt2 = pf2 = None
    if len(forms) == 2:
        step2, dt2, out2, pf2 = measure(True)
        assert torch.equal(out, out2)                      # (bit for bit the same gradients either way)
    pb = None
    zero_us = zero_fresh_us = pf_rot = None
    if prof:
        zero_us = zero_fill_us(torch, V * (C + 1) * H * W, dev, sync)
        nbytes_out = 4.0 * V * (C + 1) * H * W
        zero_fresh_us = zero_fill_us(torch, V * (C + 1) * H * W, dev, sync, rotate=8) if nbytes_out < 1.2e9 else None
        if nbytes_out < 1.2e9:
            pf_rot, rot_flags = rotating()
        _lib.prof_enable(True, every=1, kinds=(1,))     # the backward compositor: a few untimed steps behind the timed regions
        _lib.prof_read(1)
        for _ in range(10):
            step()
        sync()
        pb = _lib.prof_read_quantiles(1)
        _lib.prof_enable(False)
    assert torch.isfinite(out).all()
    PATHS = {
        False: "C ABI sks_forward, then sks_backward (incl. the mean over the views) on one stream -- the form a caller whose upstream "
               "gradient depends on the rendered image can use (train.py:141-161); eager launches, outputs in a reused workspace, fill "
               "configuration from Workspace.tune",
        True: "C ABI sks_forward_backward: forward + backward (incl. the mean over the views) as one call -- the backward reads the "
              "forward's geometry records, not its image, and runs on a second stream BESIDE the dense forward.  Needs an upstream "
              "gradient that is complete before the forward starts (this benchmark's resident dL; NOT a loss of the image being "
              "rendered): no train.py-shaped caller can use it; eager launches, outputs in a reused workspace"}
    res = {
        "metric": METRIC, "value": V * args.steps / dt, "unit": "views/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "views_per_step": V, "P": P, "C": C, "W": W, "H": H, "parallelism": "single GPU",
                   "form": "one call" if step.one_call else "two calls", "path": PATHS[step.one_call]},
    }
    res["config"]["autotuned"] = step.tuned
    alg_bytes = 4.0 * H * W * (C + 1) * V
    if dt2 is not None:
        # the same step through sks_forward_backward, K steps timed the same way: an extra, not the headline (see PATHS)
        res["one_call_step"] = {"ms_per_step": 1e3 * dt2 / args.steps, "views_per_s": V * args.steps / dt2, "autotuned": step2.tuned,
                                "upstream_gradient": "independent of the image (resident dL)", "path": PATHS[True]}
    if pf and pf[1]:
        same = roofline_entry(alg_bytes, pf, wl["name"], args.steps)
        if pf_rot and pf_rot[1]:
            # THE roofline figure: the kernel over 8 output sets in turn -- an HBM rate.  The same kernel inside the timed region
            # rewrites ONE set of buffers: with non-temporal stores the same rate (they bypass the cache), reported beside it
            res["roofline"] = roofline_entry(alg_bytes, pf_rot, wl["name"], pf_rot[1])
            res["roofline"]["timed_in"] = ("untimed steps of the two-call form behind the timed region, cycling through 8 workspaces "
                                           "(8 output sets = 2.3 GB in turn), every forward launch bracketed by a hipEvent pair on its "
                                           "own dispatch")
            res["roofline"]["rotating_output_sets"] = 8
            res["roofline"]["stores"] = "plain" if rot_flags & 16 else "non-temporal"
            res["roofline"]["frac_same_buffer"] = same["frac"]
            res["roofline"]["same_buffer"] = {
                "what": f"the same kernel inside the timed region of this line (the {'one' if step.one_call else 'two'}-call form): one set "
                        "of output buffers rewritten every step", "bound": "hbm (non-temporal stores bypass the Infinity Cache)"
                        if not (rot_flags & 16) else "infinity-cache-assisted (plain stores into a buffer rewritten every step)",
                "avg_launch_us": same["avg_launch_us"], "launch_us_p10_p50_p90": same["launch_us_p10_p50_p90"],
                "launches_timed": same["launches_timed"], "frac": same["frac"], "frac_median": same["frac_median"]}
        else:
            res["roofline"] = same
            res["roofline"]["timed_in"] = f"the timed region of this line (the {'one' if step.one_call else 'two'}-call form)"
        res["roofline"]["whole_step_frac"] = alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS   # the step's mandatory bytes over its time
        if pf2 and pf2[1]:
            a_s = pf2[0] * 1e-3 / pf2[1]
            res["roofline"]["kernel_beside_backward"] = {
                "what": "the same kernel in the one-call form of the step (the backward runs beside it on a second queue and takes wave "
                        "slots from its fill blocks), same run, same event pairs, one set of output buffers",
                "avg_launch_us": a_s * 1e6, "launch_us_p10_p50_p90": [round(1e3 * x, 2) for x in pf2[2]], "launches_timed": pf2[1],
                "achieved": alg_bytes / a_s / 1e9, "frac": alg_bytes / a_s / 1e9 / HBM_PEAK_GBS,
                "frac_median": alg_bytes / (pf2[2][1] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if not args.no_extras:
            tb = measure_traffic(wl["dataset"])
            if tb:
                res["roofline"]["traffic_file"] = res["roofline"]["traffic"]
                res["roofline"]["traffic"] = tb
                res["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate child "
                                                     "passes over tools/one_call_step.py, KiB units, FETCH_SIZE doubled for gfx950)")
                res["roofline"]["traffic_over_algorithmic"] = tb / res["roofline"]["algorithmic_bytes_per_launch"]
        if zero_us and zero_fresh_us:
            # what HBM itself takes: zero_() over 8 buffers in turn (2.3 GB), no help from the Infinity Cache
            res["roofline"]["zero_fill_fresh_memory_us"] = zero_fresh_us
            res["roofline"]["hbm_write_rate_by_zero_fill_GBps"] = nbytes_out / zero_fresh_us / 1e3
            res["roofline"]["frac_of_zero_fill_fresh_memory"] = zero_fresh_us / res["roofline"]["launch_us_p10_p50_p90"][1]
        if zero_us:
            res["roofline"]["zero_fill_same_buffer_us"] = zero_us     # (plain stores into one buffer: the cache holds part of it back)
        if pb and pb[1]:
            res["bwd_kernel_avg_us"] = pb[0] * 1e3 / pb[1]
            res["bwd_kernel_us_p10_p50_p90"] = [round(1e3 * x, 2) for x in pb[2]]
        if not args.no_extras and nbytes_out < 1.2e9:
            # what round 5 reported as the headline: plain stores into the one buffer set the step rewrites -- cache-assisted
            try:
                sp = ApiStep(views, params, dL, one_call=False)
                sp.cands = R.TUNE_CANDIDATES_WITH_PLAIN
                sp.autotune()
                _lib.prof_enable(True, every=4, kinds=(0,))
                _lib.prof_read(0)
                dtp, _ = timed(sp, max(40, args.steps // 2), 5, sync)
                pfp = _lib.prof_read_quantiles(0)
                _lib.prof_enable(False)
                res["same_buffer_plain_stores"] = {
                    "what": "the two-call step with plain-store candidates in the tuner's list (rasterizer.TUNE_CANDIDATES_WITH_PLAIN): "
                            "where it picks them the forward retires before its bytes are in HBM -- an Infinity-Cache-assisted "
                            "duration, not an HBM rate",
                    "bound": "infinity-cache-assisted", "autotuned": sp.tuned, "ms_per_step": 1e3 * dtp / max(40, args.steps // 2),
                    "fwd_kernel_us": (pfp[0] * 1e3 / pfp[1]) if pfp[1] else None,
                    "fwd_bytes_over_duration_over_8TBps": (alg_bytes / (pfp[0] * 1e-3 / pfp[1]) / 1e9 / HBM_PEAK_GBS) if pfp[1] else None}
                del sp
            except Exception as e:
                res["same_buffer_plain_stores"] = {"error": repr(e)[:200]}
    if args.no_extras:
        return res, scene, params

    extras = {}
    # ---- the same step replayed as a hipGraph --------------------------------------------------------------------
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(5):
            graph.replay()
        tg, _ = timed(graph.replay, args.steps, 0, sync)
        extras["graph_replay_views_per_s"] = V * args.steps / tg
        extras["graph_replay_ms_per_step"] = 1e3 * tg / args.steps
    except Exception as e:  # capture is an optimisation, never a requirement
        extras["graph_replay_error"] = repr(e)[:200]
    # ---- the full loop step: render + masked-L2 + backward + Adam (train.py:130-222) -----------------------------
    try:
        from skelsplat_amd.loop import MultiViewLoop
        from skelsplat_amd.heatmaps import generate_heatmaps
        gm.training_setup()
        p2d = torch.tensor(scene.poses_2d, device=dev)
        hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, scene.cameras)
        nl = max(10, args.steps // 2)
        for tag, ug in (("", False), ("_hipgraph", True)):
            loop = MultiViewLoop(gm, scene.cameras, hm, dataset=wl["dataset"], accumulation_steps=V, use_graph=ug)
            ms = loop_ms(torch, loop, nl, sync)
            extras["grad_step_ms" + tag] = ms
            extras["loop_views_per_s" + tag] = V / ms * 1e3
        # one whole scene like configs/h36m.yaml (500 iterations), accumulation groups captured 25 per hipGraph
        loop = MultiViewLoop(gm, scene.cameras, hm, dataset=wl["dataset"], accumulation_steps=V, use_graph=True)
        loop.run(500)                      # captures
        sync()
        ts = time.perf_counter()
        for _ in range(3):
            loop.iteration = 0             # parameters keep evolving; the work per iteration does not change
            loop.run(500)
        sync()
        ts = (time.perf_counter() - ts) / 3
        extras["scene_500it_ms_hipgraph"] = 1e3 * ts
        extras["grad_step_ms_scene_hipgraph"] = 1e3 * ts / (500 / V)
        # frames streamed through one loop object (train.py:74-99 per frame: re-initialise the Gaussians, generate the
        # heat-maps from the 2D detections, 500 iterations), everything in place so the hipGraphs are reused
        pts = torch.tensor(scene.pose_3d_init, device=dev, dtype=torch.float32)
        loop.new_scene(pts, poses_2d=p2d)
        loop.run(500)
        sync()
        tf = time.perf_counter()
        for _ in range(5):
            loop.new_scene(pts, poses_2d=p2d)
            loop.run(500)
        sync()
        extras["frame_stream_ms"] = 1e3 * (time.perf_counter() - tf) / 5
        del loop, hm
        torch.cuda.empty_cache()
        # the same streamed frames, F at a time through the same two launches per group (loop.FrameBatchLoop: frames
        # are independent, train.py:74-99; each one's trajectory is bit-identical to running it alone)
        from skelsplat_amd.loop import FrameBatchLoop
        F = max(1, min(16, 64 // V))
        fb = FrameBatchLoop(fresh_model(scene, wl["dataset"], dev), scene.cameras, F, dataset=wl["dataset"],
                            accumulation_steps=V, use_graph=True)
        ptsF, p2dF = pts[None].repeat(F, 1, 1), p2d[None].repeat(F, 1, 1, 1)
        fb.new_scenes(ptsF, poses_2d=p2dF)
        fb.run(500)
        sync()
        tf = time.perf_counter()
        for _ in range(3):
            fb.new_scenes(ptsF, poses_2d=p2dF)
            fb.run(500)
        sync()
        tb = (time.perf_counter() - tf) / 3
        extras["frame_batch"] = {"frames_per_launch": F, "batch_ms": 1e3 * tb, "frames_per_s": F / tb,
                                 "one_at_a_time_frames_per_s": 1e3 / extras["frame_stream_ms"]}
        del fb
        torch.cuda.empty_cache()
        # ... and several such batches in flight on separate HIP streams (loop.FramePipeline): one batch's single-workgroup
        # step tails run under the others' backward kernels
        from skelsplat_amd.loop import FramePipeline
        S = 4
        pipe = FramePipeline(fresh_model(scene, wl["dataset"], dev), scene.cameras, frames=F, streams=S, dataset=wl["dataset"],
                             accumulation_steps=V)
        ptsN, p2dN = pts[None].repeat(S * F, 1, 1), p2d[None].repeat(S * F, 1, 1, 1)
        pipe.optimize_sequence(ptsN, p2dN, iterations=500)
        sync()
        tf = time.perf_counter()
        for _ in range(2):
            pipe.optimize_sequence(ptsN, p2dN, iterations=500)
        sync()
        tp = (time.perf_counter() - tf) / 2
        extras["frame_batch"].update({"pipeline_streams": S, "pipeline_frames_per_s": S * F / tp})
        del pipe
        torch.cuda.empty_cache()
    except Exception as e:
        extras["loop_error"] = repr(e)[:300]
    # ---- the reference's own iteration with only the modules swapped (train.py:130-161) --------------------------
    if not args.no_dropin:
        try:
            extras.update(dropin_iteration(args, torch, dev, wl, scene))
        except Exception as e:
            extras["dropin_error"] = repr(e)[:200]
    # ---- other BASELINE configs, as extras of the same line ------------------------------------------------------
    if wl["dataset"] == "h36m":
        for name, fn in (("h36m_mixed_1002", extra_mixed), ("panoptic", extra_panoptic), ("stress", extra_stress),
                         ("rank_step_8gpu", extra_rank_step)):
            try:
                extras[name] = fn(args, torch, dev, sync)
            except Exception as e:
                extras[name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
    res.update(extras)
    return res, scene, params


def dropin_iteration(args, torch, dev, wl, scene):
    """render() of ONE view through the drop-in gaussian_renderer, masked-L2, autograd backward, Adam every V views."""
    import types
    from gaussian_renderer import render_functions
    from skelsplat_amd.loop import l2_loss_gaussian
    from skelsplat_amd.ops import l2_loss_gaussian as l2_loss_gaussian_fused
    from skelsplat_amd.heatmaps import generate_heatmaps
    V = wl["V"]
    render = render_functions["diff-gaussian-rasterization-" + wl["dataset"]]
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, convert_SHs_python=False)
    bgc = torch.zeros(3, device=dev)
    gm2 = fresh_model(scene, wl["dataset"], dev)
    hm = generate_heatmaps(gm2._xyz.detach(), gm2.get_scaling.detach(), gm2._rotation.detach(),
                           torch.tensor(scene.poses_2d, device=dev), scene.cameras)

    def one_launch_adam():
        # the same groups and state under skelsplat_amd.optim.Adam (a torch.optim.Adam whose step() is one launch): the one-line
        # change to scene/gaussian_model.py:218 that INTEGRATION.md names
        from skelsplat_amd.optim import Adam
        groups = [{k: g[k] for k in ("params", "lr", "name")} for g in gm2.optimizer.param_groups]
        opt = Adam(groups, lr=0.0, eps=1e-15)
        opt.load_state_dict(gm2.optimizer.state_dict())
        gm2.optimizer = opt

    def it(i, criterion):
        pkg = render(scene.cameras[i % V], gm2, pipe, bgc)
        loss, _ = criterion(pkg["render"], hm[i % V])   # (loss, error image) like loss_utils.py:100
        loss.backward()
        if (i + 1) % V == 0:
            gm2.optimizer.step()
            gm2.optimizer.zero_grad(set_to_none=True)

    out = {}
    # tensor-op criterion as in the reference, then the fused criterion registered in its `losses` table
    for tag, crit in (("dropin_iteration_ms", l2_loss_gaussian), ("dropin_iteration_fused_loss_ms", l2_loss_gaussian_fused),
                      ("dropin_iteration_fused_loss_one_launch_adam_ms", l2_loss_gaussian_fused)):
        if "one_launch_adam" in tag:
            one_launch_adam()
        for i in range(2 * V):
            it(i, crit)
        # host-bound (a few dozen Python-level launches per view): the median of 3 repetitions of >= 16 accumulation groups --
        # a dozen iterations, as this used to time, read 0.38-0.83 ms for the same code on different boxes of the pool
        nd, reps = max(16 * V, args.steps // 4), []
        for _ in range(3):
            torch.cuda.synchronize()
            td = time.perf_counter()
            for i in range(nd):
                it(i, crit)
            torch.cuda.synchronize()
            reps.append(1e3 * (time.perf_counter() - td) / nd)
        out[tag] = sorted(reps)[1]
        out[tag.replace("_ms", "_ms_reps")] = [round(r, 4) for r in reps]
    return out


def extra_mixed(args, torch, dev, sync):
    """The real H36M sensor mix (scene/dataset_readers.py:68-80: 1002- and 1000-wide cameras in every subject): the API
    step as two size groups (the dense tensors cannot mix sizes), the sparse loop group as ONE launch sequence."""
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.scene import SyntheticScene
    from skelsplat_amd.loop import MultiViewLoop
    from skelsplat_amd.heatmaps import generate_heatmaps
    a = SyntheticScene("h36m", n_views=4, seed=0, device=dev, W=1000, H=1000)
    b = SyntheticScene("h36m", n_views=4, seed=0, device=dev, W=1002, H=1000)
    cams = [b.cameras[0], a.cameras[1], a.cameras[2], b.cameras[3]]          # widths 1002, 1000, 1000, 1002
    _, gm, params = make_scene(torch, WORKLOADS["h36m"], dev)
    C = 17
    out = {}
    # all four views 1002 wide through the API: the 16-byte fill with half-masked float4s vs the 1000-wide headline
    v1002 = R.ViewBatch.from_cameras(b.cameras)
    dLb = torch.randn((4, C, 1000, 1002), device=dev)
    step = ApiStep(v1002, params, dLb)
    n = max(20, args.steps // 2)
    _lib.prof_enable(True, every=4)
    _lib.prof_read(0)
    dt, _ = timed(step, n, 5, sync)
    pf = _lib.prof_read_quantiles(0)
    _lib.prof_enable(False)
    out["api_4x1002_ms_per_step"] = 1e3 * dt / n
    out["api_4x1002_views_per_s"] = 4 * n / dt
    if pf[1]:
        out["fwd_kernel_us_4x1002"] = pf[0] * 1e3 / pf[1]
        out["fwd_frac_of_hbm_peak_4x1002"] = 4.0 * 1000 * 1002 * (C + 1) * 4 / (pf[0] * 1e-3 / pf[1]) / 1e9 / HBM_PEAK_GBS
    del step, dLb
    # the mix: two API calls of two views each
    groups = []
    for sc, ids in ((b, [0, 3]), (a, [1, 2])):
        vb = R.ViewBatch.from_cameras([sc.cameras[i] for i in ids])
        groups.append(ApiStep(vb, params, torch.randn((2, C, sc.H, sc.W), device=dev)))

    def mixed_step():
        return groups[0]() + groups[1]()
    dt, _ = timed(mixed_step, n, 5, sync)
    out["api_mixed_ms_per_step"] = 1e3 * dt / n
    out["api_mixed_views_per_s"] = 4 * n / dt
    del groups
    # the loop: one sparse fused group over all four views
    gm.training_setup()
    hms = [generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                             torch.tensor((b if cams[v].image_width == 1002 else a).poses_2d[v:v + 1], device=dev), [cams[v]])[0]
           for v in range(4)]
    for tag, ug in (("", False), ("_hipgraph", True)):
        loop = MultiViewLoop(gm, cams, hms, dataset="h36m", accumulation_steps=4, use_graph=ug)
        out["loop_mixed_grad_step_ms" + tag] = loop_ms(torch, loop, n, sync)
        out["loop_mixed_launch_sequences"] = 1 if loop.views_all is not None and loop.views_all.mixed else len(loop.size_groups)
    return out


def extra_panoptic(args, torch, dev, sync):
    """BASELINE configs[2]: Panoptic, 19 joints, 31 HD views, one GPU."""
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.loop import MultiViewLoop
    from skelsplat_amd.heatmaps import generate_heatmaps
    wl = WORKLOADS["panoptic"]
    scene, gm, params = make_scene(torch, wl, dev)
    V, W, H, C = wl["V"], scene.W, scene.H, scene.n_joints
    views = R.ViewBatch.from_cameras(scene.cameras)
    dL = torch.randn((V, C, H, W), device=dev)
    step = ApiStep(views, params, dL, one_call=False)     # the headline form: sks_forward, then sks_backward
    step1 = ApiStep(views, params, dL, one_call=True)     # beside it: one call (the backward on a second stream under the forward)
    step.autotune(), step1.autotune()
    n = max(10, args.steps // 10)
    reps = {"two": [], "one": []}
    for _ in range(3):       # interleaved: both forms see the same box state
        for tag, fn in (("two", step), ("one", step1)):
            dtr, _ = timed(fn, n, 2, sync)          # the step itself: no event brackets inside the timed region
            reps[tag].append(1e3 * dtr / n)
    ms2, ms1 = sorted(reps["two"])[1], sorted(reps["one"])[1]
    _lib.prof_enable(True, every=1)                 # (marker-packet events on this path: ~3 us of queue time per bracket)
    _lib.prof_read(0), _lib.prof_read(1)
    timed(step, n, 0, sync)                         # kernel durations from the two-call form (the forward alone on the chip)
    pf, pb = _lib.prof_read_quantiles(0), _lib.prof_read_quantiles(1)
    _lib.prof_enable(False)
    out = {"workload": wl["name"], "form": "two calls", "ms_per_step": ms2, "views_per_s": V / ms2 * 1e3, "autotuned": step.tuned,
           "ms_per_step_reps": {k: [round(x, 4) for x in v] for k, v in reps.items()},
           "one_call_step": {"ms_per_step": ms1, "views_per_s": V / ms1 * 1e3, "autotuned": step1.tuned,
                             "upstream_gradient": "independent of the image (resident dL)"},
           # the step's form per workload: what a caller with a resident gradient should take here
           "faster_form": "one call" if ms1 < ms2 else "two calls"}
    if pf[1]:
        alg = 4.0 * H * W * (C + 1) * V
        out["kernel_durations_from"] = "the two-call form of the step (the forward alone on the chip)"
        out["roofline_note"] = ("5.1 GB per launch, 20 x the Infinity Cache: the same-buffer duration IS the HBM figure (non-temporal "
                                "stores)")
        out["fwd_kernel_us"] = pf[0] * 1e3 / pf[1]
        out["fwd_kernel_us_p10_p50_p90"] = [round(1e3 * x, 1) for x in pf[2]]
        out["fwd_frac_of_hbm_peak"] = alg / (pf[0] * 1e-3 / pf[1]) / 1e9 / HBM_PEAK_GBS
        zus = zero_fill_us(torch, V * (C + 1) * H * W, dev, sync)
        if zus:
            out["zero_fill_same_bytes_us"] = zus
            out["frac_of_zero_fill"] = zus / (1e3 * pf[2][1])
    if pb[1]:
        out["bwd_kernel_us"] = pb[0] * 1e3 / pb[1]
    if not args.no_extras and pf[1]:
        tb = measure_traffic("panoptic", steps=6)
        if tb:
            out["fwd_traffic_bytes"] = tb
            out["fwd_traffic_over_algorithmic"] = tb / (4.0 * H * W * (C + 1) * V)
            out["traffic_source"] = "measured in this run (child rocprofv3 --pmc passes, as for the headline)"
    del step1
    del step, dL
    gm.training_setup()
    hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(),
                           torch.tensor(scene.poses_2d, device=dev), scene.cameras)
    for tag, kw in (("loop_sparse_grad_step_ms", dict(use_graph=False)), ("loop_sparse_grad_step_ms_hipgraph", dict(use_graph=True)),
                    ("loop_dense_grad_step_ms", dict(sparse=False))):
        loop = MultiViewLoop(gm, scene.cameras, hm, dataset="panoptic", accumulation_steps=V, **kw)
        out[tag] = loop_ms(torch, loop, n, sync)
        del loop
    return out


def extra_stress(args, torch, dev, sync):
    """BASELINE configs[4]: 256 skeletons (P = 4352, C = 17), 8 views @ 2048x2048, binned path."""
    import numpy as np
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.scene import stress_scene
    V, C, H, W = 8, 17, 2048, 2048
    sc, g = stress_scene(V, W=W, H=H)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    views = R.ViewBatch.from_cameras([cam.to(dev) for cam in sc.cameras])
    params = (t(g["means"]), t(g["feat"]), t(g["opac"]), t(g["scales"]), t(g["quats"]))
    dL = torch.randn((V, C, H, W), device=dev)
    ws = R.Workspace()

    wsd = R.Workspace()

    def step():      # the arena checked lazily: counts in pinned host memory, looked at when the next call comes in
        color, inv, radii, st = R.forward_views(views, *params, None, bin_capacity=400000, workspace=ws, check_capacity="auto")
        return R.backward_views(st, *params, None, dL, workspace=ws)["means3D"]

    def step_default():     # the library's default: the pair count of EVERY forward is checked before the call returns
        color, inv, radii, st = R.forward_views(views, *params, None, bin_capacity=400000, workspace=wsd)
        return R.backward_views(st, *params, None, dL, workspace=wsd)["means3D"]
    wso = R.Workspace()

    def step_one_call():    # forward + backward as ONE call: the library pipelines view groups over two streams (SKS_BIN_GROUPS) --
        # the VALU-bound tile backward of group g runs beside the HBM-bound forward of group g + 1.  Needs the resident dL.
        return R.forward_backward_views(views, *params, None, dL, workspace=wso, bin_capacity=400000)[4]["means3D"]
    n = max(10, args.steps // 10)
    for fn in (step, step_default, step_one_call):
        for _ in range(3):
            fn()
    assert torch.equal(step_one_call(), step_default())     # bit for bit the two calls' gradients
    reps = {"auto": [], "default": [], "one_call": []}
    for _ in range(5):       # interleaved: the modes see the same box state
        for tag, fn in (("auto", step), ("default", step_default), ("one_call", step_one_call)):
            dtr, _ = timed(fn, n, 0, sync)
            reps[tag].append(1e3 * dtr / n)
    med = {k: sorted(v)[len(v) // 2] for k, v in reps.items()}
    _lib.prof_enable(True, every=1)
    _lib.prof_read(0), _lib.prof_read(1)
    timed(step, n, 0, sync)
    pf, pb = _lib.prof_read_quantiles(0), _lib.prof_read_quantiles(1)
    _lib.prof_enable(False)
    out = {"workload": "stress_256skeletons_8view_2048x2048_P4352_C17", "ms_per_step": med["default"], "views_per_s": V / med["default"] * 1e3,
           "mode": "check_capacity=True (the library's default: every forward's pair count is checked before the call returns)",
           "form": "two calls",
           "ms_per_step_check_capacity_auto": med["auto"], "default_over_auto": med["default"] / med["auto"],
           "one_call_step": {"ms_per_step": med["one_call"], "views_per_s": V / med["one_call"] * 1e3,
                             "whole_step_frac_of_hbm_peak": 4.0 * H * W * (C + 1) * V / (med["one_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "upstream_gradient": "independent of the image (resident dL)",
                             "what": "sks_forward_backward on the binned path: the views as view groups (SKS_BIN_GROUPS), group g's tile "
                                     "backward on a second stream beside group g + 1's forward; check_capacity=True like the default mode"},
           "whole_step_frac_of_hbm_peak": 4.0 * H * W * (C + 1) * V / (med["default"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "ms_per_step_reps": {k: [round(x, 4) for x in v] for k, v in reps.items()}}
    if pf[1]:
        out["fwd_kernel_us"] = pf[0] * 1e3 / pf[1]
        out["fwd_kernel_us_p10_p50_p90"] = [round(1e3 * x, 1) for x in pf[2]]
        out["fwd_frac_of_hbm_peak"] = 4.0 * H * W * (C + 1) * V / (pf[0] * 1e-3 / pf[1]) / 1e9 / HBM_PEAK_GBS
        zus = zero_fill_us(torch, V * (C + 1) * H * W, dev, sync)
        if zus:
            out["zero_fill_same_bytes_us"] = zus
            out["frac_of_zero_fill"] = zus / (1e3 * pf[2][1])
    if pb[1]:
        out["bwd_kernel_us"] = pb[0] * 1e3 / pb[1]
        out["bwd_kernel_us_p10_p50_p90"] = [round(1e3 * x, 1) for x in pb[2]]
    # byte models and PMC traffic (profiles/traffic.json: separate rocprofv3 --pmc passes over tools/bench_stress.py, the same
    # scene).  Forward: every plane is written once.  Backward: dL/d(colour, inverse depth) is read where a tile's list is not
    # empty -- at most (C+1) planes of the covered tiles; channels no entry of the list has a feature for are skipped.
    st = R.forward_views(views, *params, None, bin_capacity=400000, check_capacity="auto")[3]
    pl, rg, nr = R.export_lists(st)
    covered = int((rg[..., 1] > rg[..., 0]).sum())
    out["covered_tiles"] = covered
    out["pairs_per_view"] = [int(x) for x in nr.cpu()]
    out["fwd_algorithmic_bytes"] = 4.0 * H * W * (C + 1) * V
    out["bwd_model_bytes_upper"] = 4.0 * 256 * (C + 1) * covered
    if not args.no_extras:
        tb = measure_traffic("stress", kernel_prefix="k_render_fwd_binned", steps=6)
        if tb:
            out["fwd_traffic_bytes"] = tb
            out["fwd_traffic_over_algorithmic_in_run"] = tb / out["fwd_algorithmic_bytes"]
            out["fwd_traffic_source"] = "measured in this run (child rocprofv3 --pmc passes, as for the headline)"
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["stress_256skeletons_8view_2048x2048_P4352_C17"]["kernels"]
        f = [v for k, v in tr.items() if k.startswith("k_render_fwd_binned")][0]
        b = [v for k, v in tr.items() if k.startswith("k_render_bwd_tile") or k.startswith("k_render_bwd_binned")][0]
        out["fwd_traffic_over_algorithmic"] = f["hbm_bytes_per_launch"] / out["fwd_algorithmic_bytes"]
        out["bwd_fetch_bytes"] = 2048.0 * b["FETCH_SIZE_KiB_per_launch"]
        out["bwd_write_bytes"] = 1024.0 * b["WRITE_SIZE_KiB_per_launch"]
        out["bwd_fetch_over_model_upper"] = out["bwd_fetch_bytes"] / out["bwd_model_bytes_upper"]
        out["traffic_source"] = "profiles/traffic.json (separate rocprofv3 --pmc passes, not measured in this run)"
    except Exception:
        pass
    return out


def extra_rank_step(args, torch, dev, sync):
    """What ONE rank of the 8-GPU run (BASELINE configs[3]) does per step, measured on this one GPU: forward + backward of its 4 of
    the 31 Panoptic views into its all_gather shard, the step's collective on a communicator of ONE rank (its launch and local copy
    are in; the xGMI hop is not -- a one-wavefront idle kernel of 10 / 20 / 30 us in front of it stands in for the wire), the mean
    over the gathered rank-major rows.  Two forms: "two_calls" (sks_forward, sks_backward, collective, all on one stream) and
    "one_call" (what `bench.py --gpus N` runs: sks_forward_backward with the backward AND the collective on a second stream, beside
    the dense forward).  Every figure is the median of >= 50 repetitions of a short loop, the variants interleaved round-robin on
    this box; p10 / p90 beside it.  predicted speed-up = this GPU's 31-view step (same form) / the rank step."""
    import torch.distributed as dist
    from skelsplat_amd import rasterizer as R
    wl = WORKLOADS["panoptic"]
    world, V = 8, wl["V"]
    scene, gm, params = make_scene(torch, wl, dev)
    W, H, P, C = scene.W, scene.H, scene.n_points, scene.n_joints
    vmax = (V + world - 1) // world
    local = [v for v in range(V) if v % world == 0]                      # rank 0 of 8: views 0, 8, 16, 24
    own = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        own = True
    try:
        views = R.ViewBatch.from_cameras([scene.cameras[v] for v in local])
        dL = torch.randn((len(local), C, H, W), device=dev)
        dL_all = torch.randn((V, C, H, W), device=dev)
        views_all = R.ViewBatch.from_cameras(scene.cameras)
        variants, tuned = {}, {}
        # the 3-view rank of the partition (4,4,4,4,4,4,4,3): rank 7 holds views 7, 15, 23 and a zero pad row
        local3 = [v for v in range(V) if v % world == world - 1]
        views3 = R.ViewBatch.from_cameras([scene.cameras[v] for v in local3])
        for form, oc in (("two_calls", False), ("one_call", True)):
            st = ApiStep(views, params, dL, V_total=V, exchange=(world, 0, None), one_call=oc)
            st.allg_dst = st.allg[:vmax]
            st.autotune()
            tuned[form] = {"rank_step": st.tuned}
            for us in (None, 0, 10, 20, 30):
                def fn(st=st, us=us):
                    st.no_collective, st.wire_us = us is None, float(us or 0)
                    return st()
                variants[(form, "no_exchange" if us is None else f"exchange_{us}us")] = fn
            st3 = ApiStep(views3, params, dL[:len(local3)], V_total=V, exchange=(world, world - 1, None), one_call=oc)
            st3.allg_dst = st3.allg[:vmax]
            st3.autotune()
            variants[(form, "three_view_rank_exchange_0us")] = st3
            full = ApiStep(views_all, params, dL_all, one_call=oc)
            full.autotune()
            tuned[form]["one_gpu_31views"] = full.tuned
            variants[(form, "one_gpu_31views")] = full
        inner = {k: (4 if k[1] == "one_gpu_31views" else 16) for k in variants}
        for k, fn in variants.items():      # allocations, recorded calls, the communicator
            for _ in range(3):
                fn()
        sync()
        reps = max(50, min(200, args.steps // 2))
        samples = {k: [] for k in variants}

        def sample(keys, n):
            for _ in range(n):
                for k in keys:
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(inner[k]):
                        variants[k]()
                    sync()
                    samples[k].append(1e3 * (time.perf_counter() - t0) / inner[k])
        sample(list(variants), reps)
        q = lambda xs, f: sorted(xs)[min(len(xs) - 1, int(f * (len(xs) - 1) + 0.5))]
        med = lambda k: q(samples[k], 0.5)
        # Self-consistency: the step without its exchange is not slower than with it, a longer wire not faster than a shorter one.
        # Where the medians say otherwise the pair in question is measured again (interleaved, up to 4 x the repetitions) instead
        # of allowing for it with a tolerance; what is left after that is reported as it is (`consistent`, `inversions`).
        extra_rounds = {}
        for form in ("two_calls", "one_call"):
            chain = [(form, n) for n in ("no_exchange", "exchange_0us", "exchange_10us", "exchange_20us", "exchange_30us")]
            for _ in range(3):
                bad = [(a, b) for a, b in zip(chain, chain[1:]) if med(a) > med(b)]
                if not bad:
                    break
                keys = sorted({k for pair in bad for k in pair})
                sample(keys, reps)
                extra_rounds[form] = extra_rounds.get(form, 0) + 1
        out = {"status": "PREDICTION from one GPU, not a measurement: no multi-GPU run was available to this build",
               "repetitions": reps, "ideal_speedup": V / vmax, "target": 6.0,
               "gather": "torch.distributed all_gather_into_tensor on a 1-rank RCCL communicator + sks_mean_views",
               "wire": "exchange_Nus = an idle one-wavefront kernel of N us in front of the collective (sks_prof_spin): the xGMI hop "
                       "a one-rank communicator does not make",
               "not_modelled": "RCCL's multi-rank kernel and proxy cost, rank skew (the step ends when the slowest rank's shard has "
                               "arrived everywhere): the emulation bounds the prediction from above",
               "rank_step": "the slower of the partition's two kinds of rank: 4 views (ranks 0-6) and 3 views + a pad row (rank 7)"}
        # the speed-ups are against the FASTER of the two forms of the one-GPU step, whichever form the rank step takes
        base = min(med((form, "one_gpu_31views")) for form in ("two_calls", "one_call"))
        out["one_gpu_31views_ms"] = round(base, 5)
        for form in ("two_calls", "one_call"):
            blk = {"one_gpu_31views_ms_this_form": round(med((form, "one_gpu_31views")), 5)}
            t3 = med((form, "three_view_rank_exchange_0us"))
            blk["three_view_rank_exchange_0us_ms"] = round(t3, 5)
            for (f2, name), xs in samples.items():
                if f2 != form or name in ("one_gpu_31views", "three_view_rank_exchange_0us"):
                    continue
                m = q(xs, 0.5)
                slow = max(m, t3) if name == "exchange_0us" else m      # (the step is the slowest rank's)
                blk[name] = {"rank_step_ms": round(slow, 5), "p10_p90_ms": [round(q(xs, 0.1), 5), round(q(xs, 0.9), 5)],
                             "samples": len(xs), "predicted_8gpu_speedup": round(base / slow, 3)}
            blk["autotuned"] = tuned[form]
            chain = ["no_exchange", "exchange_0us", "exchange_10us", "exchange_20us", "exchange_30us"]
            inv = [f"{a} {blk[a]['rank_step_ms']} > {b} {blk[b]['rank_step_ms']}" for a, b in zip(chain, chain[1:])
                   if blk[a]["rank_step_ms"] > blk[b]["rank_step_ms"]]
            blk["consistent"] = not inv
            blk["inversions"] = inv
            blk["extra_sampling_rounds"] = extra_rounds.get(form, 0)
            out[form] = blk
        # the headline form of this benchmark is the two-call step (its upstream gradient may depend on the image); the one-call form
        # hides the exchange under the forward and needs a gradient that does not
        out["rank_step_4views_panoptic_ms"] = out["two_calls"]["exchange_0us"]["rank_step_ms"]
        out["predicted_8gpu_speedup"] = out["two_calls"]["exchange_0us"]["predicted_8gpu_speedup"]
        out["predicted_8gpu_speedup_at_30us_wire"] = out["two_calls"]["exchange_30us"]["predicted_8gpu_speedup"]
        out["predicted_8gpu_speedup_one_call"] = out["one_call"]["exchange_0us"]["predicted_8gpu_speedup"]
        out["predicted_8gpu_speedup_one_call_at_30us_wire"] = out["one_call"]["exchange_30us"]["predicted_8gpu_speedup"]
        return out
    finally:
        if own:
            from skelsplat_amd import rccl_direct
            rccl_direct.destroy_all()
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
# N > 1: BASELINE configs[3], the same views split over the ranks (strong scaling)
# ------------------------------------------------------------------------------------------------------------
def run_sharded(args, torch, dist, dev, wl, world, rank):
    from skelsplat_amd import _lib
    from skelsplat_amd import rasterizer as R
    from skelsplat_amd.loop import MultiViewLoop
    from skelsplat_amd.heatmaps import generate_heatmaps

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    V = wl["V"]
    scene, gm, params = make_scene(torch, wl, dev)          # the same frame and cameras on every rank (seed 0)
    W, H, P, C = scene.W, scene.H, scene.n_points, scene.n_joints
    local = [v for v in range(V) if v % world == rank]
    gen = torch.Generator(device=dev).manual_seed(0)
    dL_all = torch.randn((V, C, H, W), device=dev, generator=gen)     # identical on every rank
    n_ref = max(5, args.steps // 10)
    # (A) one GPU alone: every rank runs the whole 31-view step side by side, no communication.  Its time is the N = 1
    #     reference of this run; N x 31 views / that time is the weak-scaling aggregate.
    full = ApiStep(R.ViewBatch.from_cameras(scene.cameras), params, dL_all, one_call=os.environ.get("SKS_BENCH_ONE_CALL") == "1")
    full.autotune()
    dt_full, _ = timed(full, n_ref, 3, sync)
    dt_full = max_over_ranks(dt_full)
    one_gpu_ms = 1e3 * dt_full / n_ref
    ref_mean = full().clone()     # what the sharded step must reproduce: the mean over all V views' joint gradients
    del full
    # (B) the timed region: the same V views, view v on rank v % world, one all_gather of the joint gradients per step
    lviews = R.ViewBatch.from_cameras([scene.cameras[v] for v in local]) if local else None
    dL = dL_all[local].contiguous() if local else None
    del dL_all
    torch.cuda.empty_cache()
    # the headline form, as at N = 1: sks_forward, sks_backward, the step's collective, one stream (SKS_BENCH_ONE_CALL=1: the
    # one-call form as the timed region instead -- its backward AND collective on a second stream, hidden under the forward)
    step = ApiStep(lviews, params, dL, V_total=V, exchange=(world, rank, None), one_call=os.environ.get("SKS_BENCH_ONE_CALL") == "1")
    step()                      # (allocations, the first launches)
    exchange_mode = step.choose_mode()
    step.autotune()             # (every rank the same number of steps: the collectives stay matched)
    for _ in range(args.warmup):
        step()
    sync()
    prof = not args.no_prof
    stride = prof_stride(args.steps)
    if prof:    # the forward launches of every n-th timed step carry a hipEvent pair (run_single explains)
        _lib.prof_enable(True, every=stride, kinds=(0,))
        _lib.prof_read(0), _lib.prof_read(1)
    dt, out = timed(step, args.steps, 0, sync)      # THE timed region: exactly K steps
    dt = max_over_ranks(dt)
    pf = None
    if prof:
        # topped up to ROOF_MIN_LAUNCHES by untimed steps of the same form behind the timed region -- the same number on every
        # rank (the steps hold the collective), whatever each rank has sampled
        extra = max(0, ROOF_MIN_LAUNCHES - (args.steps + stride - 1) // stride)
        _lib.prof_enable(True, every=1, kinds=(0,))
        for _ in range(extra):
            step()
        torch.cuda.synchronize()
        pf = _lib.prof_read_quantiles(0) if local else None
        _lib.prof_enable(False)
    assert torch.isfinite(out).all()
    # all_gather: the V rows are summed in view order on every rank -- bit for bit the one-GPU mean; all_reduce re-associates
    same = bool(torch.equal(out, ref_mean)) if exchange_mode == "all_gather" else \
        bool(torch.allclose(out, ref_mean, rtol=1e-5, atol=1e-6 * float(ref_mean.abs().max())))
    assert same, "the sharded step's mean differs from the one-GPU mean"
    used_direct = step.direct is not None
    step_one_call = step.one_call
    one_call_extra = None
    if not step_one_call and not args.no_extras:
        # beside it: the one-call form (needs an upstream gradient that does not depend on the image): the rank's backward and the
        # collective run on a second stream under its dense forward.  Every rank issues the same calls: the collectives stay matched
        s1 = ApiStep(lviews, params, dL, V_total=V, exchange=(world, rank, None), one_call=True)
        s1.mode = exchange_mode if s1.direct is not None or exchange_mode == "all_gather" else "all_gather"
        s1()
        s1.autotune()
        dt1, out1 = timed(s1, args.steps, args.warmup, sync)
        dt1 = max_over_ranks(dt1)
        one_call_extra = {"ms_per_step": 1e3 * dt1 / args.steps, "views_per_s": V * args.steps / dt1,
                          "speedup_vs_one_gpu_same_workload": one_gpu_ms / (1e3 * dt1 / args.steps),
                          "upstream_gradient": "independent of the image (resident dL)", "autotuned": s1.tuned,
                          "mean_equals_one_gpu": bool(torch.equal(out1, ref_mean)) if s1.mode == "all_gather" else
                          bool(torch.allclose(out1, ref_mean, rtol=1e-5, atol=1e-6 * float(ref_mean.abs().max())))}
        del s1
    del step, dL
    ms = 1e3 * dt / args.steps
    vmax = (V + world - 1) // world
    backend = dist.get_backend()
    coll = "RCCL over xGMI" if backend == "nccl" else f"{backend}: single-device test mode, timings mean nothing"
    if backend == "nccl":
        coll += ", RCCL's C API on the launch stream" if used_direct else ", through torch.distributed"
    what = (f"one all-reduce (pre-multiplied sum, weight V_local / V) of the (P,3) means of the ranks' local joint gradients"
            if exchange_mode == "all_reduce" else f"one all_gather_into_tensor of the ({vmax},P,3) joint gradients")
    strong = {"ideal_speedup": V / vmax,
              "api_step": {"one_gpu_ms_per_step": one_gpu_ms, "ms_per_step": ms, "speedup": one_gpu_ms / ms}}
    res = {
        "metric": METRIC, "value": V * args.steps / dt, "unit": "views/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "views_total": V, "views_on_rank0": len(local), "P": P, "C": C, "W": W, "H": H,
                   "parallelism": f"views sharded v % {world} over {world} ranks; {what} per step ({coll})",
                   "exchange": exchange_mode,
                   "path": "C ABI sks_forward_backward (the backward beside the dense forward on a second stream) with the step's "
                           "collective and the mean behind the backward on that second stream: hidden under the forward; eager "
                           "launches" if step_one_call else "C ABI sks_forward + sks_backward + collective on one stream, eager launches"},
        # the SAME 31-view step on one GPU alone, measured in this run (all ranks side by side, no communication): the
        # reference point for this line's `value` (the N = 1 line of bench.py times a different workload, BASELINE configs[1])
        "one_gpu_same_workload_views_per_s": V * n_ref / dt_full,
        "one_gpu_same_workload_ms_per_step": one_gpu_ms,
        "speedup_vs_one_gpu_same_workload": one_gpu_ms / ms,
        "mean_equals_one_gpu": "bit for bit" if exchange_mode == "all_gather" else "rtol 1e-5",
        "weak_scaling_views_per_s": world * V * n_ref / dt_full,
    }
    if one_call_extra:
        res["one_call_step"] = one_call_extra
    if pf and pf[1]:
        res["roofline"] = roofline_entry(4.0 * H * W * (C + 1) * len(local), pf, wl["name"], args.steps,
                                         traffic_scale=len(local) / V)
        res["roofline"]["note"] = f"rank 0's launch: its {len(local)} local views"
        res["roofline"]["timed_in"] = ("the timed region of this line" + (": the backward and the step's collective run BESIDE the forward "
                                       "on a second queue there" if step_one_call else ""))
    if not args.no_extras:
        # (C, D) the loop's own step -- render + masked-L2 + backward + [all_gather] + Adam (train.py:130-222): dense and
        # sparse fused, each alone on one GPU (shard_views=False, all ranks side by side) and sharded over the ranks
        try:
            p2d = torch.tensor(scene.poses_2d, device=dev)
            gm.training_setup()
            hm = generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d, scene.cameras)
            nl = max(5, args.steps // 10)
            graph_dist = os.environ.get("SKS_GRAPH_COLLECTIVES") == "1"
            for tag, kw in (("loop_dense", dict(sparse=False)), ("loop_sparse", dict(sparse=True))):
                one = MultiViewLoop(fresh_model(scene, wl["dataset"], dev), scene.cameras, hm, dataset=wl["dataset"],
                                    accumulation_steps=V, shard_views=False, **kw)
                t1 = max_over_ranks(loop_ms(torch, one, nl, sync))
                del one
                shl = MultiViewLoop(fresh_model(scene, wl["dataset"], dev), scene.cameras, hm, dataset=wl["dataset"],
                                    accumulation_steps=V, **kw)
                tn = max_over_ranks(loop_ms(torch, shl, nl, sync))
                strong[tag] = {"one_gpu_ms_per_step": t1, "ms_per_step": tn, "speedup": t1 / tn}
                del shl
                if graph_dist:   # the sharded group, RCCL all_gather included, replayed as a hipGraph (opt-in)
                    shg = MultiViewLoop(fresh_model(scene, wl["dataset"], dev), scene.cameras, hm, dataset=wl["dataset"],
                                        accumulation_steps=V, use_graph=True, graph_collectives=True, **kw)
                    strong[tag]["ms_per_step_hipgraph"] = max_over_ranks(loop_ms(torch, shg, nl, sync))
                    del shg
            # (E) frame sharding: every rank optimises its own frames (500 iterations each, hipGraphs), no communication
            fl = MultiViewLoop(fresh_model(scene, wl["dataset"], dev), scene.cameras, hm, dataset=wl["dataset"],
                               accumulation_steps=V, shard_views=False, use_graph=True)
            pts = torch.tensor(scene.pose_3d_init, device=dev, dtype=torch.float32)
            fl.new_scene(pts, poses_2d=p2d)
            fl.run(500)
            sync()
            t0 = time.perf_counter()
            for _ in range(3):
                fl.new_scene(pts, poses_2d=p2d)
                fl.run(500)
            sync()
            tf = max_over_ranks((time.perf_counter() - t0) / 3)
            res["frame_sharded"] = {"frame_ms": 1e3 * tf, "frames_per_s_all_ranks": world / tf,
                                    "note": "500 iterations per frame incl. heat-map generation; zero communication"}
            del fl
            if V <= 8:   # few-view rigs: every rank also batches its frames (loop.FramePipeline)
                from skelsplat_amd.loop import FramePipeline
                F, S = min(16, 64 // V), 2
                pipe = FramePipeline(fresh_model(scene, wl["dataset"], dev), scene.cameras, frames=F, streams=S,
                                     dataset=wl["dataset"], accumulation_steps=V)
                ptsN, p2dN = pts[None].repeat(S * F, 1, 1), p2d[None].repeat(S * F, 1, 1, 1)
                pipe.optimize_sequence(ptsN, p2dN, iterations=500)
                sync()
                t0 = time.perf_counter()
                pipe.optimize_sequence(ptsN, p2dN, iterations=500)
                sync()
                tp = max_over_ranks(time.perf_counter() - t0)
                res["frame_sharded"]["pipeline_frames_per_s_all_ranks"] = world * S * F / tp
                del pipe
        except Exception as e:
            res["loop_error"] = repr(e)[:300]
    res["strong_scaling"] = strong
    # the three steps' speed-ups at the top level: north_star's >= 6 x is expected of the DENSE loop step (it is made of bytes,
    # DESIGN.md section 6); the API step's ~35 us of fixed cost and one exchange leave it around 6 x, the sparse step is a
    # 90 us chain of latencies that does not shard
    res["api_step_speedup_vs_one_gpu"] = strong["api_step"]["speedup"]
    for tag in ("loop_dense", "loop_sparse"):
        if tag in strong:
            res[tag + "_speedup_vs_one_gpu"] = strong[tag]["speedup"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: h36m on one GPU (BASELINE configs[1]), panoptic when sharded (configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--no-dropin", action="store_true", help="skip the literal drop-in iteration extra (its single-view "
                    "launches would mix into per-kernel averages of a rocprofv3 run)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement")
    ap.add_argument("--form", default="both", choices=["both", "one", "two"],
                    help="N = 1: the step as two C-ABI calls (sks_forward + sks_backward: the headline), as one (sks_forward_backward, "
                         "which needs an upstream gradient independent of the image), or both (default: the headline is the two "
                         "calls, the one call is reported beside it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs without a launcher around us: start one rank per GPU as CHILD processes (nothing here has
        # touched the GPU yet, and nothing is exec'ed) and hand their result line and exit code through
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # Native libraries write to the process's stdout too (RCCL prints a version banner from C stdio, flushed at exit, i.e.
    # AFTER the result line): fd 1 is pointed at stderr for the whole run and the one JSON line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback)")
    # SKS_BENCH_ONE_DEVICE=1 (tests): every rank on GPU 0 over gloo -- RCCL refuses two ranks on one device; this checks
    # the sharded logic at world > 1 on a single-GPU box, its timings mean nothing
    one_device = os.environ.get("SKS_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SKS_BENCH_FORCE_DIST") == "1"   # the env switch runs the sharded path at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = WORKLOADS[args.workload or ("panoptic" if use_dist else "h36m")]
    if use_dist:
        res = run_sharded(args, torch, dist, dev, wl, world, rank)
        if world == 1 and not args.no_cpu_baseline:    # (the CPU sample is the N = 1 line's: no rank waits in a collective for it)
            from skelsplat_amd.scene import SyntheticScene
            sc = SyntheticScene(wl["dataset"], n_views=wl["V"], seed=0)
            _, _, prm = make_scene(torch, wl, dev)
            res["cpu_baseline"] = cpu_baseline(sc, prm, n_views=2)
        dist.barrier()     # all ranks leave the process group together
    else:
        res, scene, params = run_single(args, torch, dev, wl)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(scene, params, n_views=2)
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    # nothing may follow the result line on either stream: what native libraries still hold in their stdio buffers (RCCL's
    # banner) or print while tearing down goes to /dev/null from here on
    sys.stdout.flush()
    sys.stderr.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    os.dup2(devnull, 2)
    if use_dist:
        from skelsplat_amd import rccl_direct
        rccl_direct.destroy_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
