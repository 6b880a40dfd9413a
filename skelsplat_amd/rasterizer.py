"""Python surface of the rasterizer: the reference's GaussianRasterizationSettings / GaussianRasterizer API
(DGR/diff_gaussian_rasterization_h36m/__init__.py:21-207) on top of the C ABI in include/skelsplat_hip.h,
plus the batched multi-view entry points the MI355X loop uses.

"DGR/" = submodules/diff-gaussian-rasterization-h36m/ of the reference.
"""
from typing import NamedTuple, Optional

import os
import weakref

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    """Same 13 fields, same order as DGR/diff_gaussian_rasterization_h36m/__init__.py:143-156."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


# ------------------------------------------------------------------------------------------------------------
# scratch management (replaces resizeFunctional, DGR/rasterize_points.cu:27-33): torch owns every byte
# ------------------------------------------------------------------------------------------------------------
_accum_cache = {}


def _accum(device, stream, V, P, C):
    """Backward partial-sum scratch (uninitialised is fine): one buffer per (device, stream, shape)."""
    key = (device.index, stream, V, P, C)
    buf = _accum_cache.get(key)
    if buf is None:
        _, _, nbytes = _lib.scratch_bytes(V, max(P, 1), C, 16, 16)
        buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        # a buffer allocated while a hipGraph is being captured belongs to that graph's private pool: the graph keeps it
        # alive for its own replays, but it must not be handed to other graphs or to eager code on a recycled stream handle
        if not torch.cuda.is_current_stream_capturing():
            _accum_cache[key] = buf
    return buf


def reset_scratch():
    """Drop cached accumulators (call after an aborted backward, e.g. an exception between kernels)."""
    _accum_cache.clear()


class Workspace:
    """Reusable outputs and scratch of forward_views / backward_views.  A caller that runs the same shapes every step
    (a training loop: train.py:130-222 renders the same cameras 500 times per frame) passes the same Workspace to every
    call and the calls allocate nothing: a dozen caching-allocator round trips per step are ~20 us of host time next
    to ~78 us of kernels on the H36M step.  What the calls return are the workspace's tensors: the NEXT call with the same
    shapes overwrites them (the reference's own outputs are fresh tensors per call; callers that keep results across
    calls simply do not pass a workspace).  Every kernel writes every element, so nothing is ever cleared."""

    def __init__(self):
        self._t = {}
        self._plans = {}     # "fwd" / "bwd" -> the last call's recorded C-ABI argument list (see _plan_key)
        self._aux = {}
        self._pending_join = None     # device whose second stream still holds a forward_backward_views(join=False) backward
        self._tuned = {}     # recorded forward's key -> the tuner's pick for it (tune())

    def aux_stream(self, dev_index):
        """The second stream forward_backward_views runs the backward on (created on first use, one per device)."""
        st = self._aux.get(dev_index)
        if st is None:
            st = self._aux[dev_index] = torch.cuda.Stream(device=dev_index)
        return st

    def join(self, dev_index):
        """The current stream waits for everything enqueued on the second stream (forward_backward_views(join=False))."""
        torch.cuda.current_stream(dev_index).wait_stream(self.aux_stream(dev_index))
        self._pending_join = None

    def tune(self, step_fn, candidates=None, reps=8, rounds=3, confirm=True):
        """Picks how the forward's fill role writes for the step `step_fn` issues through THIS workspace's recorded forward, by
        timing it: 4 KB passes (or rows, on row-aligned widths) per fill block (bits 0..7 of a candidate -> bits 8..15 of the flags,
        0 = the library's default of two) and the KIND of store (candidate bit 8, PLAIN_STORES -> SKS_NO_NT_STORES).  Why per step
        and at run time: the fill role is bound by the life time of its ~36 000 blocks, and what shares the chip with them decides
        the best size -- two passes when the forward runs alone (sks_forward, then sks_backward), three for the H36M step through
        sks_forward_backward (the backward's wavefronts hold slots beside it), four for all 31 Panoptic views, five for four of
        them.  The store kind decides where the zeros go first: non-temporal stores stream past the 256 MB Infinity Cache to HBM;
        plain stores may stay in it, so a call that rewrites the SAME ~cache-sized output buffers step after step (a Workspace)
        hands them over at the cache's rate while the previous step's lines drain behind it (H36M, 288 MB per call) -- a gain of
        the STEP only where nothing else wants that bandwidth, and 40 % slower on Panoptic and the stress scene
        (NOTES_experiments.md).  The pick is the candidate with the lowest step time in its least disturbed round; the default
        stays unless another beats it by more than 2 % twice (_pick).  None of it moves a result bit (tests/test_raster_gpu.py).  `step_fn()` is called len(candidates) x rounds x (2 + reps)
        times (+ TUNE_WARM untimed calls in front, + up to 3 x 2 x (2 + reps) for the confirmation unless confirm=False: a caller
        whose step holds a collective wants every rank to issue the same number of steps) with device synchronisations in between
        (once, before a long loop); the candidates are interleaved round-robin.
        The pick also becomes the default of later Workspace recordings of the same shape.  Returns (best, {candidate: us})."""
        return _workspace_tune(self, step_fn, TUNE_CANDIDATES if candidates is None else candidates, reps, rounds, confirm)

    def settle(self):
        """A forward_backward_views(join=False) whose caller never joined: the next call through this workspace makes the join
        itself, before its geometry kernel overwrites what the backward on the second stream may still be reading."""
        if self._pending_join is not None:
            self.join(self._pending_join)

    def get(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype, device)
        t = self._t.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._t[key] = t
        return t


def _sig(t):
    """What identifies a tensor argument of a recorded call: None (not provided), (pointer, shape) of a contiguous fp32
    ROCm tensor, or False = "take the validating path" (anything else)."""
    if t is None or t.numel() == 0:     # (an empty tensor is the reference's "not provided": GaussianRasterizer hands torch.Tensor([]))
        return None
    if t.dtype is not torch.float32 or not t.is_cuda or not t.is_contiguous():
        return False
    return (t.data_ptr(), t.shape)


def _replay(fn, args, dev_index):
    """Issue a recorded call on the CURRENT stream of the tensors' device.  The host side of a step matters here: the
    H36M step is ~78 us of kernels, and building two ~30-argument ctypes calls from tensors (validation, data_ptr,
    current_stream, device guard) was ~60 us of Python per step -- host-bound on a slow box.  A recorded call is one
    tuple comparison and one ctypes call."""
    args[-1] = torch._C._cuda_getCurrentRawStream(dev_index)
    if torch._C._cuda_getDevice() == dev_index:
        return fn(*args)
    with torch.cuda.device(dev_index):
        return fn(*args)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"skelsplat_amd: `{name}` must live on a ROCm device (got {t.device}); "
                           "there is no CPU fallback")


def _f32c(t, name):
    if t is None or t.numel() == 0:
        return None
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"skelsplat_amd: `{name}` must be float32 (got {t.dtype})")
    return t.contiguous()


_VIEW_CACHE = {}     # ViewBatch.from_settings


class ViewBatch:
    """V cameras packed for one launch sequence.  The dense entry points (sks_forward / sks_backward write and read
    (V,C,H,W) tensors) need one image size; the sparse fused-loss path writes nothing dense and takes a batch whose
    views differ in size (`sizes`: per-view (W, H); H36M mixes 1000- and 1002-wide sensors, dataset_readers.py:68-80):
    then W, H are the largest and `wh` is the HOST array the C ABI's `view_wh` argument wants."""

    def __init__(self, viewmatrices, projmatrices, tanfovx, tanfovy, W, H, sizes=None):
        import ctypes
        self.viewmatrix = _f32c(viewmatrices, "viewmatrix").reshape(-1, 16)
        self.projmatrix = _f32c(projmatrices, "projmatrix").reshape(-1, 16)
        self.V = self.viewmatrix.shape[0]
        if self.V > _lib.SKS_MAX_VIEWS:
            raise RuntimeError(f"at most {_lib.SKS_MAX_VIEWS} views per call")
        self.tanfovx = _lib.farray(tanfovx)
        self.tanfovy = _lib.farray(tanfovy)
        assert len(tanfovx) == self.V and len(tanfovy) == self.V
        self.W, self.H = int(W), int(H)
        self.sizes = [(self.W, self.H)] * self.V if sizes is None else [(int(w), int(h)) for w, h in sizes]
        assert len(self.sizes) == self.V
        self.mixed = any(sz != (self.W, self.H) for sz in self.sizes)
        self.wh = (ctypes.c_int * (2 * self.V))(*[x for sz in self.sizes for x in sz]) if self.mixed else None

    @classmethod
    def from_cameras(cls, cams, allow_mixed=False):
        import math
        sizes = [(int(c.image_width), int(c.image_height)) for c in cams]
        W, H = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if not allow_mixed and any(sz != (W, H) for sz in sizes):
            raise RuntimeError("all views of a batch must share the image size")
        vm = torch.stack([c.world_view_transform.reshape(16) for c in cams])
        pm = torch.stack([c.full_proj_transform.reshape(16) for c in cams])
        return cls(vm, pm, [math.tan(c.FoVx * 0.5) for c in cams], [math.tan(c.FoVy * 0.5) for c in cams], W, H, sizes)

    @classmethod
    def from_settings(cls, rs):
        """One view from a GaussianRasterizationSettings.  A training loop builds the settings of the same few cameras over
        and over (train.py:140 -> gaussian_renderer/__init__.py:46-60): the batch of a camera is kept while its two matrices
        are the same, unmodified tensors (the reference's are transposed views, scene/cameras.py:94-97: each rebuild would
        cost two small copy kernels and two ctypes arrays)."""
        vm, pm = rs.viewmatrix, rs.projmatrix
        key = (vm.data_ptr(), vm._version, pm.data_ptr(), pm._version, rs.tanfovx, rs.tanfovy, rs.image_width, rs.image_height)
        hit = _VIEW_CACHE.get(key)
        if hit is not None and hit[1]() is vm and hit[2]() is pm:
            return hit[0]
        if len(_VIEW_CACHE) > 256:
            _VIEW_CACHE.clear()
        vb = cls(vm, pm, [rs.tanfovx], [rs.tanfovy], rs.image_width, rs.image_height)
        _VIEW_CACHE[key] = (vb, weakref.ref(vm), weakref.ref(pm))
        return vb


class ForwardState:
    """What backward needs (the reference keeps geomBuffer / binningBuffer / imgBuffer + num_rendered,
    DGR/diff_gaussian_rasterization_h36m/__init__.py:87-89)."""
    __slots__ = ("views", "P", "C", "flags", "scale_modifier", "geom", "binning", "bin_capacity", "radii",
                 "num_rendered_dev", "frames", "plan_key", "chunks")

    def __init__(self):
        self.frames = 1
        self.plan_key = None
        self.chunks = None      # more than SKS_MAX_CHANNELS channels: [(view, c0, c1, feature chunk, ForwardState of that call)]


def forward_views(views: ViewBatch, means3D, features, opacities, scales, rotations, cov3D_precomp,
                  scale_modifier=1.0, antialiasing=False, clamp01=False, debug=False, force_binned=False,
                  bin_capacity=None, want_aux=False, tune_flags=0, check_capacity=True, workspace=None, plans=None):
    """Raw batched forward.  Returns (color (V,C,H,W), invdepth (V,1,H,W), radii (V,P) int32, state[, final_T, n_contrib]).
    `workspace`: a Workspace whose tensors receive the outputs (see there).  `check_capacity` (binned path, P > 256):
    True (the default) = read the pair count back every call (one host sync, like the reference: rasterizer_impl.cu:283-288) and
    redo the forward with a larger arena when it was too small -- never a wrong image; "lazy" = never synchronise, detect an
    overflowed arena at a LATER call of the shape, as soon as the GPU has been through the overflowed one (that call raises: the
    image before it missed entries; the arena has been grown for the calls after it); "auto" (what loops that own their error
    handling ask for: bench.py's stress step) = True for the first call of a shape, which also sizes the arena with 50 % headroom
    over that call's count, lazy afterwards; False = no check.
    `plans` (the autograd path, small path only): a dict that keeps the validated C-ABI argument block of a call; while the same
    parameter tensors and switches come back, a call allocates its four FRESH tensors (outputs + geometry scratch: they belong to
    the caller's autograd graph), patches their pointers into a copy of the block and launches -- no re-validation."""
    lib = _lib.load()
    key = None
    if features is not None and features.numel() and means3D is not None and means3D.dim() == 2 and means3D.shape[0] \
            and features.numel() // means3D.shape[0] > _lib.SKS_MAX_CHANNELS:
        return _forward_views_wide(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing,
                                   clamp01, debug, force_binned, bin_capacity, want_aux, tune_flags, check_capacity)
    if plans is not None and workspace is None and not want_aux:
        pkey = _fwd_key(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing,
                        clamp01, debug, force_binned, bin_capacity, tune_flags, check_capacity)
        hit = plans.get(pkey)
        if hit is not None:
            args_t, dev_index, dev, (V, P, C, H, W, gbytes, flags) = hit
            color = torch.empty((V, C, H, W), dtype=torch.float32, device=dev)
            invdepth = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((V, P), dtype=torch.int32, device=dev)
            geom = torch.empty((gbytes,), dtype=torch.uint8, device=dev)
            args = list(args_t)
            args[17], args[18], args[19], args[20] = color.data_ptr(), invdepth.data_ptr(), radii.data_ptr(), geom.data_ptr()
            _lib.check(_replay(lib.sks_forward, args, dev_index), "sks_forward")
            st = ForwardState()
            st.views, st.P, st.C, st.flags, st.scale_modifier = views, P, C, flags, float(scale_modifier)
            st.geom, st.binning, st.bin_capacity, st.radii, st.num_rendered_dev = geom, None, 0, radii, None
            st.plan_key = pkey
            return color, invdepth, radii, st
    if workspace is not None:
        workspace.settle()
    if workspace is not None and not want_aux:
        # the same call as last time (same tensors, same switches)?  Then the validated argument list is replayed as is.
        key = _fwd_key(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing,
                       clamp01, debug, force_binned, bin_capacity, tune_flags, check_capacity)
        plan = workspace._plans.get("fwd")
        if plan is not None and plan[0] == key:
            _, _views, args, dev_index, result, cap_check = plan
            _replay_cap_prepare(workspace, plan, check_capacity)
            rc = _replay(lib.sks_forward, args, dev_index)
            if rc != 0:
                del workspace._plans["fwd"]
            _lib.check(rc, "sks_forward")
            if _replay_cap_finish(workspace, plan, check_capacity):
                return result
            # (the binning arena overflowed: the validating path below grows it and redoes)
    if views.mixed:
        raise RuntimeError("the dense forward writes one (V,C,H,W) tensor: all views of the batch must share the image size")
    if means3D is None or means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # DGR/rasterize_points.cu:58-60
    _need_gpu(means3D, "means3D")
    if means3D.shape[0] == 0:   # DGR/rasterize_points.cu:88: nothing is rasterised, the outputs stay zero
        dev, V, W, H = means3D.device, views.V, views.W, views.H
        C = int(features.shape[-1])
        st = ForwardState()
        st.views, st.P, st.C, st.flags, st.scale_modifier = views, 0, C, 0, float(scale_modifier)
        st.geom = st.binning = st.num_rendered_dev = None
        st.bin_capacity, st.radii = 0, torch.zeros((V, 0), dtype=torch.int32, device=dev)
        out = (torch.zeros((V, C, H, W), device=dev), torch.zeros((V, 1, H, W), device=dev), st.radii, st)
        if want_aux:
            out += (torch.ones((V, H, W), device=dev), torch.zeros((V, H, W), dtype=torch.int32, device=dev))
        return out
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    features = _f32c(features, "features")
    feat2 = features.reshape(P, -1)
    C = feat2.shape[1]
    opacities = _f32c(opacities, "opacities")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
    V, W, H = views.V, views.W, views.H
    flags = (_lib.SKS_ANTIALIASING if antialiasing else 0) | (_lib.SKS_CLAMP01 if clamp01 else 0) | \
            (_lib.SKS_DEBUG_SYNC if debug else 0) | (_lib.SKS_FORCE_BINNED if force_binned else 0) | int(tune_flags) | _ENV_TUNE
    if plans is not None and workspace is None and AUTOTUNE and not flags & _FILL_BITS and not _MEASURING[0] and not force_binned \
            and (dev.index, V, P, C, W, H, "fresh") not in _FILL_TUNE:
        # the autograd path sees a shape for the first time: its fill configuration is measured once (tune_forward), on this call's
        # own tensors -- every later call of the shape, recorded or not, launches with the pick
        tune_forward(views, means3D, feat2, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01,
                     tune_flags=tune_flags)
    if not flags & _FILL_BITS and not _MEASURING[0]:      # no explicit fill configuration: what tune_forward / Workspace.tune measured for this shape, if anything
        flags |= _FILL_TUNE.get((dev.index, V, P, C, W, H, "workspace" if workspace is not None else "fresh"), 0)
    binned = force_binned or P > _lib.SKS_SMALL_P
    bin_capacity_given = bin_capacity
    if binned and bin_capacity is None:
        bin_capacity = _BIN_CAP_HINT.get((means3D.device.index, views.V, P, C, views.W, views.H), max(4096, 16 * P))
    cap = int(bin_capacity or 0)
    gbytes, bbytes, _ = _scratch_bytes_cached(V, max(P, 1), C, W, H, cap)
    def new(name, shape, dtype):
        if workspace is None:
            return torch.empty(shape, dtype=dtype, device=dev)
        return workspace.get(("fwd", name), shape, dtype, dev)

    color = new("color", (V, C, H, W), torch.float32)
    invdepth = new("invdepth", (V, 1, H, W), torch.float32)
    radii = new("radii", (V, P), torch.int32)
    geom = new("geom", (gbytes,), torch.uint8)
    binning = new("binning", (bbytes,), torch.uint8) if binned else None
    cap_key = (dev.index, V, P, C, W, H)
    lazy = binned and (check_capacity == "lazy" or (check_capacity == "auto" and cap_key in _BIN_CAP_SEEN))
    if lazy:
        # the pair counts go straight to pinned host memory (k_bin_scan stores them there: no copy launch, no event): they are
        # looked at when the NEXT call of the shape comes in -- first the previous call's, which may raise
        nrend = _lazy_probe(cap_key, cap)     # (None only inside a hipGraph capture)
    elif binned and check_capacity is True and not torch.cuda.is_current_stream_capturing():
        nrend = _sync_counts(cap_key)     # pinned host memory, preset to -1 (see _wait_counts)
    else:
        nrend = new("nrend", (V + 1,), torch.int32) if binned else None   # [0, V): written by k_bin_scan
    final_T = torch.empty((V, H, W), dtype=torch.float32, device=dev) if want_aux else None
    n_contrib = torch.empty((V, H, W), dtype=torch.int32, device=dev) if want_aux else None
    args = [V, P, C, W, H, views.viewmatrix.data_ptr(), views.projmatrix.data_ptr(), views.tanfovx,
            views.tanfovy, _lib.ptr(means3D), _lib.ptr(feat2), _lib.ptr(opacities), _lib.ptr(scales),
            _lib.ptr(rotations), _lib.ptr(cov3D_precomp), float(scale_modifier), flags,
            color.data_ptr(), invdepth.data_ptr(), _lib.ptr(radii), geom.data_ptr(),
            _lib.ptr(binning), cap, _lib.ptr(nrend[0] if isinstance(nrend, tuple) else nrend), _lib.ptr(final_T), _lib.ptr(n_contrib), None]
    _lib.check(_replay(lib.sks_forward, args, dev.index), "sks_forward")
    if binned and check_capacity:
        # The reference reads the pair count back on EVERY forward to size its buffers (rasterizer_impl.cu:283-288: a
        # blocking D2H copy between the scan and the duplication).  Here the arena is persistent and the count stays on
        # the device: check_capacity=True reads it back (exact: grow and redo when the arena was too small -- entries
        # beyond it were dropped); "lazy" (what "auto" does after it has sized the arena once per shape with a
        # synchronous first call) has the counts stored into pinned host memory and looks at them when the NEXT call for the
        # shape comes in (_lazy_probe) -- no host synchronisation, no copy launch on the fast path; an overflow found that way
        # grows the arena for the calls to come and raises, because the image that call produced was missing entries.
        if not lazy:
            need = _wait_counts(nrend, V, dev.index) if isinstance(nrend, tuple) else int(nrend[:V].max().item())
            _BIN_CAP_SEEN.add(cap_key)
            if check_capacity == "auto" and need <= cap and bin_capacity_given is None:
                # later calls of the shape go unchecked until the call after them: leave them room to grow
                _BIN_CAP_HINT[cap_key] = max(cap, int(need * 1.5) + 1024)
            if need > cap:
                _BIN_CAP_HINT[cap_key] = int(need * 1.25) + 1024
                return forward_views(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier,
                                     antialiasing, clamp01, debug, force_binned, int(need * 1.25) + 1024, want_aux, tune_flags,
                                     check_capacity, workspace)
    st = ForwardState()
    st.views, st.P, st.C, st.flags, st.scale_modifier = views, P, C, flags, float(scale_modifier)
    st.geom, st.binning, st.bin_capacity, st.radii = geom, binning, cap, radii
    st.num_rendered_dev = nrend[0] if isinstance(nrend, tuple) else nrend
    if want_aux:
        return color, invdepth, radii, st, final_T, n_contrib
    if plans is not None and workspace is None and not binned and P and all(sg is not False for sg in pkey[1:7]):
        if len(plans) > 64:
            plans.clear()
        # (the block keeps the parameter tensors' pointers: `keep` holds the tensors so that the pointers stay theirs)
        plans[pkey] = (list(args), dev.index, dev, (V, P, C, H, W, gbytes, flags))
        plans[("keep", pkey)] = (views, means3D, feat2, opacities, scales, rotations, cov3D_precomp)
        st.plan_key = pkey
    if key is not None and all(sg is not False for sg in key[1:7]) and not torch.cuda.is_current_stream_capturing():
        # (the tensors the pointers belong to stay alive in `keep`; the views object is held so that its id stays its own)
        keep = (means3D, feat2, opacities, scales, rotations, cov3D_precomp)
        cap_check = None
        if binned:
            # (the binning buffer needs no clearing between calls: every counter is written before it is read)
            args = list(args)
            if check_capacity is True:
                cap_check = (nrend, cap, cap_key)
            elif check_capacity:
                cap_check = (None, cap, cap_key)     # replays are lazy calls: each takes a probe buffer of the shape (_lazy_probe)
        workspace._plans["fwd"] = (key, (views, keep), args, dev.index, (color, invdepth, radii, st), cap_check)
    return color, invdepth, radii, st


def _replay_cap_prepare(workspace, plan, check_capacity):
    """In front of a REPLAYED binned forward (forward_views, forward_backward_views): where this call's pair counts go.  Lazy modes:
    a pinned probe buffer of the shape (looking at earlier calls' counts first: raises when one of them overflowed its arena);
    synchronous check: the shape's pinned counts preset to "not written yet"."""
    args, result, cap_check = plan[2], plan[4], plan[5]
    if cap_check is None:
        return
    if check_capacity is not True:
        try:
            host = _lazy_probe(cap_check[2], cap_check[1])     # earlier calls' counts; a buffer for this call's (or None)
        except RuntimeError:
            # the recorded call holds the overflowed arena: drop it, so that the next call takes the validating path
            # and allocates the grown one (`_BIN_CAP_HINT`)
            workspace._plans.pop("fwd", None)
            raise
        args[23] = None if host is None else host.data_ptr()
        result[3].num_rendered_dev = host
    else:
        cap_check[0][1][:args[0]] = -1       # (the pinned counts of the synchronous check: "not written yet")


def _replay_cap_finish(workspace, plan, check_capacity):
    """Behind it: True = the call stands.  False (synchronous check only): its arena was too small -- the recorded call is dropped,
    the caller takes the validating path, which grows the arena and redoes the forward."""
    args, dev_index, cap_check = plan[2], plan[3], plan[5]
    if cap_check is None or check_capacity is not True:
        return True       # (lazy: the counts of this call are looked at when the next one comes in)
    nr, pcap, ckey = cap_check
    if _wait_counts(nr, args[0], dev_index) <= pcap:
        return True
    workspace._plans.pop("fwd", None)
    return False


def _forward_views_wide(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01,
                        debug, force_binned, bin_capacity, want_aux, tune_flags, check_capacity):
    """Any number of channels ("any C via a generic path").  The kernels hold a pixel's channels in registers,
    SKS_MAX_CHANNELS at most; the channels of a Gaussian splat are independent given its alpha, so a wider feature row is rendered
    as ceil(C / 32) calls per view on 32-channel slices of the features, each writing its own planes: the same arithmetic per
    channel as one wide call would do (the oracle, which takes any C, is the judge: bit for bit).  The backward is linear in the
    channels (dL/dalpha is a sum over them): the slices' gradients add up.  A generic path, not a fast one: one launch sequence
    per (view, slice), outputs copied into place."""
    if want_aux:
        raise RuntimeError("final_T / n_contrib are not available beyond SKS_MAX_CHANNELS channels")
    if views.mixed:
        raise RuntimeError("the dense forward writes one (V,C,H,W) tensor: all views of the batch must share the image size")
    _need_gpu(means3D, "means3D")
    P = means3D.shape[0]
    feat2 = _f32c(features, "features").reshape(P, -1)
    C, V, W, H, dev = feat2.shape[1], views.V, views.W, views.H, means3D.device
    M = _lib.SKS_MAX_CHANNELS
    color = torch.empty((V, C, H, W), dtype=torch.float32, device=dev)
    invdepth = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((V, P), dtype=torch.int32, device=dev)
    st = ForwardState()
    st.views, st.P, st.C, st.scale_modifier, st.flags = views, P, C, float(scale_modifier), 0
    st.geom = st.binning = st.num_rendered_dev = None
    st.bin_capacity, st.radii, st.chunks = 0, radii, []
    for v in range(V):
        one = ViewBatch(views.viewmatrix[v:v + 1], views.projmatrix[v:v + 1], [views.tanfovx[v]], [views.tanfovy[v]], W, H)
        for c0 in range(0, C, M):
            c1 = min(C, c0 + M)
            fc = feat2[:, c0:c1].contiguous()
            col, inv, rad, sub = forward_views(one, means3D, fc, opacities, scales, rotations, cov3D_precomp, scale_modifier,
                                               antialiasing, clamp01, debug, force_binned, bin_capacity, False, tune_flags, check_capacity)
            color[v, c0:c1].copy_(col[0])
            if c0 == 0:
                invdepth[v].copy_(inv[0])
                radii[v].copy_(rad[0])
            st.chunks.append((v, c0, c1, fc, sub))
    return color, invdepth, radii, st


def _backward_views_wide(st, means3D, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg, want_dfeatures,
                         tune_flags):
    dev, V, P, C = means3D.device, st.views.V, st.P, st.C
    dL_dcolor = _f32c(dL_dcolor, "dL_dout_color").reshape(V, C, st.views.H, st.views.W)
    dL_dinvdepth = _f32c(dL_dinvdepth, "dL_dout_invdepth")
    z = lambda *s_: torch.zeros(s_, dtype=torch.float32, device=dev)
    out = dict(means3D=z(V, P, 3), means2D=z(V, P, 3), opacities=z(V, P, 1), cov3D=z(V, P, 6),
               scales=z(V, P, 3) if scales is not None and scales.numel() else None,
               rotations=z(V, P, 4) if rotations is not None and rotations.numel() else None,
               features=z(V, P, C) if want_dfeatures else None)
    bgC = _bg_channels(bg, C, dev)
    for v, c0, c1, fc, sub in st.chunks:
        g = backward_views(sub, means3D, fc, opacities, scales, rotations, cov3D_precomp, dL_dcolor[v:v + 1, c0:c1],
                           None if (dL_dinvdepth is None or c0) else dL_dinvdepth.reshape(V, 1, st.views.H, st.views.W)[v:v + 1],
                           None if bgC is None else bgC[c0:c1], want_dfeatures, tune_flags)
        for k in ("means3D", "means2D", "opacities", "cov3D", "scales", "rotations"):
            if out[k] is not None:
                out[k][v] += g[k][0]
        if want_dfeatures:
            out["features"][v, :, c0:c1] = g["features"][0]
    return out


def _fwd_key(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01, debug,
             force_binned, bin_capacity, tune_flags, check_capacity):
    return (id(views), _sig(means3D), _sig(features), _sig(opacities), _sig(scales), _sig(rotations), _sig(cov3D_precomp),
            scale_modifier, antialiasing, clamp01, debug, force_binned, bin_capacity, tune_flags, check_capacity)


def _bwd_key(st, means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg, want_dfeatures,
             tune_flags, want_mean, out_means3D, stream):
    return (id(st), _sig(means3D), _sig(features), _sig(opacities), _sig(scales), _sig(rotations), _sig(cov3D_precomp),
            _sig(dL_dcolor), _sig(dL_dinvdepth), None if bg is None else (id(bg), bg._version), want_dfeatures, tune_flags,
            want_mean, None if out_means3D is None else out_means3D.data_ptr(), stream)   # (the partial-sum scratch is per stream)


_SCRATCH_BYTES = {}
_BIN_CAP_HINT = {}     # (device, V, P, C, W, H) -> arena capacity learned from an overflow
_BIN_CAP_SEEN = set()  # shapes whose arena a synchronous call has sized already ("auto" goes lazy after that)
_BIN_PROBE = {}        # shape -> _Probes: the pinned host buffers lazy calls of the shape write their pair counts into


class _Probes:
    __slots__ = ("pending", "free")

    def __init__(self):
        self.pending, self.free = [], []      # pending: [(pinned int32 tensor, its numpy view, capacity of that call)], oldest first


_SYNC_PROBE = {}       # shape -> (pinned int32 tensor, its numpy view): the counts of the synchronous check


def _sync_counts(cap_key):
    hit = _SYNC_PROBE.get(cap_key)
    if hit is None:
        host = torch.empty((cap_key[1] + 1,), dtype=torch.int32).pin_memory()
        hit = _SYNC_PROBE[cap_key] = (host, host.numpy())
    hit[1][:] = -1
    return hit


def _wait_counts(probe, V, dev_index):
    """check_capacity=True: the pair counts of the call just enqueued.  The reference reads them back with a blocking copy between
    its scan and its duplication kernels (rasterizer_impl.cu:283-288).  Here k_bin_scan stores them straight into pinned host
    memory ~30 us into the launch sequence and the host spins on THAT -- not on the stream: it has the counts long before the
    forward's compositor is through (0.4 ms on the stress scene), returns, and the caller's next launches queue up behind the
    running forward.  The check stays synchronous and exact (an arena that was too small is grown and the forward redone before
    anything is returned); what it no longer costs is the idle GPU between two calls (bench.py stress: default mode vs "auto")."""
    import time
    view = probe[1]
    t0 = time.perf_counter()
    while int(view[:V].min()) < 0:
        if time.perf_counter() - t0 > 0.2:      # (counts that never arrive: wait for the stream, look once more)
            torch.cuda.current_stream(dev_index).synchronize()
            if int(view[:V].min()) < 0:
                raise RuntimeError("skelsplat_amd: the binned forward's pair counts did not reach the host")
            break
    return int(view[:V].max())


_PROBES_IN_FLIGHT = 1024     # calls of one shape the host may be ahead of the GPU by (36 bytes of pinned memory each)


def _lazy_probe(cap_key, cap):
    """The lazy capacity check (see forward_views): never synchronises.  A probed call hands sks_forward a pinned int32 buffer
    as `num_rendered_dev`; k_bin_scan stores the call's pair counts there (device-visible host memory: a V-int store, no copy
    launch, no event).  Called in front of every lazy call of the shape: looks at the buffers of EARLIER calls that the GPU has
    been through by now (they complete in call order; -1 = not written yet), raises if one of them needed more pairs than its
    arena held -- that image missed entries; the arena has been grown for the calls to come --, and returns a buffer for this
    call: EVERY eager call is probed (the host runs hundreds of microseconds ahead of the GPU on this path, so up to
    _PROBES_IN_FLIGHT buffers wait to be looked at; waiting for the previous call's counts, as an earlier version did, stalled
    every step).  None only inside a hipGraph capture.  The gradients of an overflowed call are NaN (k_geom_bwd_binned)."""
    V = cap_key[1]
    st = _BIN_PROBE.get(cap_key)
    if st is None:
        st = _BIN_PROBE[cap_key] = _Probes()
    while st.pending and int(st.pending[0][1][:V].min()) >= 0:
        host, view, pcap = st.pending.pop(0)
        pneed = int(view[:V].max())
        st.free.append((host, view))
        if pneed > pcap:
            _BIN_CAP_HINT[cap_key] = int(pneed * 1.25) + 1024
            _BIN_ZOMBIES.extend(st.pending)      # (the GPU may still write them: kept alive, never looked at again)
            del _BIN_PROBE[cap_key]
            raise RuntimeError(f"skelsplat_amd: a previous binned forward of this shape needed {pneed} (Gaussian, tile) pairs "
                               f"per view but its arena held {pcap}: that image missed entries.  The arena has been grown; "
                               "call again (check_capacity=True checks every call synchronously).")
    if torch.cuda.is_current_stream_capturing():
        return None
    if len(st.pending) >= _PROBES_IN_FLIGHT:     # (never reached by a loop that synchronises now and then: every call is probed)
        torch.cuda.synchronize()
        return _lazy_probe(cap_key, cap)
    if st.free:
        host, view = st.free.pop()
    else:
        host = torch.empty((V + 1,), dtype=torch.int32).pin_memory()
        view = host.numpy()
    view[:V] = -1
    st.pending.append((host, view, cap))
    return host


_BIN_ZOMBIES = []


def _scratch_bytes_cached(V, P, C, W, H, cap):
    key = (V, P, C, W, H, cap)
    r = _SCRATCH_BYTES.get(key)
    if r is None:
        r = _SCRATCH_BYTES[key] = _lib.scratch_bytes(V, P, C, W, H, cap)
    return r


_BG_CACHE = {}
_ENV_TUNE = int(os.environ.get("SKS_FWD_TUNE", "0"), 0)   # tuning experiments: extra forward flag bits (include/skelsplat_hip.h)


def _bg_channels(bg, C, dev):
    """The background as C floats, or None when it is absent or all zero (the reference's default `[0, 0, 0]`,
    train.py:112-113): a zero background contributes nothing to the backward (backward.cu:612-615), and passing NULL
    selects the faster kernels.  The reference reads C floats from its 3-float bg tensor (backward.cu:613-614); pad
    with zeros instead.  One host read per (tensor, version), cached."""
    if bg is None or bg.numel() == 0:
        return None
    key = (id(bg), C, str(dev))
    hit = _BG_CACHE.get(key)
    if hit is not None and (hit[0]() is not bg or hit[1] != bg._version):
        hit = None     # another tensor at a recycled id, or modified in place since
    if hit is None:
        if len(_BG_CACHE) > 64:
            _BG_CACHE.clear()
        bgC = None
        if bool((bg != 0).any()):
            bgC = torch.zeros(C, dtype=torch.float32, device=dev)
            k = min(C, bg.numel())
            bgC[:k] = bg.reshape(-1)[:k].to(device=dev, dtype=torch.float32)
        hit = (weakref.ref(bg), bg._version, bgC)
        _BG_CACHE[key] = hit
    return hit[2]


def backward_views(st: ForwardState, means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor,
                   dL_dinvdepth=None, bg=None, want_dfeatures=False, tune_flags=0, workspace=None, want_mean=False,
                   out_means3D=None, plans=None):
    """Raw batched backward: per-view gradients, dict of (V,P,...) tensors (`workspace`: see Workspace).  `want_mean`: also
    "means3D_mean" (P,3), the mean of the joint gradients over the views (train.py:215-217), formed inside the library.
    `out_means3D`: a contiguous fp32 (V,P,3) tensor to receive "means3D" (a view-sharded caller passes the rows of its
    all_gather shard: no copy between the backward and the exchange)."""
    lib = _lib.load()
    key = None
    pkey = None
    if st.chunks is not None:
        if want_mean or out_means3D is not None:
            raise RuntimeError("want_mean / out_means3D are not available beyond SKS_MAX_CHANNELS channels")
        return _backward_views_wide(st, means3D, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg,
                                    want_dfeatures, tune_flags)
    if plans is not None and workspace is None and st.plan_key is not None and out_means3D is None and not want_mean \
            and _sig(dL_dcolor) is not False and (dL_dinvdepth is None or _sig(dL_dinvdepth) is not False):
        # (the autograd path: same parameter tensors as the recorded forward, gradient tensors fresh every call)
        stream = torch._C._cuda_getCurrentRawStream(st.geom.device.index)
        pkey = ("bwd", st.plan_key, dL_dcolor.shape, dL_dinvdepth is None, None if bg is None else (id(bg), bg._version),
                want_dfeatures, tune_flags, stream)
        hit = plans.get(pkey)
        if hit is not None:
            args_t, dev_index, dev, (V, P, C), has_sr = hit
            e = lambda *s_: torch.empty(s_, dtype=torch.float32, device=dev)
            out = dict(means3D=e(V, P, 3), means2D=e(V, P, 3), opacities=e(V, P, 1), cov3D=e(V, P, 6),
                       scales=e(V, P, 3) if has_sr[0] else None, rotations=e(V, P, 4) if has_sr[1] else None,
                       features=e(V, P, C) if want_dfeatures else None)
            args = list(args_t)
            args[18], args[19], args[22], args[23] = st.radii.data_ptr(), st.geom.data_ptr(), dL_dcolor.data_ptr(), _lib.ptr(dL_dinvdepth)
            args[25], args[26], args[27] = out["means3D"].data_ptr(), out["means2D"].data_ptr(), out["opacities"].data_ptr()
            args[28], args[29], args[30], args[31] = _lib.ptr(out["scales"]), _lib.ptr(out["rotations"]), out["cov3D"].data_ptr(), _lib.ptr(out["features"])
            rc = _replay(lib.sks_backward, args, dev_index)
            if rc != 0:
                reset_scratch()
                plans.pop(pkey, None)
            _lib.check(rc, "sks_backward")
            return out
    if workspace is not None:
        workspace.settle()
    if workspace is not None and st.P:
        key = _bwd_key(st, means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg,
                       want_dfeatures, tune_flags, want_mean, out_means3D, torch._C._cuda_getCurrentRawStream(st.geom.device.index))
        plan = workspace._plans.get("bwd")
        if plan is not None and plan[0] == key:
            _, _keep, args, dev_index, result = plan
            rc = _replay(lib.sks_backward, args, dev_index)
            if rc != 0:
                reset_scratch()
                del workspace._plans["bwd"]
            _lib.check(rc, "sks_backward")
            return result
    if st.P == 0:
        dev, V, C = means3D.device, st.views.V, st.C
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        empty = dict(means3D=z(V, 0, 3), means2D=z(V, 0, 3), opacities=z(V, 0, 1), cov3D=z(V, 0, 6), scales=z(V, 0, 3),
                     rotations=z(V, 0, 4), features=z(V, 0, C) if want_dfeatures else None)
        if want_mean:
            empty["means3D_mean"] = z(0, 3)
        return empty
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    V, P, C = st.views.V, st.P, st.C
    W, H = st.views.W, st.views.H
    feat2 = _f32c(features, "features").reshape(P, -1)
    opacities = _f32c(opacities, "opacities")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
    dL_dcolor = _f32c(dL_dcolor, "dL_dout_color")
    dL_dinvdepth = _f32c(dL_dinvdepth, "dL_dout_invdepth")
    if dL_dcolor.numel() != V * C * H * W:
        raise RuntimeError("dL_dout_color has the wrong number of elements")
    bgC = _bg_channels(bg, C, dev)
    if workspace is None:
        e = lambda name, *s: torch.empty(s, dtype=torch.float32, device=dev)
    else:
        e = lambda name, *s: workspace.get(("bwd", name), s, torch.float32, dev)
    out = dict(means3D=e("m3", V, P, 3), means2D=e("m2", V, P, 3), opacities=e("op", V, P, 1), cov3D=e("cov", V, P, 6),
               scales=e("sc", V, P, 3) if scales is not None else None,
               rotations=e("rot", V, P, 4) if rotations is not None else None,
               features=e("feat", V, P, C) if want_dfeatures else None)
    if want_mean:
        out["means3D_mean"] = e("m3mean", P, 3)
    if out_means3D is not None:
        if tuple(out_means3D.shape) != (V, P, 3) or out_means3D.dtype != torch.float32 or not out_means3D.is_contiguous() \
                or out_means3D.device != dev:
            raise ValueError(f"out_means3D must be a contiguous fp32 (V,P,3) = {(V, P, 3)} tensor on {dev}")
        out["means3D"] = out_means3D
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    accum = _accum(dev, stream, V, P, C)
    args = [V, P, C, W, H, st.views.viewmatrix.data_ptr(), st.views.projmatrix.data_ptr(),
            st.views.tanfovx, st.views.tanfovy, _lib.ptr(bgC), _lib.ptr(means3D), _lib.ptr(feat2),
            _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(cov3D_precomp),
            st.scale_modifier, st.flags | int(tune_flags), _lib.ptr(st.radii), st.geom.data_ptr(), _lib.ptr(st.binning),
            st.bin_capacity, dL_dcolor.data_ptr(), _lib.ptr(dL_dinvdepth), accum.data_ptr(),
            _lib.ptr(out["means3D"]), _lib.ptr(out["means2D"]), _lib.ptr(out["opacities"]),
            _lib.ptr(out["scales"]), _lib.ptr(out["rotations"]), _lib.ptr(out["cov3D"]),
            _lib.ptr(out["features"]), _lib.ptr(out.get("means3D_mean")), None]
    rc = _replay(lib.sks_backward, args, dev.index)
    if rc != 0:
        reset_scratch()
    _lib.check(rc, "sks_backward")
    if key is not None and all(sg is not False for sg in key[1:9]) and not torch.cuda.is_current_stream_capturing():
        keep = (st, means3D, feat2, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg, bgC, accum)
        workspace._plans["bwd"] = (key, keep, args, dev.index, out)
    if pkey is not None and st.binning is None:
        plans[pkey] = (list(args), dev.index, dev, (V, P, C), (scales is not None, rotations is not None))
        plans[("keep", pkey)] = (bg, bgC, accum)
    return out


def forward_backward_views(views: ViewBatch, means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor,
                           dL_dinvdepth=None, bg=None, scale_modifier=1.0, antialiasing=False, clamp01=False, want_dfeatures=False,
                           tune_flags=0, workspace=None, want_mean=False, out_means3D=None, overlap=True, join=True,
                           force_binned=False, bin_capacity=None, check_capacity=True):
    """forward_views + backward_views of the same inputs as ONE C-ABI call (sks_forward_backward), for a caller whose upstream
    gradient `dL_dcolor` is complete when the call is made -- it does not depend on the image this call renders.  Returns
    (color, invdepth, radii, state, grads): the same tensors, bit for bit, the two calls return.  On the small path (P <= 256) the
    backward's launches go to the workspace's second stream, ordered behind the geometry kernel, and run BESIDE the dense
    forward (include/skelsplat_hip.h); everything is the caller's in current-stream order when the call returns.  Needs a
    Workspace: its first call of a shape IS the two separate calls (they validate, allocate and record their argument lists), the
    later ones replay both records through the combined entry point.  overlap=False: the combined entry point, one stream.
    join=False: the current stream is NOT made to wait for the backward; the caller enqueues what follows the gradients -- a
    view-sharded step's collective -- on `workspace.aux_stream(dev)` (under the forward, too) and then calls `workspace.join(dev)`.
    (On the paths that do not overlap -- first call, binned path, overlap=False -- the gradients are on the current stream and
    join() is a no-op hand-over in the other direction: aux waits for the current stream first, so the same caller code is right.)"""
    if workspace is None:
        raise ValueError("forward_backward_views needs a Workspace (outputs, scratch and the second stream live there)")
    lib = _lib.load()
    workspace.settle()
    fplan, bplan = workspace._plans.get("fwd"), workspace._plans.get("bwd")
    if fplan is not None and bplan is not None:
        dev_index = fplan[3]
        fkey = _fwd_key(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01,
                        False, force_binned, bin_capacity, tune_flags, check_capacity)
        if fplan[0] == fkey:
            st = fplan[4][3]
            bkey = _bwd_key(st, means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg,
                            want_dfeatures, tune_flags, want_mean, out_means3D, torch._C._cuda_getCurrentRawStream(dev_index))
            if bplan[0] == bkey:
                aux = workspace.aux_stream(dev_index) if (overlap or not join) else None
                # (a binned forward keeps its capacity check: the probe buffer / pinned counts go in front of the call, the
                # synchronous check's wait behind it -- on the binned path the library runs the backward of a view group beside
                # the forward of the next one, SKS_BIN_GROUPS)
                _replay_cap_prepare(workspace, fplan, check_capacity)
                fa, ba = fplan[2], bplan[2]
                no_join = overlap and not join
                args = fa[:24] + [ba[9], ba[22], ba[23], ba[24]] + ba[25:33] + [None, aux.cuda_stream if overlap else None,
                                                                             _lib.SKS_FB_NO_JOIN if no_join else 0]
                args[-3] = torch._C._cuda_getCurrentRawStream(dev_index)
                if torch._C._cuda_getDevice() == dev_index:
                    rc = lib.sks_forward_backward(*args)
                else:
                    with torch.cuda.device(dev_index):
                        rc = lib.sks_forward_backward(*args)
                if rc != 0:
                    reset_scratch()
                    workspace._plans.pop("fwd", None), workspace._plans.pop("bwd", None)
                _lib.check(rc, "sks_forward_backward")
                if no_join:
                    workspace._pending_join = dev_index     # (the caller's workspace.join(); settled by the next call otherwise)
                if not join and not overlap:     # (the gradients were produced on the current stream: aux must see them)
                    aux.wait_stream(torch.cuda.current_stream(dev_index))
                if _replay_cap_finish(workspace, fplan, check_capacity):
                    return fplan[4] + (bplan[4],)
                workspace.settle()     # the arena overflowed (synchronous check): the two calls below grow it and redo the step
    out = forward_views(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01,
                        force_binned=force_binned, bin_capacity=bin_capacity, check_capacity=check_capacity,
                        tune_flags=tune_flags, workspace=workspace)
    g = backward_views(out[3], means3D, features, opacities, scales, rotations, cov3D_precomp, dL_dcolor, dL_dinvdepth, bg,
                       want_dfeatures, tune_flags, workspace, want_mean, out_means3D)
    if not join:
        workspace.aux_stream(means3D.device.index).wait_stream(torch.cuda.current_stream(means3D.device))
    return out + (g,)


# ------------------------------------------------------------------------------------------------------------
# the forward's fill configuration, measured per shape and per kind of caller (no result bit depends on it)
# ------------------------------------------------------------------------------------------------------------
PLAIN_STORES = 0x100     # a tuner candidate's bit 8: plain stores instead of non-temporal ones (SKS_NO_NT_STORES)
# What the tuners try by default: passes per fill block, non-temporal stores -- the kind that streams past the 256 MB Infinity Cache
# at the rate HBM takes writes, whatever buffer it writes.  Plain stores (TUNE_CANDIDATES_WITH_PLAIN, on request) may sit in that
# cache when a call rewrites the same ~cache-sized outputs step after step: the kernel then retires before its bytes are in HBM
# (H36M forward 46 -> 38-41 us) and they drain under whatever runs next -- a cache-assisted figure, not an HBM rate, and 40 %
# slower on outputs beyond the cache's size; round 5 took such candidates at equal step time, which moved the reported kernel
# fraction by 0.15 without moving the step.
TUNE_CANDIDATES = (0, 3, 4, 5)
TUNE_CANDIDATES_WITH_PLAIN = TUNE_CANDIDATES + (PLAIN_STORES | 3, PLAIN_STORES | 4, PLAIN_STORES | 5)
TUNE_CANDIDATES_FRESH = TUNE_CANDIDATES  # outputs in fresh memory every call (the autograd path)
_FILL_BITS = (0xff << 8) | 16            # flag bits a candidate occupies: passes per fill block, SKS_NO_NT_STORES
_FILL_TUNE = {}          # (device, V, P, C, W, H, "workspace" | "fresh") -> flag bits: what forward_views ORs in when the caller sets none
_MEASURING = [False]     # tune_forward is timing candidates: forward_views leaves the flags as given
_FILL_TUNE_LOG = {}      # same key -> {candidate name: median microseconds} of the measurement behind the pick


def _tune_flag_bits(c):
    return ((int(c) & 0xff) << 8) | (_lib.SKS_NO_NT_STORES if int(c) & PLAIN_STORES else 0)


def tune_name(c):
    """A tuner candidate in words ("default (2 passes, non-temporal)", "4 passes, plain stores")."""
    if not c:
        return "default (2 passes, non-temporal stores)"
    return f"{int(c) & 0xff or 2} passes, {'plain' if int(c) & PLAIN_STORES else 'non-temporal'} stores"


def _time_candidates(set_bits, step_fn, candidates, reps, rounds, dev_index, warm=None):
    """Interleaved round-robin timing of `step_fn` under every candidate; {candidate: [microseconds per call, one per round]}.
    `warm` untimed calls first: the measurement is host wall time over device synchronisations, and a process's first calls (cold
    Python paths, clocks still ramping) are host-bound at several times the step's GPU time."""
    import time
    set_bits(_tune_flag_bits(candidates[0]))
    for _ in range(TUNE_WARM if warm is None else warm):
        step_fn()
    times = {c: [] for c in candidates}
    for _ in range(rounds):
        for c in candidates:
            set_bits(_tune_flag_bits(c))
            for _ in range(2):
                step_fn()
            torch.cuda.synchronize(dev_index)
            t0 = time.perf_counter()
            for _ in range(reps):
                step_fn()
            torch.cuda.synchronize(dev_index)
            times[c].append(1e6 * (time.perf_counter() - t0) / reps)
    return times


def _pick(times, candidates, set_bits=None, step_fn=None, reps=8, dev_index=None, confirm=True):
    """The candidate to keep, judged by each one's LEAST disturbed round (a host hiccup only ever adds time).  The library's default
    (the first candidate) stays unless another beats it by more than 2 % -- and, when the caller's step can be re-run, beats it
    again in a confirmation measurement (three more interleaved rounds of just the two): a pick made from one noisy process state
    (round 6 saw a cold bench process read 80 / 85 / 76 / 76 us for a 66 us step and choose five passes: 11 % slower) does not survive
    that.  Returns (best, {candidate: least-disturbed microseconds})."""
    low = {c: min(v) for c, v in times.items()}
    best = min(low, key=low.get)
    if best != candidates[0] and low[best] > 0.98 * low[candidates[0]]:
        best = candidates[0]
    if best != candidates[0] and step_fn is not None and confirm:
        again = _time_candidates(set_bits, step_fn, (candidates[0], best), reps, 3, dev_index, warm=0)
        if min(again[best]) > 0.98 * min(again[candidates[0]]):
            best = candidates[0]
    return best, low


TUNE_WARM = 12      # untimed calls in front of a tuner's measurement (_time_candidates)


def _workspace_tune(workspace, step_fn, candidates=TUNE_CANDIDATES, reps=8, rounds=3, confirm=True):
    """Workspace.tune (see there)."""
    plan = workspace._plans.get("fwd")
    if plan is None or torch.cuda.is_current_stream_capturing():
        return None, {}
    args, dev_index = plan[2], plan[3]
    base = args[16] & ~_FILL_BITS

    def set_bits(bits):
        args[16] = base | bits
    times = _time_candidates(set_bits, step_fn, candidates, reps, rounds, dev_index)
    best, med = _pick(times, candidates, set_bits, step_fn, reps, dev_index, confirm)
    set_bits(_tune_flag_bits(best))
    V, P, C, W, H = args[0], args[1], args[2], args[3], args[4]
    key = (dev_index, V, P, C, W, H, "workspace")
    _FILL_TUNE[key] = _tune_flag_bits(best)
    _FILL_TUNE_LOG[key] = {tune_name(c): round(v, 2) for c, v in med.items()}
    workspace._tuned[plan[0]] = best
    return best, med


def autotune_fill_passes(workspace, step_fn, candidates=TUNE_CANDIDATES, reps=8, rounds=3):
    """Former name of Workspace.tune (kept for callers of the round-5 API)."""
    return _workspace_tune(workspace, step_fn, candidates, reps, rounds)


def tune_forward(views, means3D, features, opacities, scales, rotations, cov3D_precomp, scale_modifier=1.0, antialiasing=False,
                 clamp01=False, tune_flags=0, reps=6, rounds=3, rotate=4):
    """The fill configuration for callers whose outputs are FRESH tensors every call (the autograd path: GaussianRasterizer,
    render_*; MultiViewLoop's dense step): times sks_forward of this very call over `rotate` output sets in turn (memory the
    kernel has not just written), non-temporal candidates only, and keeps the pick for the shape -- forward_views then uses it
    whenever a caller of that shape passes no fill bits of its own.  Small path only (the binned path's fill geometry has its own
    measured default).  ~100 forwards and a few device synchronisations, once per shape.  Returns the flag bits."""
    P = means3D.shape[0]
    dev = means3D.device
    feat2 = features.reshape(P, -1)
    C = feat2.shape[1]
    key = (dev.index, views.V, P, C, views.W, views.H, "fresh")
    if key in _FILL_TUNE:
        return _FILL_TUNE[key]
    if torch.cuda.is_current_stream_capturing():
        return 0
    _FILL_TUNE[key] = 0      # (the measuring calls below must not recurse into a measurement; shapes without one keep the default)
    if P == 0 or P > _lib.SKS_SMALL_P or C > _lib.SKS_MAX_CHANNELS:
        return 0
    wss = [Workspace() for _ in range(rotate)]
    state = {"bits": 0, "i": 0}

    def step():
        ws = wss[state["i"] % rotate]
        state["i"] += 1
        forward_views(views, means3D, feat2, opacities, scales, rotations, cov3D_precomp, scale_modifier, antialiasing, clamp01,
                      tune_flags=(int(tune_flags) & ~_FILL_BITS) | state["bits"], workspace=ws)

    def set_bits(bits):
        state["bits"] = bits
    _MEASURING[0] = True      # (candidate 0 means "no bits": the measuring calls must get exactly what they ask for)
    try:
        with torch.no_grad():
            times = _time_candidates(set_bits, step, TUNE_CANDIDATES_FRESH, reps, rounds, dev.index, warm=2 * rotate)
            best, med = _pick(times, TUNE_CANDIDATES_FRESH, set_bits, step, reps, dev.index)
    finally:
        _MEASURING[0] = False
    _FILL_TUNE[key] = _tune_flag_bits(best)
    _FILL_TUNE_LOG[key] = {tune_name(c): round(v, 2) for c, v in med.items()}
    return _FILL_TUNE[key]


def fill_tuning():
    """What has been measured so far: {(device, V, P, C, W, H, kind): (pick in words, {candidate: median us})}."""
    names = {_tune_flag_bits(c): tune_name(c) for c in TUNE_CANDIDATES_WITH_PLAIN}
    return {k: (names.get(v, hex(v)), _FILL_TUNE_LOG.get(k, {})) for k, v in _FILL_TUNE.items()}


AUTOTUNE = os.environ.get("SKS_AUTOTUNE", "1") != "0"     # the autograd path and the loops measure a shape's fill configuration once


def mean_views(grads, V, world=1, out=None):
    """Mean over the V views of (rows,P,3) joint gradients, summed in view order (train.py:215-217).  world > 1: `grads` is
    what all_gather_into_tensor left for a view-sharded group (world * ceil(V / world) rows, rank-major)."""
    vmax = (V + world - 1) // world
    P = grads.shape[1]
    if grads.dtype != torch.float32 or not grads.is_contiguous() or grads.shape[0] < (V if world == 1 else world * vmax) \
            or grads.shape[2] != 3:
        raise ValueError("mean_views: grads must be a contiguous fp32 (rows,P,3) tensor with a row for every view")
    _need_gpu(grads, "grads")
    if out is None:
        out = torch.empty((P, 3), dtype=torch.float32, device=grads.device)
    rc = _lib.load().sks_mean_views(V, P, grads.data_ptr(), world, out.data_ptr(),
                                    torch._C._cuda_getCurrentRawStream(grads.device.index))
    _lib.check(rc, "sks_mean_views")
    return out


def export_lists(st: ForwardState):
    """Parity/debug: (point_list (V,cap) int32, ranges (V,Tx*Ty,2) int32, num_rendered (V,) int32) of the binned path."""
    lib = _lib.load()
    if st.binning is None:
        raise RuntimeError("lists exist only on the binned path (force_binned=True)")
    dev = st.geom.device
    V, W, H = st.views.V, st.views.W, st.views.H
    NT = ((W + 15) // 16) * ((H + 15) // 16)
    pl = torch.empty((V, max(st.bin_capacity, 1)), dtype=torch.int32, device=dev)
    rg = torch.empty((V, NT, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sks_export_lists(V, W, H, st.binning.data_ptr(), st.bin_capacity, pl.data_ptr(), rg.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sks_export_lists")
    # (a lazily checked call keeps its counts in a pinned host buffer, or nowhere: the ranges say the same -- a view's entries
    # end where its last non-empty tile's list ends)
    nr = rg[..., 1].amax(dim=1) if st.num_rendered_dev is None or not st.num_rendered_dev.is_cuda else st.num_rendered_dev[:V]
    return pl, rg, nr


# ------------------------------------------------------------------------------------------------------------
# autograd: single view (the reference's _RasterizeGaussians, __init__.py:44-141) and multi-view
# ------------------------------------------------------------------------------------------------------------
def _features_of(sh, colors_precomp, P):
    # the kernels read features straight from `sh` as flat (P, C) and need M == 1; colors_precomp is
    # ignored by the reference's kernels -- here it is used when no sh is given (the reference would dereference null).
    if sh is not None and sh.numel() > 0:
        if sh.dim() == 3 and sh.shape[1] != 1:
            raise RuntimeError(f"features must have exactly one SH coefficient (M == 1), got shape {tuple(sh.shape)}")
        return sh, "sh"
    if colors_precomp is not None and colors_precomp.numel() > 0:
        return colors_precomp, "colors"
    raise RuntimeError("no features: provide shs (P,1,C) or colors_precomp (P,C)")


_AUTOGRAD_PLANS = {}     # recorded C-ABI argument blocks of the autograd path (forward_views / backward_views `plans`)
if os.environ.get("SKS_AUTOGRAD_PLANS", "1") == "0":      # A/B runs: every call takes the validating path
    _AUTOGRAD_PLANS = None


class _RasterizeViews(torch.autograd.Function):
    """V views of the same Gaussians; outputs (V,C,H,W), (V,P), (V,1,H,W); gradients summed over views."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, views,
                scale_modifier, antialiasing, clamp01, debug, bg, single, raw=False):
        P = means3D.shape[0] if means3D.dim() == 2 else 0
        feats, src = _features_of(sh, colors_precomp, P)
        # raw: opacities / scales / rotations are the LEAF parameters; the kernels run sigmoid / exp / normalize themselves and
        # the backward returns the leaves' gradients (SKS_RAW_PARAMS | SKS_RAW_GRADS): no activation launches around the call
        tune = (_lib.SKS_RAW_PARAMS | _lib.SKS_RAW_GRADS) if raw else 0
        # check_capacity=True: like the reference, which reads the pair count back on every forward, the autograd path never
        # returns an image (and then gradients) with dropped entries -- a too-small arena is grown and the forward redone
        color, invdepth, radii, st = forward_views(views, means3D, feats, opacities, scales, rotations, cov3Ds_precomp,
                                                   scale_modifier, antialiasing, clamp01, debug, check_capacity=True,
                                                   tune_flags=tune, plans=_AUTOGRAD_PLANS)
        ctx.st, ctx.src, ctx.bg, ctx.single = st, src, bg, single
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # an unused inverse-depth output costs nothing in backward
        if single:
            return color[0], radii[0], invdepth[0]
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_invdepth):
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = ctx.saved_tensors
        st = ctx.st
        feats = sh if ctx.src == "sh" else colors_precomp
        if grad_color is None:
            V, C, H, W = st.views.V, st.C, st.views.H, st.views.W
            grad_color = torch.zeros((V, C, H, W), dtype=torch.float32, device=means3D.device)
        need_feat = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        g = backward_views(st, means3D, feats, opacities, scales, rotations, cov3Ds_precomp, grad_color, grad_invdepth,
                           ctx.bg, want_dfeatures=need_feat, plans=_AUTOGRAD_PLANS)
        red = (lambda t: None if t is None else t[0]) if st.views.V == 1 else (lambda t: None if t is None else t.sum(0))
        gf = red(g["features"])
        grad_sh = gf.reshape(sh.shape) if (gf is not None and ctx.src == "sh") else None
        grad_cp = gf.reshape(colors_precomp.shape) if (gf is not None and ctx.src == "colors") else None
        has_cov = cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0
        return (red(g["means3D"]), red(g["means2D"]), grad_sh, grad_cp, red(g["opacities"]).reshape(opacities.shape),
                red(g["scales"]), red(g["rotations"]), red(g["cov3D"]) if has_cov else None,
                None, None, None, None, None, None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, clamp01=False, raw_params=False):
    """DGR/diff_gaussian_rasterization_h36m/__init__.py:21-42."""
    views = ViewBatch.from_settings(raster_settings)
    return _RasterizeViews.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                 views, raster_settings.scale_modifier, bool(raster_settings.antialiasing), clamp01,
                                 bool(raster_settings.debug), raster_settings.bg, True, bool(raw_params))


def rasterize_views(views: ViewBatch, means3D, means2D, sh, opacities, scales=None, rotations=None, cov3D_precomp=None,
                    colors_precomp=None, scale_modifier=1.0, antialiasing=False, clamp01=False, debug=False, bg=None):
    """Multi-view autograd entry point (no reference counterpart: the reference renders one view per call)."""
    emp = torch.empty(0)
    z = lambda t: emp if t is None else t
    return _RasterizeViews.apply(means3D, z(means2D), z(sh), z(colors_precomp), opacities, z(scales), z(rotations),
                                 z(cov3D_precomp), views, scale_modifier, antialiasing, clamp01, debug, bg, False)


class GaussianRasterizer(nn.Module):
    """DGR/diff_gaussian_rasterization_h36m/__init__.py:158-207.  `num_channels` pins C like the reference's
    compile-time NUM_CHANNELS (config.h:15); None accepts any C (beyond SKS_MAX_CHANNELS = 32 through the generic path of
    _forward_views_wide: 32-channel slices)."""
    num_channels: Optional[int] = None

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            _need_gpu(positions, "positions")
            pos = positions.contiguous().float()
            P = pos.shape[0]
            present = torch.zeros(P, dtype=torch.bool, device=pos.device)
            if P:
                with torch.cuda.device(pos.device):
                    rc = _lib.load().sks_mark_visible(P, pos.data_ptr(), _f32c(rs.viewmatrix, "viewmatrix").data_ptr(),
                                                      _f32c(rs.projmatrix, "projmatrix").data_ptr(), present.data_ptr(),
                                                      torch.cuda.current_stream(pos.device).cuda_stream)
                _lib.check(rc, "sks_mark_visible")
        return present

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, clamp01=False, raw_params=False):
        """The reference's signature plus two extensions: `clamp01` folds render_*'s clamp(0, 1) into the kernels;
        `raw_params`: `opacities`, `scales`, `rotations` are the model's LEAF parameters (_opacity logits, _scaling log-scales,
        raw _rotation, scene/gaussian_model.py:39-47) -- the activations and their Jacobians run inside the kernels."""
        raster_settings = self.raster_settings
        if raster_settings.prefiltered:
            # the reference's kernels TRAP the device when prefiltered is set and a point fails the frustum test
            # (auxiliary.h:166-174: printf + __trap()); render_* always passes False (gaussian_renderer/__init__.py:56)
            raise RuntimeError("raster_settings.prefiltered=True is refused: the reference aborts the device on the first "
                               "culled point in that mode (auxiliary.h:166-174); pass False (INTEGRATION.md)")
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        if self.num_channels is not None:
            f = shs if shs.numel() else colors_precomp
            if f.shape[-1] != self.num_channels:
                raise RuntimeError(f"this rasterizer package is fixed to NUM_CHANNELS={self.num_channels}, "
                                   f"got features with {f.shape[-1]} channels")
        if raw_params and (scales.numel() == 0 or rotations.numel() == 0):
            raise Exception('raw_params needs the scale/rotation pair (the leaves), not a precomputed 3D covariance')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings, clamp01=clamp01, raw_params=raw_params)


def make_package(num_channels):
    """Class pair for one of the reference's three packages (NUM_CHANNELS 17 / 19 / 15)."""
    cls = type(f"GaussianRasterizer{num_channels}", (GaussianRasterizer,), {"num_channels": num_channels})
    return GaussianRasterizationSettings, cls


def decode_geom(st: ForwardState):
    """Debug/parity view of the forward's per-(view, Gaussian) records: dict of conic_opacity (V,P,4),
    xy (V,P,2), depths (V,P), rect (V,P,4) [xmin,ymin,xmax,ymax in tiles]."""
    V, P = st.views.V, st.P
    n = V * P * 16
    seg = (n + 255) // 256 * 256
    g = st.geom
    co = g[0:n].view(torch.float32).reshape(V, P, 4)
    xyd = g[seg:seg + n].view(torch.float32).reshape(V, P, 4)
    rect = g[2 * seg:2 * seg + n].view(torch.int32).reshape(V, P, 4)
    return dict(conic_opacity=co, xy=xyd[..., :2], depths=xyd[..., 2], rect=rect)


# ------------------------------------------------------------------------------------------------------------
# sparse fused training step (sks_gt_tile_stats / sks_geometry / sks_backward_fused_loss)
# ------------------------------------------------------------------------------------------------------------
class GtStats:
    """Per-scene statistics of the constant pseudo-GT heat-maps (V,C,H,W): what the masked-L2 loss sees wherever the
    render is zero.  `offsets` (HOST size_t array or None): views of different sizes -- `gt` is then a flat fp32 buffer
    and offsets[v] the start (in floats) of view v's (C,H_v,W_v) planes (HeatmapSet)."""
    __slots__ = ("gt", "tile_S", "tile_N", "totals", "offsets", "factors")

    def __init__(self):
        self.offsets = None
        self.factors = None      # HeatmapFactors: the heat-maps in separable form, no planes (then gt is None)


class HeatmapFactors:
    """The pseudo-GT heat-maps of V views in SEPARABLE form -- plane(v, j) = (row[v,j][:, None] * col[v,j][None, :] -
    cmin[v,j]) / den[v,j], what heatmaps.heatmap_factors computes -- for the sparse fused step, which evaluates the few
    thousand pixels it needs from the factors (bit for bit the value sks_heatmaps would have stored) instead of reading
    them back from (V,J,H,W) planes nobody else looks at: 68 MB per H36M view never written.  row (V,J,H), col (V,J,W) with
    H, W the LARGEST view (views of different sizes use the leading part of their rows), cmin / den (V,J).
    `totals()` fills a (V,2) fp64 table with each view's {sum gt^2, count gt > 0} (GtStats.totals)."""

    def __init__(self, V, J, W, H, device):
        import ctypes
        self.V, self.J, self.W, self.H = int(V), int(J), int(W), int(H)
        self.row = torch.zeros((V, J, H), dtype=torch.float32, device=device)
        self.col = torch.zeros((V, J, W), dtype=torch.float32, device=device)
        self.cmin = torch.zeros((V, J), dtype=torch.float32, device=device)
        self.den = torch.ones((V, J), dtype=torch.float32, device=device)
        self.ptrs = (ctypes.c_void_p * 4)(self.row.data_ptr(), self.col.data_ptr(), self.cmin.data_ptr(), self.den.data_ptr())

    def totals(self, views, out):
        lib = _lib.load()
        if tuple(out.shape) != (self.V, 2) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("HeatmapFactors.totals: `out` must be a contiguous fp64 (V,2) tensor")
        dev = self.row.device
        with torch.cuda.device(dev):
            rc = lib.sks_heatmap_totals(self.V, self.J, self.W, self.H, self.row.data_ptr(), self.col.data_ptr(),
                                        self.cmin.data_ptr(), self.den.data_ptr(), views.wh, out.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sks_heatmap_totals")
        return out

    def planes(self, v, size=None):
        """View v's (J,H_v,W_v) planes as tensor ops (tests, debugging): the same fp32 expression, same order."""
        w, h = size or (self.W, self.H)
        return (self.row[v, :, :h, None] * self.col[v, :, None, :w] - self.cmin[v, :, None, None]) / self.den[v, :, None, None]


class HeatmapSet:
    """The heat-maps of V views whose image sizes may differ, in ONE flat buffer (so that a single launch can address
    all of them): `planes[v]` is view v's (C,H_v,W_v) tensor, a view into `flat`; views of equal size are adjacent, so
    `group(key)` is a (Vg,C,H,W) tensor for the dense entry points.  offsets: HOST size_t array for the C ABI."""

    def __init__(self, sizes, C, device):
        import ctypes
        self.sizes = [(int(w), int(h)) for w, h in sizes]
        self.C = int(C)
        order = {}
        for v, sz in enumerate(self.sizes):
            order.setdefault(sz, []).append(v)
        self.groups = order                         # (W,H) -> views, in first-appearance order of the sizes
        total = sum(self.C * w * h * len(vs) for (w, h), vs in order.items())
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        off = [0] * len(self.sizes)
        self._group_t = {}
        pos = 0
        for (w, h), vs in order.items():
            n = self.C * w * h
            self._group_t[(w, h)] = self.flat[pos:pos + n * len(vs)].view(len(vs), self.C, h, w)
            for i, v in enumerate(vs):
                off[v] = pos + i * n
            pos += n * len(vs)
        self.offsets_list = off
        self.offsets = (ctypes.c_size_t * len(off))(*off)
        self.planes = [self.flat[off[v]:off[v] + self.C * w * h].view(self.C, h, w) for v, (w, h) in enumerate(self.sizes)]

    def group(self, key):
        return self._group_t[key]

    @classmethod
    def adopt(cls, tensor):
        """A contiguous (V,C,H,W) tensor as a (single-size) set, without a copy."""
        import ctypes
        V, C, H, W = tensor.shape
        self = cls.__new__(cls)
        self.sizes, self.C = [(W, H)] * V, C
        self.groups = {(W, H): list(range(V))}
        self.flat = tensor.view(-1)
        self._group_t = {(W, H): tensor}
        n = C * H * W
        self.offsets_list = [v * n for v in range(V)]
        self.offsets = (ctypes.c_size_t * V)(*self.offsets_list)
        self.planes = [tensor[v] for v in range(V)]
        return self


def gt_tile_stats(gt, out=None, tiles=False):
    """Per-view heat-map totals (what the masked-L2 loss is for an all-zero render); `tiles=True` also fills the per
    (view, tile, channel) arrays.  `out`: a GtStats of the same shape to refill in place (scene streaming keeps every
    pointer stable)."""
    gt = _f32c(gt, "gt")
    V, C, H, W = gt.shape
    NT = ((W + 15) // 16) * ((H + 15) // 16)
    dev = gt.device
    if out is not None:
        if out.totals.shape != (V, 2) or out.totals.device != dev or out.gt.shape != gt.shape or out.offsets is not None:
            raise ValueError("gt_tile_stats: `out` was made for another shape / device")
        st = out
    else:
        st = GtStats()
        st.tile_S = torch.empty((V, NT, C), dtype=torch.float32, device=dev) if tiles else None
        st.tile_N = torch.empty((V, NT, C), dtype=torch.float32, device=dev) if tiles else None
        st.totals = torch.empty((V, 2), dtype=torch.float64, device=dev)
    st.gt = gt
    with torch.cuda.device(dev):
        rc = _lib.load().sks_gt_tile_stats(V, C, W, H, gt.data_ptr(), _lib.ptr(st.tile_S), _lib.ptr(st.tile_N),
                                           st.totals.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sks_gt_tile_stats")
    return st


def geometry_views(views: ViewBatch, means3D, C, opacities, scales, rotations, cov3D_precomp, scale_modifier=1.0,
                   antialiasing=False, raw_params=False, out=None, frames=1):
    """Geometry stage only (no image): returns a ForwardState usable by backward_fused_loss.  raw_params: the three
    tensors are the leaf parameters (_opacity, _scaling, _rotation); activations run in-kernel (SKS_RAW_PARAMS).
    frames > 1: `views` holds frames x Vf views (frame-major) and the parameter tensors are stacked (frames, P, ..):
    view f*Vf + j renders frame f's Gaussians (see sks_loop_fused_step)."""
    lib = _lib.load()
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    frames = int(frames)
    if frames < 1 or views.V % frames:
        raise ValueError(f"frames = {frames} must divide the number of views ({views.V})")
    if frames > 1 and (means3D.dim() != 3 or means3D.shape[0] != frames):
        raise ValueError(f"frames = {frames} needs parameters stacked (frames, P, ..); means3D is {tuple(means3D.shape)}")
    P = means3D.shape[-2]
    opacities = _f32c(opacities, "opacities")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
    V, W, H = views.V, views.W, views.H
    flags = (_lib.SKS_ANTIALIASING if antialiasing else 0) | (_lib.SKS_RAW_PARAMS if raw_params else 0)
    gbytes, _, _ = _lib.scratch_bytes(V, max(P, 1), C, W, H, 0)
    if out is not None and out.P == P and out.C == C and out.views is views and out.flags == flags and out.frames == frames:
        radii, geom = out.radii, out.geom       # refill in place (persistent state of the fused loop step)
    else:
        radii = torch.empty((V, P), dtype=torch.int32, device=dev)
        geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sks_geometry(V, P, C, W, H, views.viewmatrix.data_ptr(), views.projmatrix.data_ptr(), views.tanfovx,
                              views.tanfovy, _lib.ptr(means3D), _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations),
                              _lib.ptr(cov3D_precomp), float(scale_modifier), flags, radii.data_ptr(), geom.data_ptr(),
                              views.wh, frames, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sks_geometry")
    st = ForwardState()
    st.frames = frames
    st.views, st.P, st.C, st.flags, st.scale_modifier = views, P, C, flags, float(scale_modifier)
    st.geom, st.binning, st.bin_capacity, st.radii, st.num_rendered_dev = geom, None, 0, radii, None
    return st


def _check_heatmaps(views, C, stats):
    """The heat-maps must be what the views address: one (V,C,H,W) tensor, or (mixed sizes) a HeatmapSet's flat buffer,
    or (factored) a HeatmapFactors of the views' largest size."""
    if stats.factors is not None:
        f = stats.factors
        if (f.V, f.J, f.W, f.H) != (views.V, C, views.W, views.H):
            raise RuntimeError(f"heat-map factors {(f.V, f.J, f.W, f.H)} do not match the views {(views.V, C, views.W, views.H)}")
        return
    if views.mixed:
        if stats.offsets is None or len(stats.offsets) != views.V:
            raise RuntimeError("views of different sizes need heat-maps in one flat buffer with per-view offsets (HeatmapSet)")
        need = max(int(o) + C * w * h for o, (w, h) in zip(stats.offsets, views.sizes))
        if stats.gt.numel() < need:
            raise RuntimeError(f"heat-map buffer of {stats.gt.numel()} floats is too small for the views ({need})")
    elif stats.offsets is None and tuple(stats.gt.shape) != (views.V, C, views.H, views.W):
        raise RuntimeError(f"heat-maps {tuple(stats.gt.shape)} do not match the views {(views.V, C, views.H, views.W)}")


def loop_fused_step(st: ForwardState, stats: GtStats, features, packed, sums, slots, group_mask, last_view, xyz, scaling,
                    rotation, opacity, exp_avg, exp_avg_sq, counters, acc_steps, lr_sched, lrs, adam, lambda_consistency, limb):
    """sks_loop_fused_step: fused-loss compositing backward + (geometry backward, Adam step, geometry forward of the
    updated parameters) for one accumulation group; `st` must describe the current parameters and is left describing the
    updated ones.  lr_sched / lrs / adam / limb: ctypes arrays as for sks_loop_adam_step.  A state made with
    geometry_views(frames=F) steps F independent frames at once (stacked parameter / moment / slot / counter tensors)."""
    lib = _lib.load()
    dev = xyz.device
    V, P, C = st.views.V, st.P, st.C
    W, H = st.views.W, st.views.H
    _check_heatmaps(st.views, C, stats)
    feat2 = _f32c(features, "features").reshape(P, -1)
    stream = torch.cuda.current_stream(dev).cuda_stream
    accum = _accum(dev, stream, V, P, C)
    with torch.cuda.device(dev):
        rc = lib.sks_loop_fused_step(V, P, C, W, H, st.views.viewmatrix.data_ptr(), st.views.projmatrix.data_ptr(),
                                     st.views.tanfovx, st.views.tanfovy, feat2.data_ptr(), st.scale_modifier, st.flags,
                                     st.radii.data_ptr(), st.geom.data_ptr(), _lib.ptr(stats.gt), stats.totals.data_ptr(),
                                     accum.data_ptr(), sums.data_ptr(), packed.data_ptr(), slots.data_ptr(), group_mask,
                                     last_view, xyz.data_ptr(), scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(),
                                     exp_avg.data_ptr(), exp_avg_sq.data_ptr(), counters.data_ptr(), acc_steps, lr_sched, lrs,
                                     adam, float(lambda_consistency), limb, st.views.wh, stats.offsets,
                                     st.frames, None if stats.factors is None else stats.factors.ptrs, stream)
    _lib.check(rc, "sks_loop_fused_step")


def backward_fused_loss(st: ForwardState, stats: GtStats, means3D, features, opacities, scales, rotations, cov3D_precomp,
                        bg=None, packed_out=None, sums_out=None):
    """Render + clamp + masked-L2 + backward on the covered tiles only.  Returns (grads dict of (V,P,..) UNSCALED
    gradients, loss_sums (V,2) f64 = per-view {S, N}); the true gradient is grads / N_v, loss_v = S_v / N_v."""
    lib = _lib.load()
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    V, P, C = st.views.V, st.P, st.C
    W, H = st.views.W, st.views.H
    _check_heatmaps(st.views, C, stats)
    feat2 = _f32c(features, "features").reshape(P, -1)
    opacities = _f32c(opacities, "opacities")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
    bgC = _bg_channels(bg, C, dev)
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out = dict(means3D=e(V, P, 3), means2D=e(V, P, 3), opacities=e(V, P, 1), cov3D=e(V, P, 6),
               scales=e(V, P, 3) if scales is not None else None, rotations=e(V, P, 4) if rotations is not None else None)
    if sums_out is None:
        sums = torch.empty((V, 2), dtype=torch.float64, device=dev)
    else:
        sums = sums_out[:V]
        if sums.shape != (V, 2) or sums.dtype != torch.float64 or not sums.is_contiguous():
            raise ValueError("sums_out must be a contiguous fp64 tensor with at least V rows of 2")
    if packed_out is not None and (tuple(packed_out.shape) != (V, P, 11) or not packed_out.is_contiguous()):
        raise ValueError(f"packed_out must be a contiguous (V,P,11) = {(V, P, 11)} tensor")
    stream = torch.cuda.current_stream(dev).cuda_stream
    accum = _accum(dev, stream, V, P, C)
    with torch.cuda.device(dev):
        rc = lib.sks_backward_fused_loss(V, P, C, W, H, st.views.viewmatrix.data_ptr(), st.views.projmatrix.data_ptr(),
                                         st.views.tanfovx, st.views.tanfovy, _lib.ptr(bgC), _lib.ptr(means3D), _lib.ptr(feat2),
                                         _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(cov3D_precomp),
                                         st.scale_modifier, st.flags, st.radii.data_ptr(), st.geom.data_ptr(),
                                         _lib.ptr(stats.gt), _lib.ptr(stats.tile_S), _lib.ptr(stats.tile_N),
                                         stats.totals.data_ptr(), accum.data_ptr(), _lib.ptr(out["means3D"]),
                                         _lib.ptr(out["means2D"]), _lib.ptr(out["opacities"]), _lib.ptr(out["scales"]),
                                         _lib.ptr(out["rotations"]), _lib.ptr(out["cov3D"]), sums.data_ptr(),
                                         _lib.ptr(packed_out), st.views.wh, stats.offsets,
                                         None if stats.factors is None else stats.factors.ptrs, stream)
    _lib.check(rc, "sks_backward_fused_loss")
    return out, sums
