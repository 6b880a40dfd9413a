"""Multi-view optimisation loop of one scene (reference: train.py:130-222), MI355X layout.

What the reference does per iteration: pick view (iteration-1) % V, render it, masked-L2 against that view's heat-map
plus lambda * limb-symmetry loss, autograd.grad wrt (xyz, _scaling, _rotation, _opacity); store the xyz gradient in
slot `view` of a V-slot buffer, overwrite the other three .grad with this view's; every `accumulation_steps`
iterations: xyz.grad = mean over the V slots, Adam step.

Because the parameters only change at those steps, the views of one accumulation group are independent given the
parameters.  Here a group is ONE batched forward + ONE batched backward launch sequence (blockIdx.z = view), and
with torch.distributed the views are sharded over ranks (view v -> rank v % world): each rank renders its views and a
single all_gather of the (V_local, P, 11) per-view parameter gradients over RCCL rebuilds the V slots in view order
on every rank, so the mean uses the reference's summation order and every rank takes the identical Adam step
(no parameter broadcast).  Nothing in the group synchronises with the host.
"""
import numpy as np

import torch
import torch.distributed as dist

from . import rasterizer as R
from .scene import DATASETS


def l2_loss_gaussian(rendering, gt_heatmap):
    """utils/loss_utils.py:86-100 ('mean' reduction): mean squared error over pixels where gt > 0 or rendering > 0."""
    mask = (gt_heatmap > 0) | (rendering > 0)
    error = (rendering - gt_heatmap) ** 2
    return error[mask].mean(), error


def limb_3d_consistency_loss(xyz, dataset):
    """utils/loss_utils.py:226-250."""
    (la0, la1), (ra0, ra1), (ll0, ll1), (rl0, rl1) = DATASETS[dataset]["limbs"]
    l_arm = torch.norm(xyz[la0] - xyz[la1], dim=-1)
    r_arm = torch.norm(xyz[ra0] - xyz[ra1], dim=-1)
    l_leg = torch.norm(xyz[ll0] - xyz[ll1], dim=-1)
    r_leg = torch.norm(xyz[rl0] - xyz[rl1], dim=-1)
    return torch.norm(l_arm - r_arm) + torch.norm(l_leg - r_leg)


def masked_l2_grad_torch(render, gt):
    """Plain-tensor-op statement of the masked-L2 gradient (device-side, no host sync); render, gt: (V,C,H,W).
    Returns (dL, per-view loss, per-view scale): the true gradient is dL * scale[v]."""
    mask = (gt > 0) | (render > 0)
    diff = render - gt
    n = mask.sum(dim=(1, 2, 3)).clamp_min(1).to(render.dtype)
    loss = (diff * diff * mask).sum(dim=(1, 2, 3)) / n
    dL = 2.0 * diff * mask
    return dL, loss, 1.0 / n


def activation_chain(gm, g):
    """Per-view gradients wrt the activated tensors -> wrt the raw parameters (what autograd does through
    exp / normalize / sigmoid in gaussian_model.py:39-47, 102-131).  g: dict of (V,P,...) tensors."""
    s = gm.get_scaling.detach()
    o = gm.get_opacity.detach()
    raw_q = gm._rotation.detach()
    nrm = raw_q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    q = raw_q / nrm
    d_scaling = g["scales"] * s[None]
    d_opacity = g["opacities"] * (o * (1 - o))[None]
    gq = g["rotations"]
    d_rotation = (gq - q[None] * (q[None] * gq).sum(-1, keepdim=True)) / nrm[None]
    return d_scaling, d_rotation, d_opacity


class OptEarlyStopping:
    """utils/general_utils.py:467-491: stop when the last `window_size` losses repeat the `window_size` before them to
    within `repeat_tolerance` (compared as fp32 tensors, like the reference's torch.tensor(list) of loss.item())."""

    def __init__(self, window_size=4, repeat_tolerance=1e-6):
        self.window_size = window_size
        self.repeat_tolerance = repeat_tolerance
        self.loss_history = []

    def __call__(self, current_loss):
        self.loss_history.append(current_loss)
        if len(self.loss_history) < 2 * self.window_size:
            return False
        w1 = torch.tensor(self.loss_history[-2 * self.window_size:-self.window_size])
        w2 = torch.tensor(self.loss_history[-self.window_size:])
        return bool(torch.all(torch.abs(w1 - w2) < self.repeat_tolerance))


class NotStopping:
    """utils/general_utils.py:493-498."""

    def __call__(self, current_loss):
        return False


early_stopping_strategy = {"opt_early_stopping": OptEarlyStopping, "no_stopping": NotStopping}   # utils/__init__.py:31-34


class MultiViewLoop:
    """One scene.  `heatmaps`: (V,C,H,W) pseudo-GT on this rank's device, or a list of V (C,H_v,W_v) tensors when the
    cameras differ in size (only the local views are read).
    `loss_grad`: callable (render, gt) -> (dL_unscaled, per-view loss, per-view scale); default: the fused HIP kernel
    (ops.masked_l2_grad_fused); loop.masked_l2_grad_torch is the same thing in tensor ops.
    `shard_views=False`: this process runs all V views itself even when torch.distributed is initialised (frame
    sharding: every rank optimises its own frames, no communication at all -- SURVEY §8e axis 2).
    `early_stopping`: a key of `early_stopping_strategy` (configs/*.yaml `training.early_stopping`), an OptEarlyStopping
    instance, or a callable loss -> bool.  The reference's criterion (OptEarlyStopping, window <= 16) runs ON THE DEVICE, inside
    the optimiser kernel (sks_loop_adam_step_es): no loss is read back, the group stays a fixed launch sequence (use_graph works,
    view-sharded ranks decide identically from the gathered sums -- they ride in the step's one all_gather), and the host learns
    the stopping iteration from a pinned flag it polls without waiting (run() synchronises once, at its end).  Any other
    callable is a host decision: one read-back per group, no use_graph."""

    def __init__(self, gaussians, cameras, heatmaps, dataset="h36m", accumulation_steps=4, lambda_consistency=1e-5,
                 bg=None, antialiasing=False, loss_grad=None, group=None, view_grad_fn=None, device_tail=None,
                 use_graph=False, sparse=None, fused_tail=None, shard_views=True, graph_collectives=None,
                 early_stopping="no_stopping"):
        import os
        self.gm = gaussians
        self.dataset = dataset
        self.V = len(cameras)
        self.acc_steps = int(accumulation_steps)
        self.lambda_consistency = float(lambda_consistency)
        self.antialiasing = antialiasing
        self.group = group
        sharded = bool(shard_views) and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if sharded else 1
        self.rank = dist.get_rank(group) if sharded else 0
        # the exchange step runs whenever views are sharded over a process group -- also a group of ONE rank, which is how
        # the single-GPU tests drive the very code path 8 GPUs take (RCCL all_gather included)
        self.exchange = sharded
        dev = gaussians._xyz.device
        self.device = dev
        self.local_ids = [v for v in range(self.V) if v % self.world == self.rank]
        # view_grad_fn(loop) -> ((V_local,P,11) raw-parameter gradients, per-view losses) replaces the HIP path;
        # only the multi-process CPU tests use it (gloo has no GPU), the product path is _local_view_grads.
        self.view_grad_fn = view_grad_fn
        self.cameras = cameras
        P = gaussians._xyz.shape[0]
        self.P = P
        Vl = len(self.local_ids)
        # Local heat-maps live in ONE flat buffer (rasterizer.HeatmapSet): views of one size are adjacent -- a (Vg,C,H,W)
        # tensor for the dense entry points -- and the sparse fused step addresses all of them, whatever their sizes,
        # through per-view offsets in a single launch (H36M mixes 1000x1000 and 1002x1000 sensors, quirk Q11).
        sizes = [(int(cameras[v].image_width), int(cameras[v].image_height)) for v in self.local_ids]
        if torch.is_tensor(heatmaps) and heatmaps.dim() == 4 and Vl == self.V and heatmaps.is_contiguous() \
                and heatmaps.dtype == torch.float32 and len(set(sizes)) == 1:
            self.hset = R.HeatmapSet.adopt(heatmaps)            # all views local, one size: no copy
        elif Vl:
            C_hm = int(heatmaps[self.local_ids[0]].shape[0])
            self.hset = R.HeatmapSet(sizes, C_hm, heatmaps[self.local_ids[0]].device)
            for k, v in enumerate(self.local_ids):
                self.hset.planes[k].copy_(heatmaps[v])
        else:
            self.hset = None
        # size groups: local slots per image size, each one batched launch sequence of the DENSE path (and of the
        # per-frame heat-map generation); entries: [slots, ViewBatch, gt (Vg,C,H,W), GtStats or None, slot index tensor]
        self.size_groups = []
        if self.hset is not None:
            for key, slots in self.hset.groups.items():
                vb = (R.ViewBatch.from_cameras([cameras[self.local_ids[k]] for k in slots]) if view_grad_fn is None else None)
                idx = torch.tensor(slots, dtype=torch.long, device=dev)
                self.size_groups.append([slots, vb, self.hset.group(key), None, idx])
        single = len(self.size_groups) == 1
        self.gt = self.size_groups[0][2] if single else None   # single-size convenience (tests, view_grad_fn)
        self.views = self.size_groups[0][1] if single else None
        self.bg = bg
        default_loss = loss_grad is None
        if loss_grad is None and view_grad_fn is None:
            from .ops import masked_l2_grad_fused
            loss_grad = masked_l2_grad_fused
        self.loss_grad = loss_grad
        # V-slot buffer of per-view xyz gradients (train.py:121); persists across groups (quirk Q8)
        self.accumulated_grads = torch.zeros((self.V, P, 3), device=dev)
        self.iteration = 0
        self.last_losses = None
        self.stopped_at = None       # iteration at which early stopping ended the scene (train.py:155,227-233)
        self._es_groups = 0          # groups enqueued since the scene began (view-sharded early stopping: step_group)
        # all_gather needs equal shard sizes: pad every rank to ceil(V / world) views
        self.vmax = (self.V + self.world - 1) // self.world
        # device-side tail (sks_loop_pack_grads + sks_loop_adam_step): default whenever the HIP path is used with the
        # fused loss; the tensor-op tail + torch.optim.Adam below is the same algorithm and stays for custom losses
        if device_tail is None:
            device_tail = view_grad_fn is None and default_loss and dev.type == "cuda" and P <= 256
        self.device_tail = bool(device_tail)
        if callable(early_stopping):
            self.early_stopping = early_stopping
        else:
            self.early_stopping = early_stopping_strategy[early_stopping]()
        self._stopping = not isinstance(self.early_stopping, NotStopping)
        # the reference's criterion on the device (sks_loop_adam_step_es); anything else is a host decision per group
        self._es_device = (self._stopping and type(self.early_stopping) is OptEarlyStopping and self.device_tail
                           and 1 <= self.early_stopping.window_size <= 16 and not self.early_stopping.loss_history)
        if self._stopping and not self._es_device and use_graph:
            raise ValueError("a custom early-stopping callable reads every iteration's loss on the host (train.py:155): "
                             "use_graph must be False")
        # hipGraph capture of a group that contains the RCCL all_gather: opt-in (graph_collectives=True or
        # SKS_GRAPH_COLLECTIVES=1); the sequence itself is fixed and allocation-free either way
        if graph_collectives is None:
            graph_collectives = os.environ.get("SKS_GRAPH_COLLECTIVES") == "1"
        self.use_graph = bool(use_graph) and self.device_tail and (not self.exchange or bool(graph_collectives))
        self._graph = None
        self._graphs = {}
        # sparse fused step: render + clamp + masked-L2 + backward only on the tiles some Gaussian rect covers, using
        # per-view statistics of the constant heat-maps (sks_gt_tile_stats); no dense image / gradient is ever written
        if sparse is None:
            sparse = self.device_tail and P <= 64
        self.sparse = bool(sparse) and self.device_tail and P <= 64
        self.views_all = None        # sparse path: ALL local views in one batch (sizes may differ)
        self.stats_all = None
        if self.sparse and Vl:
            for grp in self.size_groups:
                grp[3] = R.gt_tile_stats(grp[2])
            if single:
                self.views_all, self.stats_all = self.size_groups[0][1], self.size_groups[0][3]
            else:
                self.views_all = R.ViewBatch.from_cameras([cameras[v] for v in self.local_ids], allow_mixed=True)
                st = R.GtStats()
                st.gt, st.tile_S, st.tile_N = self.hset.flat, None, None
                st.totals = torch.empty((Vl, 2), dtype=torch.float64, device=dev)
                st.offsets = self.hset.offsets
                self.stats_all = st
                self._merge_totals()
        if self.device_tail:
            import ctypes
            cfg = gaussians.opt_cfg
            self._sched = (ctypes.c_double * 5)(cfg["lr_init"], cfg["lr_final"], cfg["lr_delay_mult"],
                                                float(cfg["lr_delay_steps"]), float(cfg["lr_max_steps"]))
            self._lrs = (ctypes.c_double * 3)(cfg["lr_scaling"], cfg["lr_rotation"], cfg["lr_opacity"])
            self._adam = (ctypes.c_double * 3)(cfg["betas"][0], cfg["betas"][1], cfg["eps"])
            limbs = [i for pair in DATASETS[dataset]["limbs"] for i in pair]
            self._limb = (ctypes.c_int * 8)(*limbs) if self.lambda_consistency != 0.0 else None
            self.exp_avg = torch.zeros((P, 11), device=dev)
            self.exp_avg_sq = torch.zeros((P, 11), device=dev)
            self.counters = torch.zeros(2, dtype=torch.int32, device=dev)
            # persistent buffers of the group (allocated here, never inside a graph capture): this rank's packed
            # raw-parameter gradients -- with the exchange padded to vmax rows, the pad rows stay zero for ever -- and what
            # all_gather_into_tensor leaves, which sks_loop_adam_step reads in place (rank-major layout, `shard_world`)
            self._es_state = self._es_flag = None
            if self._es_device:
                w = self.early_stopping.window_size
                self._es_state = torch.zeros(2 + 2 * w, dtype=torch.int32, device=dev)
                self._es_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._es_flag_np = self._es_flag.numpy()
            if self.exchange and self._es_device:
                # a rank's block = its vmax x P x 11 gradient rows (padded to an even float count) + its views' {S, N} doubles:
                # gradients and losses cross in ONE all_gather, every rank runs the same criterion (sks_loop_shard_floats)
                from . import _lib
                nfl = int(_lib.load().sks_loop_shard_floats(self.V, P, self.world))
                tail = nfl - 4 * self.vmax
                self._shard_flat = torch.zeros(nfl, device=dev)
                self._shard = self._shard_flat[:self.vmax * P * 11].view(self.vmax, P, 11)
                self._sums = self._shard_flat[tail:].view(torch.float64).view(self.vmax, 2)
                self._allg = torch.empty(self.world * nfl, device=dev)
            else:
                self._shard_flat = None
                self._shard = torch.zeros((self.vmax if self.exchange else max(Vl, 1), P, 11), device=dev)
                self._allg = torch.empty((self.world * self.vmax, P, 11), device=dev) if self.exchange else None
                self._sums = torch.zeros((max(Vl, 1), 2), dtype=torch.float64, device=dev)
            self._sums_all = (torch.zeros((self.world * self.vmax, 2), dtype=torch.float64, device=dev)
                              if self.exchange and self._stopping and not self._es_device else None)
            self._direct = None
            if self.exchange:
                # RCCL builds its communicator on the first collective: do that here, eagerly, never inside a graph
                # capture or a timed step (the gathered rows are overwritten by every group)
                dist.all_gather_into_tensor(self._allg, self._shard if self._shard_flat is None else self._shard_flat, group=self.group)
                # ... and, when asked for (SKS_RCCL_DIRECT=1), a communicator of our own, so that the group's one all_gather is
                # enqueued on the stream its neighbours run on (torch's process group runs it on an internal stream: two event
                # hand-overs, ~7 us of the GPU timeline per step at world 1); None by default and when the backend is not RCCL
                if dev.type == "cuda":
                    from .rccl_direct import DirectGather
                    self._direct = DirectGather.create(dev, self.group)
        # (without the device tail -- a custom loss_grad / view_grad_fn -- the criterion is a host decision per group: the views'
        # losses ride in the gradients' all_gather as one more column, every rank feeds the same numbers in iteration order)
        # the dense step renders into fresh tensors every group: its forward's fill configuration is measured once per image size
        # (rasterizer.tune_forward; forward_views then launches with the pick) -- the library's own tuner, no caller-side knob
        if view_grad_fn is None and dev.type == "cuda" and not self.sparse and R.AUTOTUNE and P <= 256:
            with torch.no_grad():
                feats = gaussians.get_features.reshape(P, -1)
                for slots, vb, gt, _, idx in self.size_groups:
                    R.tune_forward(vb, gaussians._xyz.detach(), feats, gaussians.get_opacity.detach(), gaussians.get_scaling.detach(),
                                   gaussians.get_rotation.detach(), None, antialiasing=self.antialiasing, clamp01=True)
        # one GPU, sparse step: the whole group is two launches (sks_loop_fused_step); the geometry of the current
        # parameters lives in a persistent state that every step leaves up to date for the next one
        # (the single-workgroup tail walks the views four at a time: a win for a handful of views -- H36M's 4 --, a loss
        # for Panoptic's 31, where the one-block-per-view kernels stay)
        self.fused_tail = (self.sparse and not self.exchange and Vl > 0 and bg is None and not self._stopping
                           and (fused_tail is True or (fused_tail is None and self.V <= 8)))
        self._fstate = None          # persistent ForwardState (geom + radii) of the fused tail
        self._geom_valid = False     # does it describe the current parameters?
        if self.fused_tail:          # persistent buffers are allocated here, never inside a graph capture
            with torch.no_grad():
                self._fstate = R.geometry_views(self.views_all, gaussians._xyz.detach(),
                                                gaussians.get_features.reshape(P, -1).shape[1], gaussians._opacity,
                                                gaussians._scaling, gaussians._rotation, None,
                                                antialiasing=self.antialiasing, raw_params=True)
            self._fbuf = (self._shard, self._sums)

    def _merge_totals(self):
        """Mixed sizes: the per-size-group heat-map totals -> the (V_local,2) table of the all-views batch."""
        for slots, vb, gt, stats, idx in self.size_groups:
            self.stats_all.totals.index_copy_(0, idx, stats.totals)

    # -- scene streaming ---------------------------------------------------------------------------------------
    def new_scene(self, points, poses_2d=None, heatmaps=None, dropout=False):
        """Next frame seen by the SAME cameras (the reference's outer loop, train.py:74-99: new GaussianModel, new
        heat-maps, iteration counter back to 0).  Everything is re-initialised in place -- parameters, Adam moments,
        step counters, V-slot buffer, heat-maps and their tile statistics keep their storage -- so the hipGraphs
        captured for the previous frame are replayed as they are.  Give either `poses_2d` (V,J,2) (the heat-maps are
        generated from the re-initialised Gaussians like general_utils.py:175-304) or ready `heatmaps`."""
        from .heatmaps import generate_heatmaps
        gm = self.gm
        gm.reset_from_points(points)
        with torch.no_grad():
            self.accumulated_grads.zero_()
            if self.device_tail:
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
                self.counters.zero_()
            elif gm.optimizer is not None:
                gm.training_setup()
            drop = None
            if dropout and poses_2d is not None:
                from .heatmaps import draw_dropout
                drop = draw_dropout(self.V, self.P)
                if self.exchange and self.world > 1:
                    # the reference makes ONE draw shared by all cameras (general_utils.py:267-283); the ranks' default
                    # generators are not synchronised, so rank 0's draw is the scene's
                    buf = drop.to(device=self.device, dtype=torch.uint8)
                    dist.broadcast(buf, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group)
                    drop = buf.to(device="cpu", dtype=torch.bool)
            for grp in self.size_groups:
                slots, vb, gt, stats, idx = grp
                ids = [self.local_ids[k] for k in slots]
                if heatmaps is not None:
                    for i, v in enumerate(ids):
                        gt[i].copy_(heatmaps[v])
                elif poses_2d is not None:
                    p2d = torch.as_tensor(poses_2d, device=self.device)[ids]
                    # the planes and (when the sparse path wants them) their per-view totals in one pass
                    fused = stats is not None and stats.tile_S is None
                    generate_heatmaps(gm._xyz.detach(), gm.get_scaling.detach(), gm._rotation.detach(), p2d,
                                      [self.cameras[v] for v in ids], out=gt, views=vb,
                                      totals=stats.totals if fused else None,
                                      drop_mask=None if drop is None else drop[ids])
                    if fused:
                        continue
                else:
                    raise ValueError("new_scene needs poses_2d or heatmaps")
                if stats is not None:
                    R.gt_tile_stats(gt, out=stats)
            if self.sparse and len(self.size_groups) > 1:
                self._merge_totals()
        if self.device_tail and self._es_device:
            torch.cuda.current_stream(self.device).synchronize()    # (nothing of the last scene may still write the flag)
            self._es_state.zero_()
            self._es_flag_np[0] = 0
        elif isinstance(self.early_stopping, (OptEarlyStopping, NotStopping)):
            self.early_stopping = type(self.early_stopping)()
        self.stopped_at = None
        self._es_groups = 0
        self._geom_valid = False
        self.iteration = 0    # (last_losses keeps pointing at the buffers the captured graphs write)
        return self

    # -- one accumulation group --------------------------------------------------------------------------
    def _local_view_grads(self):
        """Renders this rank's views and returns (V_local, P, 11) raw-parameter gradients + per-view losses."""
        gm = self.gm
        P = self.P
        with torch.no_grad():
            means = gm._xyz.detach()
            feats = gm.get_features.reshape(P, -1)
            opac = gm.get_opacity.detach()
            scales = gm.get_scaling.detach()
            quats = gm.get_rotation.detach()
            packed = torch.empty((len(self.local_ids), P, 11), device=means.device)
            losses = torch.empty(len(self.local_ids), device=means.device)
            for slots, vb, gt, _, idx in self.size_groups:
                color, inv, radii, st = R.forward_views(vb, means, feats, opac, scales, quats, None,
                                                        antialiasing=self.antialiasing, clamp01=True)
                dL, lv, scale = self.loss_grad(color, gt)
                g = R.backward_views(st, means, feats, opac, scales, quats, None, dL, None, bg=self.bg)
                d_scaling, d_rotation, d_opacity = activation_chain(gm, g)
                pk = torch.cat([g["means3D"], d_scaling, d_rotation, d_opacity], dim=-1)  # (Vg, P, 11)
                packed.index_copy_(0, idx, pk * scale[:, None, None])   # 1 / N_mask of each view (the backward is linear in dL)
                losses.index_copy_(0, idx, lv.to(losses.dtype))
        return packed, losses

    def _consistency_grad(self):
        xyz = self.gm._xyz.detach().clone().requires_grad_(True)
        loss = limb_3d_consistency_loss(xyz, self.dataset) * self.lambda_consistency
        (gx,) = torch.autograd.grad(loss, xyz)
        return gx, loss.detach()

    # -- device-side tail ----------------------------------------------------------------------------------
    def _device_grads(self):
        """This rank's views -> self._shard[:V_local] (packed raw-parameter gradients) and self._sums ({S, N} per view):
        a fixed, allocation-light launch sequence (sparse: sks_geometry + sks_backward_fused_loss for ALL local views,
        whatever their sizes; dense: forward + masked-L2 + backward + pack per size group)."""
        from . import _lib
        from .ops import masked_l2
        lib = _lib.load()
        gm, P, dev = self.gm, self.P, self.device
        Vl = len(self.local_ids)
        stream = torch.cuda.current_stream(dev).cuda_stream
        means = gm._xyz.detach()
        feats = gm.get_features.reshape(P, -1)
        packed = self._shard[:Vl]
        if self.sparse:
            # leaf parameters straight into the kernels: activations, their Jacobians and the 1/N scale all run inside
            # sks_geometry / sks_backward_fused_loss (SKS_RAW_PARAMS)
            st = R.geometry_views(self.views_all, means, feats.shape[1], gm._opacity, gm._scaling, gm._rotation, None,
                                  antialiasing=self.antialiasing, raw_params=True)
            R.backward_fused_loss(st, self.stats_all, means, feats, gm._opacity, gm._scaling, gm._rotation, None,
                                  bg=self.bg, packed_out=packed, sums_out=self._sums)
            return
        opac, scales, quats = gm.get_opacity.detach(), gm.get_scaling.detach(), gm.get_rotation.detach()
        single = len(self.size_groups) == 1
        for slots, vb, gt, stats, idx in self.size_groups:
            color, inv, radii, st = R.forward_views(vb, means, feats, opac, scales, quats, None,
                                                    antialiasing=self.antialiasing, clamp01=True)
            dL, S, N = masked_l2(color, gt)
            g = R.backward_views(st, means, feats, opac, scales, quats, None, dL, None, bg=self.bg)
            sums = torch.stack([S, N], dim=1).contiguous()
            Vg = len(slots)
            pk = packed if single else torch.empty((Vg, P, 11), device=dev)
            _lib.check(lib.sks_loop_pack_grads(Vg, P, g["means3D"].data_ptr(), g["scales"].data_ptr(),
                                               g["rotations"].data_ptr(), g["opacities"].data_ptr(),
                                               gm._scaling.data_ptr(), gm._rotation.data_ptr(), gm._opacity.data_ptr(),
                                               sums.data_ptr(), pk.data_ptr(), stream), "sks_loop_pack_grads")
            if not single:
                packed.index_copy_(0, idx, pk)
            self._sums[:Vl].index_copy_(0, idx, sums)

    def _device_adam(self, group_mask, last_view, n_iters):
        """[all_gather of the shards ->] sks_loop_adam_step on the view-major (one rank) or rank-major (gathered) table."""
        from . import _lib
        lib = _lib.load()
        gm, dev = self.gm, self.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.exchange:
            # every rank needs every view's gradients (train.py:175, 215-218): ONE all_gather of the padded shards over
            # RCCL; the optimiser kernel reads the gathered buffer in place (view v = row (v % world) * vmax + v // world)
            src = self._shard if self._shard_flat is None else self._shard_flat
            if self._direct is not None:
                self._direct.all_gather_into_tensor(self._allg, src)     # RCCL on THIS stream (rccl_direct.py)
            else:
                dist.all_gather_into_tensor(self._allg, src, group=self.group)
            full, world = self._allg, self.world
        else:
            full, world = self._shard, 1
        if self._es_device:
            es = self.early_stopping
            _lib.check(lib.sks_loop_adam_step_es(self.V, self.P, full.data_ptr(), self.accumulated_grads.data_ptr(), group_mask,
                                                 last_view, gm._xyz.data_ptr(), gm._scaling.data_ptr(), gm._rotation.data_ptr(),
                                                 gm._opacity.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                 self.counters.data_ptr(), n_iters, self._sched, self._lrs, self._adam,
                                                 float(self.lambda_consistency), self._limb, world,
                                                 None if world > 1 else self._sums.data_ptr(), self._es_state.data_ptr(),
                                                 int(es.window_size), float(es.repeat_tolerance), self._es_flag.data_ptr(), stream),
                       "sks_loop_adam_step_es")
            return
        _lib.check(lib.sks_loop_adam_step(self.V, self.P, full.data_ptr(), self.accumulated_grads.data_ptr(), group_mask,
                                          last_view, gm._xyz.data_ptr(), gm._scaling.data_ptr(), gm._rotation.data_ptr(),
                                          gm._opacity.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                          self.counters.data_ptr(), n_iters, self._sched, self._lrs, self._adam,
                                          float(self.lambda_consistency), self._limb, world, stream), "sks_loop_adam_step")

    def _device_group(self, group_mask, last_view, n_iters):
        """forward -> fused masked-L2 -> backward -> pack -> [all_gather] -> Adam, all enqueued, no host sync."""
        gm, P = self.gm, self.P
        with torch.no_grad():
            if self.fused_tail:
                feats = gm.get_features.reshape(P, -1)
                if not self._geom_valid:
                    self._fstate = R.geometry_views(self.views_all, gm._xyz.detach(), feats.shape[1], gm._opacity, gm._scaling,
                                                    gm._rotation, None, antialiasing=self.antialiasing, raw_params=True,
                                                    out=self._fstate)
                    self._geom_valid = True
                packed, sums = self._fbuf
                R.loop_fused_step(self._fstate, self.stats_all, feats, packed, sums, self.accumulated_grads, group_mask,
                                  last_view, gm._xyz, gm._scaling, gm._rotation, gm._opacity, self.exp_avg, self.exp_avg_sq,
                                  self.counters, n_iters, self._sched, self._lrs, self._adam, float(self.lambda_consistency),
                                  self._limb)
                self.last_losses = (sums[:, 0], sums[:, 1])
                return
            if self.local_ids:
                self._device_grads()
                Vl = len(self.local_ids)
                self.last_losses = (self._sums[:Vl, 0], self._sums[:Vl, 1])
            if self._stopping and not self._es_device:
                group_mask, last_view, n_iters = self._early_stop_cut(group_mask, last_view, n_iters)
            self._device_adam(group_mask, last_view, n_iters)

    ES_SYNC_GROUPS = 8      # view-sharded ranks look at the device criterion's flag every this many groups (step_group)

    def _poll_stop(self, wait=False):
        """Device-side criterion: has it fired?  The kernel stores the stopping iteration into pinned host memory; `wait`
        first lets the stream drain (run() does, once, when it has enqueued everything it was asked for)."""
        if wait:
            torch.cuda.current_stream(self.device).synchronize()
        it = int(self._es_flag_np[0])
        if it:
            self.stopped_at = it
            self.iteration = it
        return self.stopped_at

    def _early_stop_cut(self, group_mask, last_view, n_iters):
        """train.py:155-233 with the group's views batched: feed the criterion the losses of the group's iterations in
        order; if it fires at the k-th, only the first k views' slots are refreshed, view k's scaling / rotation /
        opacity gradients win, the optimiser steps at once and the scene ends.  One host sync per group."""
        Vl = len(self.local_ids)
        if self.exchange:
            pad = torch.zeros((self.vmax, 2), dtype=torch.float64, device=self.device)
            pad[:Vl] = self._sums[:Vl]
            dist.all_gather_into_tensor(self._sums_all, pad, group=self.group)
            rows = [(v % self.world) * self.vmax + v // self.world for v in range(self.V)]
            sums = self._sums_all[rows].cpu()
        else:
            sums = self._sums[:Vl].cpu()
        l2 = (sums[:, 0] / sums[:, 1].clamp_min(1.0)).to(torch.float32)
        cons = torch.zeros((), dtype=torch.float32)
        if self.lambda_consistency != 0.0:
            cons = (limb_3d_consistency_loss(self.gm._xyz.detach(), self.dataset) * self.lambda_consistency).float().cpu()
        it0 = self.iteration + 1
        mask = 0
        for k in range(n_iters):
            v = (it0 + k - 1) % self.V
            mask |= 1 << v
            if self.early_stopping(float(l2[v] + cons)):
                self.stopped_at = it0 + k
                return mask, v, k + 1
        return group_mask, last_view, n_iters

    def step_group(self, parameters_untouched=False):
        """Runs iterations self.iteration+1 .. up to the next optimiser step (train.py:130-222).
        parameters_untouched=True: the caller states that nothing has written the parameters since this object's previous
        step_group() -- the fused step's tail left the geometry of the updated parameters behind, and this step then starts from
        it instead of launching a geometry pass of its own (what run() does between its own consecutive groups, and what a
        captured graph of several groups does inside).  There is no way to see a write through `.data` or a raw pointer from here,
        so the default is to recompute."""
        gm = self.gm
        it0 = self.iteration + 1
        it1 = it0
        while it1 % self.acc_steps != 0:
            it1 += 1
        view_of_iter = [(it - 1) % self.V for it in range(it0, it1 + 1)]   # train.py:136-138
        if self.stopped_at is not None:
            return self.iteration
        if self.device_tail:
            mask = 0
            for v in view_of_iter:
                mask |= 1 << v
            key = (mask, view_of_iter[-1], it1 - it0 + 1)
            if self.use_graph:
                # (two graphs per group shape: one that first refreshes the geometry from the parameters, one -- replayed on the
                # caller's word, parameters_untouched -- that starts from what the previous group's tail left)
                chained = bool(parameters_untouched and self.fused_tail and self._geom_valid)
                gkey = key + (chained,)
                if self._graphs.get(gkey) is None:
                    # capture one group; replays advance the device counters themselves
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        if not chained:
                            self._geom_valid = False    # (see run(): a replay never assumes who ran before it)
                        self._device_group(*key)
                    self._graphs[gkey] = graph          # the capture itself does not execute: replay below
                self._graphs[gkey].replay()
                self._graph = (key, self._graphs[gkey])
            else:
                if not parameters_untouched:
                    self._geom_valid = False    # eager steps never assume the parameters were left untouched since the last one
                self._device_group(*key)
            if self._es_device:
                if self.exchange and self.world > 1:
                    # Sharded: every group holds a collective, so every rank must enqueue the SAME number of groups.  A free-running
                    # poll would not give that -- each host runs ahead of its GPU by its own amount and would see the flag a
                    # different number of groups late.  So the flag is only looked at behind a stream synchronisation, every
                    # ES_SYNC_GROUPS-th group: there the state the criterion ran on (gathered sums, es_state) is the same on
                    # every rank, and so is what each of them reads.
                    self._es_groups += 1
                    if self._es_groups % self.ES_SYNC_GROUPS == 0:
                        self._poll_stop(wait=True)
                else:
                    self._poll_stop()       # (never waits: the groups enqueued behind a stop do nothing to the parameters)
            self.iteration = it1 if self.stopped_at is None else self.stopped_at
            return self.iteration
        if not self.local_ids:
            packed, losses = None, None
        elif self.view_grad_fn is not None:
            packed, losses = self.view_grad_fn(self)
        else:
            packed, losses = self._local_view_grads()
        dev = self.device
        n11 = self.P * 11
        if self.exchange:
            # ONE collective per group here too: a rank's block = its views' gradient rows, each followed by that view's loss
            shard = torch.zeros((self.vmax, n11 + 1), device=dev)
            if packed is not None:
                shard[:packed.shape[0], :n11] = packed.reshape(packed.shape[0], n11)
                shard[:packed.shape[0], n11] = losses.to(shard.dtype)
            allg = torch.empty((self.world * self.vmax, n11 + 1), device=dev)
            dist.all_gather_into_tensor(allg, shard, group=self.group)
            # rank r, slot k  <->  view r + k * world: one precomputed row index, view order
            if getattr(self, "_rows", None) is None:
                self._rows = torch.tensor([(v % self.world) * self.vmax + v // self.world for v in range(self.V)],
                                          dtype=torch.long, device=dev)
            rows = allg.index_select(0, self._rows)
            full, losses_all = rows[:, :n11].reshape(self.V, self.P, 11), rows[:, n11]
        else:
            full, losses_all = packed, losses
        gcons, lcons = self._consistency_grad() if self.lambda_consistency != 0.0 else (0.0, None)
        if self._stopping:
            # train.py:155-233 with the group's views batched: the criterion sees the losses of the group's iterations in order;
            # if it fires at the k-th, only the first k views' slots are refreshed, view k's scaling / rotation / opacity
            # gradients win, the optimiser steps at once and the scene ends (the same numbers on every rank: the same decision)
            l2 = losses_all.detach().to(torch.float32).cpu()
            cons = torch.zeros((), dtype=torch.float32) if lcons is None else lcons.detach().to(torch.float32).cpu()
            for k, v in enumerate(view_of_iter):
                if self.early_stopping(float(l2[v] + cons)):
                    self.stopped_at = it0 + k
                    view_of_iter, it1 = view_of_iter[:k + 1], it0 + k
                    break
        # every view's loss contains the consistency term, so every slot carries its gradient (train.py:150-152,175)
        for v in dict.fromkeys(view_of_iter):
            self.accumulated_grads[v] = full[v, :, 0:3] + gcons
        last = view_of_iter[-1]                                            # quirk Q7: last view's grads win
        if gm._xyz.grad is None:
            for p in (gm._xyz, gm._scaling, gm._rotation, gm._opacity):
                p.grad = torch.zeros_like(p)
        gm._scaling.grad = full[last, :, 3:6].contiguous()
        gm._rotation.grad = full[last, :, 6:10].contiguous()
        gm._opacity.grad = full[last, :, 10:11].contiguous()
        gm._xyz.grad = self.accumulated_grads.mean(dim=0)                  # train.py:215-218
        gm.update_learning_rate(it1)                                       # quirk Q9: schedule indexed by iteration
        with torch.no_grad():
            gm.optimizer.step()
            gm.optimizer.zero_grad(set_to_none=True)
        self.iteration = it1
        self.last_losses = losses
        return it1

    def run(self, iterations=500, groups_per_graph=25):
        """Runs the loop up to `iterations`.  With use_graph, `groups_per_graph` consecutive accumulation groups are
        captured into ONE hipGraph (the step has no host state: counters, LR schedule and Adam live on the device), so
        a 500-iteration scene is a handful of graph launches."""
        if self.use_graph and self.acc_steps % self.V == 0 and self.iteration % self.acc_steps == 0 \
                and self.stopped_at is None:
            mask = (1 << self.V) - 1
            key = (mask, (self.acc_steps - 1) % self.V, self.acc_steps)
            remaining = (iterations - self.iteration) // self.acc_steps
            G = min(int(groups_per_graph), remaining)
            if G > 1:
                if getattr(self, "_multi", None) is None or self._multi[0] != (key, G):
                    self._device_group(*key)            # one eager group: warms allocations, counts as a real step
                    self.iteration += self.acc_steps
                    remaining -= 1
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._geom_valid = False    # every replay starts by refreshing the geometry (new_scene, eager steps)
                        for _ in range(G):
                            self._device_group(*key)
                    self._multi = ((key, G), graph)
                while remaining >= G and self.stopped_at is None:
                    self._multi[1].replay()
                    self.iteration += G * self.acc_steps
                    remaining -= G
                    if self._es_device:
                        self._poll_stop(wait=self.exchange and self.world > 1)     # (sharded: see step_group)
        chained = False       # (between run()'s own consecutive groups nobody else touches the parameters)
        while self.iteration < iterations and self.stopped_at is None:
            self.step_group(parameters_untouched=chained)
            chained = True
        if self.device_tail and self._es_device and self.stopped_at is None:
            self._poll_stop(wait=True)      # the ONE synchronisation of a scene with the criterion on the device
        return self.gm._xyz.detach()


class FrameBatchLoop:
    """F independent frames seen by the SAME cameras, optimised side by side in the sparse fused step.

    The reference optimises one frame after the other (train.py:74-99: a fresh GaussianModel and fresh heat-maps per
    scene, nothing carried over), and one 17-Gaussian skeleton keeps only a fraction of an MI355X busy: the fused
    render+loss+backward kernel of a 4-view group launches 4 x 17 x 16 workgroups for 256 CUs, and the single-workgroup
    tail (geometry backward, Adam, next geometry) runs on ONE.  Frames being independent, F of them go through the same
    two launches: `sks_loop_fused_step(frames=F)` renders F x V views and steps F optimisers, one workgroup per frame.
    Every frame's trajectory is bit-identical to a MultiViewLoop running it alone (tests/test_ops_gpu.py).

    `gaussians`: a GaussianModel of ONE frame after training_setup() -- the template for the initial scaling /
    rotation / opacity, the one-hot features and the optimiser's configuration; its own tensors are not touched.
    Parameters live stacked: xyz / scaling (F,P,3), rotation (F,P,4), opacity (F,P,1).  F x V <= 64 views per launch.
    Early stopping (a per-frame host decision, one sync per group) is MultiViewLoop's business, not this class's."""

    def __init__(self, gaussians, cameras, frames, dataset="h36m", accumulation_steps=4, lambda_consistency=1e-5,
                 antialiasing=False, use_graph=False, factored=True):
        import ctypes
        gm = gaussians
        self.gm = gm
        self.dataset = dataset
        self.cameras = cameras
        self.F, self.V = int(frames), len(cameras)
        F, V = self.F, self.V
        if F < 1 or F * V > R._lib.SKS_MAX_VIEWS:
            raise ValueError(f"frames x views = {F} x {V} exceeds the {R._lib.SKS_MAX_VIEWS} views of one launch")
        P = gm._xyz.shape[0]
        if P > 64:
            raise ValueError("frame batching runs the sparse fused step: P <= 64 Gaussians per frame")
        self.P = P
        dev = gm._xyz.device
        self.device = dev
        self.acc_steps = int(accumulation_steps)
        self.lambda_consistency = float(lambda_consistency)
        self.antialiasing = antialiasing
        self.use_graph = bool(use_graph)
        self.features = gm.get_features.detach().reshape(P, -1).contiguous()
        self.C = self.features.shape[1]
        init = gm._initial
        self._init = (init[0].reshape(1, P, 3), init[1].reshape(1, P, 4), init[2].reshape(1, P, 1))
        self.xyz = gm._xyz.detach().reshape(1, P, 3).repeat(F, 1, 1).contiguous()
        self.scaling = self._init[0].repeat(F, 1, 1).contiguous()
        self.rotation = self._init[1].repeat(F, 1, 1).contiguous()
        self.opacity = self._init[2].repeat(F, 1, 1).contiguous()
        self.exp_avg = torch.zeros((F, P, 11), device=dev)
        self.exp_avg_sq = torch.zeros((F, P, 11), device=dev)
        self.counters = torch.zeros((F, 2), dtype=torch.int32, device=dev)
        self.accumulated_grads = torch.zeros((F, V, P, 3), device=dev)    # train.py:121, one V-slot buffer per frame
        self._packed = torch.zeros((F * V, P, 11), device=dev)
        self._sums = torch.zeros((F * V, 2), dtype=torch.float64, device=dev)
        cfg = gm.opt_cfg
        self._sched = (ctypes.c_double * 5)(cfg["lr_init"], cfg["lr_final"], cfg["lr_delay_mult"],
                                            float(cfg["lr_delay_steps"]), float(cfg["lr_max_steps"]))
        self._lrs = (ctypes.c_double * 3)(cfg["lr_scaling"], cfg["lr_rotation"], cfg["lr_opacity"])
        self._adam = (ctypes.c_double * 3)(cfg["betas"][0], cfg["betas"][1], cfg["eps"])
        limbs = [i for pair in DATASETS[dataset]["limbs"] for i in pair]
        self._limb = (ctypes.c_int * 8)(*limbs) if self.lambda_consistency != 0.0 else None
        cams_all = [cameras[k % V] for k in range(F * V)]
        self._cams_all = cams_all
        sizes = [(int(c.image_width), int(c.image_height)) for c in cams_all]
        # factored (default): the pseudo-GT stays in its separable form (rasterizer.HeatmapFactors) -- the fused step
        # evaluates the pixels it needs, the per-view loss constants come from the factors (sks_heatmap_totals), and no
        # heat-map plane is ever written: 68 MB per H36M view and a frame's largest memory pass are gone
        self.factored = bool(factored)
        self.hset, self.size_groups, self.factors = None, [], None
        if self.factored:
            self.views_all = R.ViewBatch.from_cameras(cams_all, allow_mixed=True)
            self.factors = R.HeatmapFactors(F * V, self.C, self.views_all.W, self.views_all.H, dev)
            st = R.GtStats()
            st.gt, st.tile_S, st.tile_N = None, None, None
            st.totals = torch.zeros((F * V, 2), dtype=torch.float64, device=dev)
            st.factors = self.factors
            self.stats_all = st
        else:
            # planes of all F x V views (frame-major) in one flat buffer; views of one size adjacent (H36M: two sizes)
            self.hset = R.HeatmapSet(sizes, self.C, dev)
            # [slots (frame-major), ViewBatch of them, (Vg,C,H,W) planes, GtStats, slot index tensor]
            for key, slots in self.hset.groups.items():
                vb = R.ViewBatch.from_cameras([cams_all[k] for k in slots])
                gt = self.hset.group(key)
                gt.zero_()
                self.size_groups.append([slots, vb, gt, R.gt_tile_stats(gt), torch.tensor(slots, dtype=torch.long, device=dev)])
            if len(self.size_groups) == 1:
                self.views_all, self.stats_all = self.size_groups[0][1], self.size_groups[0][3]
            else:
                self.views_all = R.ViewBatch.from_cameras(cams_all, allow_mixed=True)
                st = R.GtStats()
                st.gt, st.tile_S, st.tile_N = self.hset.flat, None, None
                st.totals = torch.zeros((F * V, 2), dtype=torch.float64, device=dev)
                st.offsets = self.hset.offsets
                self.stats_all = st
        with torch.no_grad():
            self._fstate = R.geometry_views(self.views_all, self.xyz, self.C, self.opacity, self.scaling, self.rotation, None,
                                            antialiasing=antialiasing, raw_params=True, frames=F)
        self._geom_valid = False
        self._graph = None
        self._multi = None
        self.iteration = 0
        self.last_losses = None

    def new_scenes(self, points, poses_2d=None, heatmaps=None, drop_masks=None):
        """The next F frames: `points` (F,P,3) initial joints; either `poses_2d` (F,V,J,2) -- the heat-maps are generated
        from the re-initialised Gaussians like general_utils.py:175-304, all frames in two launches per image size -- or
        ready `heatmaps` (F,V,C,H,W) / a list of F lists of V (C,H_v,W_v) planes.  `drop_masks`: optional (F,V,J) bool of
        dropped heat-map planes (heatmaps.draw_dropout per frame).  Everything is re-initialised in place, so captured
        hipGraphs are replayed as they are."""
        from .heatmaps import generate_heatmaps, heatmap_factors
        F, V, P = self.F, self.V, self.P
        if self.factored and heatmaps is not None:
            raise ValueError("ready heat-map planes need FrameBatchLoop(..., factored=False)")
        with torch.no_grad():
            pts = points if torch.is_tensor(points) else torch.as_tensor(np.asarray(points))
            if tuple(pts.shape) != (F, P, 3):
                raise ValueError(f"points must be (F,P,3) = {(F, P, 3)}, got {tuple(pts.shape)}")
            self.xyz.copy_(pts.to(device=self.device, dtype=torch.float32))
            self.scaling.copy_(self._init[0].expand(F, P, 3))
            self.rotation.copy_(self._init[1].expand(F, P, 4))
            self.opacity.copy_(self._init[2].expand(F, P, 1))
            self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.counters.zero_(); self.accumulated_grads.zero_()
            if poses_2d is not None:
                p2d_all = torch.as_tensor(poses_2d, device=self.device).reshape(F * V, -1, 2)
                drop_all = None if drop_masks is None else torch.as_tensor(drop_masks).reshape(F * V, -1)
            elif heatmaps is None:
                raise ValueError("new_scenes needs poses_2d or heatmaps")
            if self.factored:
                # two small launches for all frames and image sizes: the factors, then the loss constants from them
                heatmap_factors(self.xyz, torch.exp(self.scaling), self.rotation, p2d_all, self._cams_all,
                                views=self.views_all, frames=F, out=self.factors, drop_mask=drop_all)
                self.factors.totals(self.views_all, self.stats_all.totals)
            for slots, vb, gt, stats, idx in self.size_groups:
                if heatmaps is not None:
                    for i, k in enumerate(slots):
                        gt[i].copy_(heatmaps[k // V][k % V])
                    R.gt_tile_stats(gt, out=stats)
                else:
                    generate_heatmaps(self.xyz, torch.exp(self.scaling), self.rotation, p2d_all[idx],
                                      [self.cameras[k % V] for k in slots], out=gt, views=vb, totals=stats.totals,
                                      drop_mask=None if drop_all is None else drop_all[idx.cpu()], frames=F)
                if len(self.size_groups) > 1:
                    self.stats_all.totals.index_copy_(0, idx, stats.totals)
        self._geom_valid = False
        self.iteration = 0
        return self

    def _device_group(self, group_mask, last_view, n_iters):
        with torch.no_grad():
            if not self._geom_valid:
                self._fstate = R.geometry_views(self.views_all, self.xyz, self.C, self.opacity, self.scaling, self.rotation,
                                                None, antialiasing=self.antialiasing, raw_params=True, out=self._fstate,
                                                frames=self.F)
                self._geom_valid = True
            R.loop_fused_step(self._fstate, self.stats_all, self.features, self._packed, self._sums, self.accumulated_grads,
                              group_mask, last_view, self.xyz, self.scaling, self.rotation, self.opacity, self.exp_avg,
                              self.exp_avg_sq, self.counters, n_iters, self._sched, self._lrs, self._adam,
                              self.lambda_consistency, self._limb)
        s = self._sums.view(self.F, self.V, 2)
        self.last_losses = (s[..., 0], s[..., 1])       # per (frame, view) {S, N}: loss = S / N

    def step_group(self, parameters_untouched=False):
        """(parameters_untouched: as MultiViewLoop.step_group)
        Iterations self.iteration+1 .. the next optimiser step of EVERY frame (train.py:130-222; the frames share the
        iteration counter, the view order and therefore the group's view mask)."""
        it0 = self.iteration + 1
        it1 = it0
        while it1 % self.acc_steps != 0:
            it1 += 1
        view_of_iter = [(it - 1) % self.V for it in range(it0, it1 + 1)]
        mask = 0
        for v in view_of_iter:
            mask |= 1 << v
        key = (mask, view_of_iter[-1], it1 - it0 + 1)
        if self.use_graph:
            if self._graph is None or self._graph[0] != key:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._geom_valid = False
                    self._device_group(*key)
                self._graph = (key, graph)
            self._graph[1].replay()
        else:
            if not parameters_untouched:
                self._geom_valid = False
            self._device_group(*key)
        self.iteration = it1
        return it1

    def run(self, iterations=500, groups_per_graph=25):
        """All F frames up to `iterations`; returns the joints (F,P,3).  With use_graph, `groups_per_graph` groups are one
        hipGraph, as in MultiViewLoop.run."""
        if self.use_graph and self.acc_steps % self.V == 0 and self.iteration % self.acc_steps == 0:
            key = ((1 << self.V) - 1, (self.acc_steps - 1) % self.V, self.acc_steps)
            remaining = (iterations - self.iteration) // self.acc_steps
            G = min(int(groups_per_graph), remaining)
            if G > 1:
                if self._multi is None or self._multi[0] != (key, G):
                    self._geom_valid = False
                    self._device_group(*key)
                    self.iteration += self.acc_steps
                    remaining -= 1
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._geom_valid = False
                        for _ in range(G):
                            self._device_group(*key)
                    self._multi = ((key, G), graph)
                while remaining >= G:
                    self._multi[1].replay()
                    self.iteration += G * self.acc_steps
                    remaining -= G
        chained = False
        while self.iteration < iterations:
            self.step_group(parameters_untouched=chained)
            chained = True
        return self.xyz

    def optimize_sequence(self, points, poses_2d, iterations=500, groups_per_graph=25):
        """The reference's outer loop over the frames of a sequence (train.py:74-99) F frames at a time: `points`
        (N,P,3) initial joints and `poses_2d` (N,V,J,2) detections of N frames -> (N,P,3) optimised joints.  A last batch
        with fewer than F frames is filled up by repeating its final frame (frames are independent: the filler changes
        nothing and is dropped)."""
        pts = points if torch.is_tensor(points) else torch.as_tensor(np.asarray(points))
        p2d = poses_2d if torch.is_tensor(poses_2d) else torch.as_tensor(np.asarray(poses_2d))
        N, F = pts.shape[0], self.F
        if p2d.shape[0] != N:
            raise ValueError(f"{N} frames of points, {p2d.shape[0]} of detections")
        out = torch.empty((N, self.P, 3), dtype=torch.float32, device=self.device)
        for b in range(0, N, F):
            if b + F <= N:
                self.new_scenes(pts[b:b + F], poses_2d=p2d[b:b + F])
            else:
                idx = [min(b + i, N - 1) for i in range(F)]
                self.new_scenes(pts[idx], poses_2d=p2d[idx])
            res = self.run(iterations, groups_per_graph)
            out[b:min(b + F, N)] = res[:min(F, N - b)]
        return out


class FramePipeline:
    """A sequence of frames through `streams` FrameBatchLoops of `frames` frames each, one HIP stream per loop.

    Inside one FrameBatchLoop a group is two dependent launches: the compositing backward of all its frames, then one
    workgroup per frame for geometry backward + Adam + next geometry (~14 us during which the chip is nearly idle).  Several
    loops on separate streams fill that hole with each other's backward kernels: on one MI355X (H36M, 4 views, 500
    iterations per frame, heat-map generation included) 292 frames/s one frame at a time, 1 480 with 16 frames per launch,
    1 900-2 100 with 2-4 such loops in flight (tools/bench_frames.py).  Results do not depend on `frames` / `streams`:
    every frame's trajectory is bit-identical to a MultiViewLoop running it alone."""

    def __init__(self, gaussians, cameras, frames=16, streams=2, **kw):
        kw.setdefault("use_graph", True)
        self.loops = [FrameBatchLoop(gaussians, cameras, frames, **kw) for _ in range(int(streams))]
        self.device = self.loops[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in self.loops]
        self.F, self.P = self.loops[0].F, self.loops[0].P

    def optimize_sequence(self, points, poses_2d, iterations=500, groups_per_graph=25, interleave=100):
        """(N,P,3) initial joints + (N,V,J,2) detections -> (N,P,3) optimised joints (train.py:74-99 over the frames).
        The loops' graph launches are issued round-robin, `interleave` iterations at a time, so that every stream always
        has work queued; a last batch with fewer than `frames` frames is padded by repeating its final frame."""
        pts = points if torch.is_tensor(points) else torch.as_tensor(np.asarray(points))
        p2d = poses_2d if torch.is_tensor(poses_2d) else torch.as_tensor(np.asarray(poses_2d))
        N, F, S = pts.shape[0], self.F, len(self.loops)
        if p2d.shape[0] != N:
            raise ValueError(f"{N} frames of points, {p2d.shape[0]} of detections")
        out = torch.empty((N, self.P, 3), dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        starts = list(range(0, N, F))
        for st in self.streams:
            st.wait_stream(cur)                      # inputs and `out` were produced on the caller's stream
        for w in range(0, len(starts), S):
            active = list(zip(self.loops, self.streams, starts[w:w + S]))
            for fb, st, b in active:
                with torch.cuda.stream(st):
                    if b + F <= N:      # a full batch: views of the inputs, no gather
                        fb.new_scenes(pts[b:b + F], poses_2d=p2d[b:b + F])
                    else:
                        idx = [min(b + i, N - 1) for i in range(F)]
                        fb.new_scenes(pts[idx], poses_2d=p2d[idx])
            for k in range(0, iterations, max(int(interleave), 1)):
                for fb, st, b in active:
                    with torch.cuda.stream(st):
                        fb.run(min(iterations, k + interleave), groups_per_graph)
            for fb, st, b in active:
                with torch.cuda.stream(st):
                    out[b:min(b + F, N)] = fb.xyz[:min(F, N - b)]
        for st in self.streams:
            cur.wait_stream(st)
        return out


def mpjpe(pred, gt):
    """eval.py:123-124: mean Euclidean joint error (same units as the inputs, mm)."""
    return torch.norm(pred - gt, dim=1).mean()
