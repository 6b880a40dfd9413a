"""torch.optim.Adam whose step() is ONE launch (sks_adam_multi) for all parameter groups.

The reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over six one-tensor parameter groups (scene/gaussian_model.py:
203-218) and steps it once per accumulation group (train.py:219-220).  torch's foreach implementation spends ~330 us of host time
and ~40 launches on those few hundred floats; here the same update -- same state layout (`step`, `exp_avg`, `exp_avg_sq` per
parameter, so `state_dict()` / `load_state_dict()` / the reference's `capture()` and `restore()` keep working), same hyper-
parameters read from `param_groups` at every step (the reference rewrites `lr` of the "xyz" group every iteration,
gaussian_model.py:234-239) -- is one C-ABI call.  Drop-in: `self.optimizer = skelsplat_amd.optim.Adam(l, lr=0.0, eps=1e-15)`.

What is not this kernel's is REFUSED, not handed to another implementation: amsgrad / weight decay / maximize / capturable /
differentiable at construction (NotImplementedError), CPU or non-fp32 parameters, sparse or strided gradients at step()
(RuntimeError) -- `torch.optim.Adam` is the class for those.  There is no fall-back path."""
import ctypes as C

import torch

from . import _lib

_MAX = 8    # tensors per sks_adam_multi call


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("foreach", None), kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=False, **kw)
        self._reset_cache()

    def _reset_cache(self):
        self._plans = {}    # first index of a chunk -> (key, argument arrays)

    # The step count of a parameter lives where torch keeps it: state[p]["step"], a host tensor, read and advanced at every step
    # (~1 us per parameter).  It therefore travels with the state dict: the reference's replace_tensor_to_optimizer /
    # _prune_optimizer / cat_tensors_to_optimizer (scene/gaussian_model.py:341-404) move a parameter's state to a NEW nn.Parameter,
    # and code that reads optimizer.state[p]["step"] sees the current value.
    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._reset_cache()
        return r

    @staticmethod
    def _check_group(group):
        bad = [k for k in ("amsgrad", "maximize", "capturable", "differentiable") if group.get(k)]
        if group.get("weight_decay", 0) != 0:
            bad.append("weight_decay")
        if isinstance(group.get("lr"), torch.Tensor):
            bad.append("a tensor lr")
        if bad:
            raise NotImplementedError(f"skelsplat_amd.optim.Adam: {', '.join(bad)} not supported (sks_adam_multi is the plain update "
                                      f"of scene/gaussian_model.py:218); use torch.optim.Adam")

    def add_param_group(self, group):
        r = super().add_param_group(group)
        self._check_group(self.param_groups[-1])
        if hasattr(self, "_plans"):
            self._plans = {}
        return r

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        todo = []       # (param, grad, group)
        dev = betas = eps = None
        for group in self.param_groups:
            first = True
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if first:
                    first = False
                    self._check_group(group)     # (param_groups is the caller's to rewrite between steps)
                    if dev is None:
                        dev, betas, eps = p.device, group["betas"], group["eps"]
                    if group["betas"] != betas or group["eps"] != eps:
                        raise NotImplementedError("skelsplat_amd.optim.Adam: one (betas, eps) for all parameter groups")
                if not (p.is_cuda and p.device == dev and g.device == dev and p.dtype == torch.float32 and g.dtype == torch.float32
                        and g.layout == torch.strided and p.is_contiguous() and g.is_contiguous()):
                    raise RuntimeError("skelsplat_amd.optim.Adam: contiguous fp32 parameters and gradients on ONE ROCm device "
                                       f"(group {group.get('name', '?')}: {p.dtype} on {p.device}, gradient {g.dtype} {g.layout}); "
                                       "there is no CPU path -- torch.optim.Adam is the class for anything else")
                todo.append((p, g, group))
        if not todo:
            return loss
        states = []
        for p, g, group in todo:
            st = self.state[p]
            if len(st) == 0:    # torch/optim/adam.py _init_group: a host-side step counter, zero moments
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["step"].is_cuda or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                raise RuntimeError("skelsplat_amd.optim.Adam: a loaded state with a device-side step counter or strided moments "
                                   "(saved by a capturable / fused optimiser?)")
            states.append(st)
        if any(p.numel() == 0 for p, _, _ in todo):   # (an empty tensor keeps its state and its step count like any other; nothing to launch for it)
            for (p, _, _), st in zip(todo, states):
                if p.numel() == 0:
                    st["step"] += 1
            keep = [i for i, t in enumerate(todo) if t[0].numel() > 0]
            todo, states = [todo[i] for i in keep], [states[i] for i in keep]
        lib = _lib.load()
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        switch = torch.cuda.current_device() != dev.index
        if switch:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(dev)
        try:
            for c0 in range(0, len(todo), _MAX):
                chunk, sts = todo[c0:c0 + _MAX], states[c0:c0 + _MAX]
                n = len(chunk)
                key = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                            for (p, g, _), st in zip(chunk, sts))
                plan = self._plans.get(c0)
                if plan is None or plan[0] != key:
                    arr = lambda vals: (C.c_void_p * n)(*vals)
                    plan = (key, arr(k[0] for k in key), arr(k[1] for k in key), arr(k[2] for k in key), arr(k[3] for k in key),
                            (C.c_longlong * n)(*(k[4] for k in key)), (C.c_double * n)(), (C.c_longlong * n)())
                    self._plans[c0] = plan
                _, ap, ag, am, av, an, alr, ast = plan
                for i, ((p, g, group), st) in enumerate(zip(chunk, sts)):
                    k = int(st["step"].item()) + 1
                    st["step"].fill_(float(k))
                    ast[i] = k
                    alr[i] = group["lr"]
                _lib.check(lib.sks_adam_multi(n, ap, ag, am, av, an, alr, ast, betas[0], betas[1], eps, stream), "sks_adam_multi")
        finally:
            if switch:
                torch.cuda.set_device(prev)
        return loss
