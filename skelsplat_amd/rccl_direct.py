"""RCCL's C API called directly on the CALLER's stream (ctypes over the librccl.so torch itself loads).

torch.distributed's NCCL process group runs every collective on its own internal stream and hands over with events on both
sides; for the loop's one tiny all_gather per step (a few KB, latency-bound) that hand-over is most of its cost on the GPU
timeline.  A communicator of our own -- ncclCommInitRank with an id rank 0 creates and torch.distributed broadcasts -- lets
`ncclAllGather` be enqueued on the stream the kernels around it run on.  OPT-IN: `DirectGather.create` returns None unless the
environment says SKS_RCCL_DIRECT=1 (and whenever anything is missing); torch.distributed stays the bootstrap and the default
exchange.  Status: exercised on a communicator of ONE rank only (no multi-GPU hardware was available to this build); `create`
therefore runs one all_gather through both paths and keeps the communicator only if every rank saw identical results."""
import ctypes
import os
import weakref

import torch
import torch.distributed as dist

NCCL_FLOAT = 7      # ncclFloat32 (nccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7)


_COMMS = {}     # (group id or 0, device index) -> (DirectGather, weak reference to the group or None, world size it was built for)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]   # (c_char arrays read back truncated at the first NUL)


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)
    lib.ncclGetUniqueId.restype = ctypes.c_int
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = ctypes.c_int
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllGather.restype = ctypes.c_int
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p]
    lib.ncclRedOpCreatePreMulSum.restype = ctypes.c_int
    lib.ncclRedOpCreatePreMulSum.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ncclRedOpDestroy.restype = ctypes.c_int
    lib.ncclRedOpDestroy.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.ncclCommDestroy.restype = ctypes.c_int
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    return lib


class DirectGather:
    """all_gather_into_tensor(out, inp) on the current stream of `device`, over a communicator of this object's own."""

    def __init__(self, lib, comm, world, device):
        self.lib, self.comm, self.world, self.device = lib, comm, world, device
        self._ops = {}     # weight -> ncclRedOp_t of all_reduce_weighted

    @classmethod
    def create(cls, device, group=None):
        """Collective over `group` (every rank must call it).  Returns None -- on EVERY rank -- unless every rank got its
        communicator (the ranks agree through one all_reduce of torch.distributed), the backend is RCCL and
        SKS_RCCL_DIRECT is not 0."""
        if os.environ.get("SKS_RCCL_DIRECT", "0") != "1" or not dist.is_initialized() or dist.get_backend(group) != "nccl":
            return None
        if device.type != "cuda" or device.index is None:
            return None
        key = (id(group) if group is not None else 0, device.index)
        hit = _COMMS.get(key)
        if hit is not None:     # one communicator per (group, device): every loop object of a process shares it
            dg, gref, world0 = hit
            # (a process group that was destroyed and built again -- same key, another membership -- must not find this one)
            if (gref is None or gref() is group) and world0 == dist.get_world_size(group) and dg.comm:
                return dg
            dg.destroy()

        def agree(flag):        # True only if `flag` holds on every rank
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        def note(what, e):
            if os.environ.get("SKS_RCCL_DIRECT_DEBUG"):
                print(f"DirectGather.create: {what}: {e!r}")

        # every step that can fail on one rank alone is followed by an agreement, so that no rank walks into a collective
        # (the broadcast, ncclCommInitRank) its peers have already given up on
        lib = None
        try:
            lib = _load()
        except Exception as e:
            note("librccl", e)
        if not agree(lib is not None):
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        uid = _UniqueId()
        blob = [None]
        if rank == 0:
            try:
                rc = lib.ncclGetUniqueId(ctypes.byref(uid))
                if rc:
                    raise RuntimeError(lib.ncclGetErrorString(rc).decode())
                blob = [ctypes.string_at(ctypes.byref(uid), 128)]
            except Exception as e:
                note("ncclGetUniqueId", e)
        dist.broadcast_object_list(blob, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                                   device=device)
        if blob[0] is None:
            return None
        ctypes.memmove(ctypes.byref(uid), blob[0], 128)
        comm = ctypes.c_void_p()
        me = None
        try:
            with torch.cuda.device(device):
                rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
            if rc:
                raise RuntimeError(lib.ncclGetErrorString(rc).decode())
            me = cls(lib, comm, world, device)
            me._key = key
        except Exception as e:
            note("ncclCommInitRank", e)
        if not agree(me is not None):
            if me is not None:
                me.destroy()
            return None
        # one all_gather through both paths: the communicator is kept only if every rank saw the same rows
        same = False
        try:
            probe = torch.arange(8, dtype=torch.float32, device=device) + 100.0 * rank
            a, b = torch.empty(8 * world, device=device), torch.empty(8 * world, device=device)
            dist.all_gather_into_tensor(a, probe, group=group)
            me.all_gather_into_tensor(b, probe)
            torch.cuda.synchronize(device)
            same = bool(torch.equal(a, b))
        except Exception as e:
            note("self-check", e)
        if not agree(same):
            me.destroy()
            return None
        gref = None
        if group is not None:
            try:
                gref = weakref.ref(group)
            except TypeError:
                gref = None
        _COMMS[key] = (me, gref, world)
        return me

    def all_gather_into_tensor(self, out, inp):
        if out.device != self.device or inp.device != self.device:
            raise ValueError(f"DirectGather: tensors must live on {self.device} (the communicator's device)")
        if out.numel() != self.world * inp.numel() or not out.is_contiguous() or not inp.is_contiguous() \
                or out.dtype != torch.float32 or inp.dtype != torch.float32:
            raise ValueError("DirectGather: contiguous fp32 tensors, out = world x inp")
        rc = self.lib.ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), NCCL_FLOAT, self.comm,
                                    torch._C._cuda_getCurrentRawStream(self.device.index))
        if rc:
            raise RuntimeError("ncclAllGather: " + self.lib.ncclGetErrorString(rc).decode())

    def all_reduce_weighted(self, out, inp, weight):
        """out = sum over the ranks of weight_r * inp_r (every rank its own weight), one collective on the current stream:
        ncclAllReduce with a pre-multiplied sum (ncclRedOpCreatePreMulSum, the scalar a host immediate).  With inp_r = the
        mean of a rank's V_r local per-view gradients and weight_r = V_r / V this is the mean over all V views
        (train.py:215-217) without gathering the per-view rows."""
        if out.device != self.device or inp.device != self.device:
            raise ValueError(f"DirectGather: tensors must live on {self.device} (the communicator's device)")
        if out.shape != inp.shape or not out.is_contiguous() or not inp.is_contiguous() or out.dtype != torch.float32 \
                or inp.dtype != torch.float32:
            raise ValueError("DirectGather.all_reduce_weighted: contiguous fp32 tensors of one shape")
        w = float(weight)
        op = self._ops.get(w)
        if op is None:
            opv, scalar = ctypes.c_int(), ctypes.c_float(w)
            rc = self.lib.ncclRedOpCreatePreMulSum(ctypes.byref(opv), ctypes.byref(scalar), NCCL_FLOAT, 1, self.comm)   # 1 = ncclScalarHostImmediate
            if rc:
                raise RuntimeError("ncclRedOpCreatePreMulSum: " + self.lib.ncclGetErrorString(rc).decode())
            op = self._ops[w] = opv.value
        rc = self.lib.ncclAllReduce(inp.data_ptr(), out.data_ptr(), inp.numel(), NCCL_FLOAT, op, self.comm,
                                    torch._C._cuda_getCurrentRawStream(self.device.index))
        if rc:
            raise RuntimeError("ncclAllReduce: " + self.lib.ncclGetErrorString(rc).decode())

    def destroy(self):
        _COMMS.pop(getattr(self, "_key", None), None)
        for op in self._ops.values():
            self.lib.ncclRedOpDestroy(op, self.comm)
        self._ops = {}
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def destroy_all():
    """ncclCommDestroy of every communicator this module built (call before dist.destroy_process_group())."""
    for dg, _, _ in list(_COMMS.values()):
        dg.destroy()
