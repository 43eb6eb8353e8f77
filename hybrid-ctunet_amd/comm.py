"""RCCL communicator behind the C ABI (include/ctunet_hip.h: ctu_comm_init / ctu_allreduce_bucket): the gradient-bucket
exchange of train.DataParallel with a bf16 payload (direct reduce-scatter as an all-to-all + all-gather, fp32
accumulation).  One communicator per process = per GPU; the 128-byte RCCL id travels from rank 0 through whatever
torch.distributed group is already up (main_CTUNet.py:116-118 creates one)."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib


def rccl_library_path() -> str:
    """The RCCL shared object this process already uses: PyTorch's own copy (so that one RCCL runtime serves
    torch.distributed and the C ABI), else ROCm's."""
    own = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(own):
        return own
    for cand in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(cand):
            return cand
    raise RuntimeError("no librccl.so found (looked in torch/lib and /opt/rocm/lib)")


class Communicator:
    def __init__(self, rank: int, world: int, unique_id: bytes, device=None, path: str | None = None):
        if device is not None:
            torch.cuda.set_device(device)
        self.rank, self.world = int(rank), int(world)
        self.path = (path or rccl_library_path()).encode()
        handle = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, 128)
        _lib.call("ctu_comm_init", self.path, self.rank, self.world, C.cast(idbuf, C.c_void_p), C.byref(handle))
        self._handle = handle
        self._scratch = None

    @staticmethod
    def unique_id(path: str | None = None) -> bytes:
        buf = C.create_string_buffer(128)
        _lib.call("ctu_comm_unique_id", (path or rccl_library_path()).encode(), C.cast(buf, C.c_void_p))
        return buf.raw

    @classmethod
    def from_torch(cls, group=None, device=None) -> "Communicator":
        """Create the communicator for this process's rank of an initialised torch.distributed group; the id is made on
        rank 0 and broadcast through that group (any backend)."""
        if not dist.is_initialized():
            return cls(0, 1, cls.unique_id(), device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0], device)

    def _scratch_for(self, n: int, device) -> torch.Tensor:
        need = _lib.lib().ctu_allreduce_scratch_bytes(self.world, n)
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != device:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=device)
        return self._scratch

    def allreduce_mean(self, buf: torch.Tensor, payload: str = "fp32"):
        """buf (contiguous fp32, device) <- mean over ranks, on torch's current stream."""
        if buf.dtype != torch.float32 or not buf.is_contiguous() or not buf.is_cuda:
            raise TypeError("allreduce_mean expects a contiguous float32 device tensor")
        if payload == "bf16":
            sc = self._scratch_for(buf.numel(), buf.device)
            _lib.call("ctu_allreduce_bucket", self._handle, buf.data_ptr(), buf.numel(), _lib.CTU_BF16, sc.data_ptr(),
                      sc.numel(), _lib.stream())
        elif payload == "fp32":
            _lib.call("ctu_allreduce_bucket", self._handle, buf.data_ptr(), buf.numel(), _lib.CTU_F32, None, 0, _lib.stream())
        else:
            raise ValueError(payload)

    def allreduce_mean_bf16(self, buf: torch.Tensor):
        self.allreduce_mean(buf, "bf16")

    def close(self):
        if self._handle:
            _lib.call("ctu_comm_destroy", self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
