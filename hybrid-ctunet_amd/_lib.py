"""ctypes binding of libctunet_hip.so (C ABI declared in include/ctunet_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libctunet_hip.so")
SOURCES = ["igemm.hip", "gemm_dma.hip", "gemm_narrow.hip", "conv3_halo.hip", "norm_elementwise.hip", "ff_fused.hip", "pwa_fused.hip", "attention.hip", "attention_mfma.hip",
           "loss_optim.hip", "infer.hip", "dropout.hip", "comm.hip", "plan.hip", "plan_dispatch.inc", "philox.h", "mma.h", "dma.h", "gemm_dma.h", "attn_common.h", "common.h"]

CTU_F32, CTU_BF16 = 0, 1
LAYOUT_NDHWC, LAYOUT_B16 = 0, 1
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Geom(C.Structure):
    """struct ctu_geom"""
    _fields_ = [(n, _i32) for n in ("B", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "C1", "C2", "N", "kd", "kh", "kw",
                                    "sd", "sh", "sw", "pd", "ph", "pw", "mode")]


class Epilogue(C.Structure):
    """struct ctu_epilogue"""
    _fields_ = [("bias", _vp), ("residual", _vp), ("act", _i32), ("ldc", _i32), ("out2", _vp), ("n_split", _i32),
                ("ldc2", _i32), ("scatter", _i32), ("n_per_tap", _i32), ("sc_D", _i32), ("sc_H", _i32),
                ("sc_W", _i32), ("sc_kd", _i32), ("sc_kh", _i32), ("sc_kw", _i32), ("splitk", _i32),
                ("w_kn", _i32), ("splitk_ws", _vp), ("in_acc", _vp), ("in_rows", _i32), ("reserved_", _i32), ("pre_out", _vp)]


class AttnGeom(C.Structure):
    """struct ctu_attn_geom"""
    _fields_ = [("part", _i32), ("B", _i32), ("D", _i32), ("H", _i32), ("W", _i32), ("win", _i32), ("heads", _i32),
                ("dh", _i32), ("scale", _f32)]


# name -> argtypes (all return int status except the two noted)
_SIGS = {
    "ctu_igemm_nt": [_i32, _vp, _vp, _vp, _vp, C.POINTER(Geom), C.POINTER(Epilogue), _vp],
    "ctu_igemm_tn": [_i32, _vp, _i32, _vp, _vp, _vp, _vp, C.POINTER(Geom), _vp, _i64, _vp],
    "ctu_conv3_halo": [_i32, _vp, _vp, _vp, _vp, _vp] + [_i32] * 10 + [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctu_in_finalize": [_i32, _i64, _i32, _vp, _vp, _vp],
    "ctu_conv3_halo_wgrad": [_i32, _vp, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _i64, _vp],
    "ctu_conv3_halo_wgrad_param": [_i32, _vp, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _i64, _vp],
    "ctu_in_apply_acc": [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "ctu_pack_frag": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp],
    "ctu_pack_frag_batched": [_vp, _i32, _i64, _vp],
    "ctu_im2col_cin1": [_vp, _vp, C.POINTER(Geom), _i32, _vp],
    "ctu_conv_cin1_fwd": [_i32, _vp, _vp, _vp, C.POINTER(Geom), _vp],
    "ctu_conv_cin1_wgrad": [_i32, _vp, _vp, _vp, C.POINTER(Geom), _vp],
    "ctu_permute3": [_vp, _vp, _i32] + [_i64] * 9 + [_i32, _vp],
    "ctu_colsum": [_i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "ctu_outer_rows": [_i32, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctu_in_stats": [_i32, _vp, _i32, _i64, _i32, _vp, _vp, _vp],
    "ctu_in_apply": [_i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _vp],
    "ctu_in_bwd_reduce": [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp],
    "ctu_in_bwd_apply": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "ctu_in_apply_dual": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp],
    "ctu_in_bwd_fused": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp],
    "ctu_layernorm_fwd": [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctu_layernorm_bwd": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctu_layernorm_bwd_add": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctu_gelu_fwd": [_i32, _vp, _vp, _i64, _vp],
    "ctu_gelu_bwd": [_i32, _vp, _vp, _vp, _i64, _vp],
    "ctu_add": [_i32, _vp, _vp, _vp, _i64, _vp],
    "ctu_add_bcast": [_i32, _vp, _vp, _vp, _i64, _i32, _i64, _vp],
    "ctu_attn_fwd": [_i32, _vp, _vp, _vp, _vp, C.POINTER(AttnGeom), _vp],
    "ctu_attn_bwd": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(AttnGeom), _vp],
    "ctu_attn_fwd_dropout": [_i32, _vp, _vp, _vp, _vp, C.POINTER(AttnGeom), _f32, C.c_uint64, C.c_uint64, _vp],
    "ctu_attn_bwd_dropout": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(AttnGeom), _f32, C.c_uint64, C.c_uint64, _vp],
    "ctu_dropout": [_i32, _vp, _vp, _vp, _i64, _f32, C.c_uint64, C.c_uint64, _vp],
    "ctu_attn_dropout_mask": [_vp, _i32, _i32, _f32, C.c_uint64, C.c_uint64, _vp],
    "ctu_ff_pack_w2": [_vp, _vp, _i32, _i32, _vp],
    "ctu_ff_fwd": [_i32] + [_vp] * 11 + [_i64, _i32, _i32, _vp],
    "ctu_pwa_pack": [_vp, _vp, _vp, _vp, _i32, _vp],
    "ctu_pwa_block_fwd": [_i32] + [_vp] * 12 + [_i64, _i32, _f32, _vp],
    "ctu_pwa_fwd": [_i32, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "ctu_pwa_bwd": [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "ctu_patchify": [_i32, _vp, _vp] + [_i32] * 7 + [_vp],
    "ctu_pixel_shuffle": [_i32, _vp, _vp] + [_i32] * 9 + [_vp],
    "ctu_upsample2_zeros": [_i32, _vp, _vp] + [_i32] * 5 + [_vp],
    "ctu_add_strided2": [_i32, _vp, _vp] + [_i32] * 5 + [_vp],
    "ctu_dicece_fwd": [_i32, _vp, _i32, _vp, _vp, _vp, _vp] + [_i32] * 8 + [_vp, _vp],
    "ctu_dicece_finalize": [_vp, _i32, _i32, _i64, _f32, _f32, _f32, _vp, _vp],
    "ctu_dicece_bwd": [_i32, _vp, _i32, _vp, _vp, _vp, _vp] + [_i32] * 8 + [_vp, _f32, _f32, _f32, _vp, _vp, _vp],
    "ctu_adamw": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, C.POINTER(_i64), _i32, _vp, _vp],
    "ctu_adamw_tick": [_vp, _f32, _f32, _vp],
    "ctu_cast": [_vp, _i32, _vp, _i32, _i64, _vp],
    "ctu_sw_accumulate": [_i32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp] + [_i32] * 11 + [_vp],
    "ctu_sw_normalize": [_vp, _vp, _i32, _i32, _i64, _vp],
    "ctu_hybrid_argmax": [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "ctu_set_option": [C.c_char_p, _i32],
    "ctu_comm_unique_id": [C.c_char_p, _vp],
    "ctu_comm_init": [C.c_char_p, _i32, _i32, _vp, C.POINTER(_vp)],
    "ctu_comm_destroy": [_vp],
    "ctu_allreduce_bucket": [_vp, _vp, _i64, _i32, _vp, _i64, _vp],
    "ctu_allreduce_bucket_stage": [_i32, _i32, _vp, _i64, _vp, _i64, _vp],
    "ctu_plan_create": [C.POINTER(C.c_uint64), _i64, C.POINTER(C.c_uint64), _i64, _i32, _i32, C.POINTER(_vp)],
    "ctu_plan_run": [_vp, C.POINTER(C.c_uint64), _i32, C.POINTER(_vp), _i32],
    "ctu_plan_destroy": [_vp],
}
EXPORTED = sorted(list(_SIGS) + ["ctu_abi_version", "ctu_last_error", "ctu_allreduce_scratch_bytes", "ctu_sync_timeouts"])

_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into csrc/libctunet_hip.so for gfx950 (hipcc cross-compiles without a GPU).  `make` decides what is
    stale (every object depends on all headers, the generated plan_dispatch.inc and include/ctunet_hip.h), so a library older
    than any source is rebuilt; without hipcc (the GPU box never builds) an up-to-date library is accepted and a stale one
    refused."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], capture_output=True, text=True)
    fresh = subprocess.run(["make", "-C", CSRC, "-q", f"HIPCC={hipcc}"], capture_output=True, text=True).returncode == 0
    if fresh and os.path.exists(LIB_PATH):
        return LIB_PATH
    cmd = ["make", "-C", CSRC, "-j4", f"HIPCC={hipcc}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:], res.stderr[-4000:])
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError(f"building libctunet_hip.so failed (exit {res.returncode}): the in-tree library is missing or older "
                           "than its sources")
    return LIB_PATH


def lib():
    """Load the shared library (never builds implicitly on a GPU box: the in-tree .so travels with the repo)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "hybrid-ctunet_amd has no CPU/eager fallback by design.")
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.ctu_abi_version.restype = C.c_int
        L.ctu_allreduce_scratch_bytes.argtypes = [_i32, _i64]
        L.ctu_allreduce_scratch_bytes.restype = _i64
        L.ctu_last_error.restype = C.c_char_p
        L.ctu_sync_timeouts.restype = C.c_int
        if L.ctu_abi_version() != 8:
            raise RuntimeError("libctunet_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib().ctu_last_error().decode()}")


PROFILER = None  # set by bench.py to an object with .names (set of entry points) and .add(name, args, ev0, ev1)


_FN = {}  # entry point name -> bound ctypes function (a step makes ~2 000 calls: keep the per-call Python short)


def _fn(name: str):
    f = _FN.get(name)
    if f is None:
        f = _FN[name] = getattr(lib(), name)
    return f


def call(name: str, *args):
    prof = PROFILER
    if prof is not None and name in prof.names:
        # HIP events on the stream the kernel is launched on (torch's current stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_fn(name)(*args), name)
        e1.record()
        prof.add(name, args, e0, e1)
    else:
        rc = _fn(name)(*args)
        if rc != 0:
            check(rc, name)


def ptr(t):
    return None if t is None else t.data_ptr()  # (ctypes converts the int for a c_void_p parameter)


_DEV = None


def stream():
    """Raw hipStream_t of torch's current stream on this process's device (one process per GPU)."""
    global _DEV
    if _DEV is None:
        _DEV = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV)


def dcode(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return CTU_F32
    if dtype == torch.bfloat16:
        return CTU_BF16
    raise TypeError(f"hybrid-ctunet_amd kernels support float32 and bfloat16 activations, got {dtype}")


def require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("hybrid-ctunet_amd runs on an MI355X (HIP) device only; got a CPU tensor. "
                           "There is no CPU fallback (use oracle/ for CPU reference numbers).")
