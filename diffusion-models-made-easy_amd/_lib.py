"""ctypes binding of libdmme_hip.so (the C ABI declared in include/dmme_hip.h).

The product path has no CPU fallback: `lib()` raises if the shared library is missing,
and `require_gpu()` raises if it loaded but sees no HIP device."""

from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMME_LIB_PATH") or os.path.join(_HERE, "libdmme_hip.so")  # (override: A/B runs of two builds on one box)
CSRC = os.path.join(_HERE, "csrc")

F16R32 = 4  # F16 below the full-resolution level; that level in fp32 tensors with three-pass split-fp16 products (within 1e-3 of fp32; inference)
F32, BF16, BF16X3, F16 = 0, 1, 2, 3  # BF16X3: fp32 buffers, three-pass bf16 MFMA convolutions (the accurate mode); F16: IEEE half (inference)
CHAIN_DDPM, CHAIN_DDIM, CHAIN_IDDPM = 0, 1, 2
DTYPES = {"fp32": F32, "float32": F32, "32": F32, "bf16": BF16, "bfloat16": BF16, "16": BF16, "bf16-mixed": BF16, "16-mixed": BF16,
          "bf16x3": BF16X3, "fp16": F16, "float16": F16, "half": F16, "fp16r32": F16R32}

_lock = threading.Lock()
_lib = None


class UNetCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int),
        ("pos_dim", C.c_int),
        ("emb_dim", C.c_int),
        ("num_groups", C.c_int),
        ("dropout", C.c_float),
        ("num_depths", C.c_int),
        ("channels_per_depth", C.c_int * 8),
        ("num_blocks", C.c_int),
        ("num_attention_depths", C.c_int),
        ("attention_depths", C.c_int * 8),
        ("arch", C.c_int),
        ("num_heads", C.c_int),
    ]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "N", "Hin", "Win", "C1", "C2", "upsample", "stride", "taps", "Cout", "pro_silu", "out_silu",
        "nt", "tproj_ld", "in_nchw", "out_nchw", "force_generic")]  # 0: best kernel, 1: generic kernel, 2: first-generation MFMA kernel


BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int64, C.c_int64)  # dmme_bucket_fn


class DmmeError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libdmme_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise DmmeError("building libdmme_hip.so failed")
    return LIB_PATH


_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); must list every symbol include/dmme_hip.h declares
PROTOTYPES = {
    "dmme_last_error": (C.c_char_p, []),
    "dmme_version": (_i, []),
    "dmme_device_count": (_i, []),
    "dmme_unet_plan_create": (_i, [C.POINTER(UNetCfg), _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "dmme_unet_plan_destroy": (None, [_vp]),
    "dmme_unet_plan_num_params": (_i, [_vp]),
    "dmme_unet_plan_param_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)]),
    "dmme_unet_plan_ref_numel": (_i64, [_vp]),
    "dmme_unet_plan_packed_bytes": (_i64, [_vp]),
    "dmme_unet_plan_workspace_bytes": (_i64, [_vp]),
    "dmme_unet_plan_dropmask_numel": (_i64, [_vp]),
    "dmme_unet_plan_out_channels": (_i, [_vp]),
    "dmme_unet_plan_num_launches": (_i, [_vp]),
    "dmme_unet_pack_params": (_i, [_vp, _vp, _vp, _vp]),
    "dmme_unet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dmme_unet_forward_nograd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dmme_unet_plan_num_ops": (_i, [_vp]),
    "dmme_unet_plan_level_info": (_i, [_vp, C.c_char_p, _i]),
    "dmme_unet_plan_check": (_i, [_vp]),
    "dmme_debug_level_stamps": (_i, [_vp, _i, _i]),
    "dmme_unet_plan_op_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dmme_unet_forward_profiled": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmme_unet_plan_packed_bwd_bytes": (_i64, [_vp]),
    "dmme_unet_plan_bwd_workspace_bytes": (_i64, [_vp]),
    "dmme_unet_pack_params_bwd": (_i, [_vp, _vp, _vp, _vp]),
    "dmme_unet_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmme_unet_backward_buckets": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, BUCKET_FN, _vp]),
    "dmme_unet_plan_grad_buckets": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i), _i]),
    "dmme_unet_plan_bwd_summary": (_i, [_vp, C.c_char_p, _i]),
    "dmme_grad_norm": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "dmme_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _vp, _f, _f, _f, _vp]),
    "dmme_amp_init": (_i, [_vp, _f, _vp]),
    "dmme_amp_scale": (_i, [_vp, _i64, _vp, _vp]),
    "dmme_adam_step_amp": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp, _f, _f, _f, _vp, _f, _f, _i, _vp]),
    "dmme_grad_pack_bf16": (_i, [_vp, _i64, _vp, _i64, _vp]),
    "dmme_shard_reduce_bf16": (_i, [_vp, _i, _i64, _f, _vp, _vp]),
    "dmme_grad_unpack_bf16": (_i, [_vp, _i64, _vp, _vp]),
    "dmme_unet_debug_read": (_i, [_vp, _vp, C.c_char_p, _vp, _i64, C.POINTER(_i64), _vp]),
    "dmme_dropout_masks": (_i, [_vp, _u64, _u64, _vp, _vp]),
    "dmme_randn": (_i, [_vp, _i64, _u64, _u64, _vp]),
    "dmme_q_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "dmme_ddpm_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _i, _i64, _vp]),
    "dmme_ddim_step": (_i, [_vp, _vp, _f, _f, _i64, _vp]),
    "dmme_chain_set": (_i, [_vp, _i64, _vp, _u64, _u64, _vp]),
    "dmme_chain_update": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "dmme_chain_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dmme_mse_loss": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _vp, _vp]),
    "dmme_image_batch": (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "dmme_iddpm_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _i, _i, _i64, _vp]),
    "dmme_iddpm_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp, _f, _vp, _vp]),
    "dmme_conv2d": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dmme_conv2d_res": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dmme_groupnorm_scale_shift": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp]),
    "dmme_attention": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "dmme_attention_proj": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dmme_attention_heads": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "dmme_nchw_to_nhwc": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "dmme_nhwc_to_nchw": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "dmme_pack_weight": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "dmme_event_create": (_i, [C.POINTER(_vp)]),
    "dmme_event_record": (_i, [_vp, _vp]),
    "dmme_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "dmme_event_destroy": (_i, [_vp]),
    "dmme_debug_set_stamps": (_i, [_vp]),
    "dmme_debug_mfma_valu": (_i, [_i, _i, _i, _vp, _vp]),
    "dmme_debug_issue_probe": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "dmme_debug_l2_stream": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
}


def lib():
    """Load (once) and return the ctypes handle; raises loudly if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DmmeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C diffusion-models-made-easy_amd/csrc`). There is no CPU fallback for the product path."
            )
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dmme_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise DmmeError(f"{what} failed (status {rc}): {msg}")


def require_gpu():
    if lib().dmme_device_count() <= 0:
        raise DmmeError("libdmme_hip.so loaded but no HIP device is visible; the denoiser path only runs on an MI355X")


def dtype_code(precision) -> int:
    key = str(precision).lower()
    if key not in DTYPES:
        raise ValueError(f"unknown precision {precision!r}; use 'fp32', 'bf16', 'fp16' / 'fp16r32' (inference) or 'bf16x3'")
    return DTYPES[key]


def csrc_sha16() -> str:
    """first 16 hex digits of SHA-256 over the library's sources (csrc/*.hip, *.h, the Makefile and include/dmme_hip.h, sorted by name):
    ties committed profiler figures (profiles/*_latest.json) to the build they were measured on - bench.py drops them when it differs"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "Makefile")])
    files.append(os.path.join(os.path.dirname(_HERE), "include", "dmme_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device/host pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
