"""ctypes binding of libmrisr.so (include/mrisr.h).  There is NO fallback: if the HIP library is missing or a call
fails, an exception is raised - nothing here (or anywhere in this package) computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libmrisr.so")

MRISR_F32, MRISR_BF16, MRISR_F16, MRISR_I64 = 0, 1, 2, 3
MRISR_NCHW, MRISR_NHWC = 0, 1
STEP_DDIM, STEP_RESSHIFT, STEP_DDPM = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GEGLU = 0, 1, 2, 3

_DT = {torch.float32: MRISR_F32, torch.bfloat16: MRISR_BF16, torch.float16: MRISR_F16, torch.int64: MRISR_I64}


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("layout", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("attn_levels", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("num_heads", C.c_int32), ("cross_attention_dim", C.c_int32),
                ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float), ("cond_channels", C.c_int32),
                ("cond_embed_channels", C.c_int32 * 4), ("compute_dtype", C.c_int32), ("lora_rank", C.c_int32),
                ("lora_fused", C.c_int32), ("flash_attention", C.c_int32), ("fp8_linears", C.c_int32),
                ("fp8_attention", C.c_int32), ("fp8_train", C.c_int32)]


class AdapterCfg(C.Structure):
    _fields_ = [("channels", C.c_int32 * 4), ("nums_rb", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32),
                ("use_conv", C.c_int32), ("compute_dtype", C.c_int32)]


class VaeCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("latent_channels", C.c_int32),
                ("num_levels", C.c_int32), ("block_out_channels", C.c_int32 * 4), ("layers_per_block", C.c_int32),
                ("norm_num_groups", C.c_int32), ("compute_dtype", C.c_int32), ("scaling_factor", C.c_float)]


class MrisrError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

# every symbol include/mrisr.h declares (tests check the .so exports all of them)
EXPORTS = [
    "mrisr_last_error", "mrisr_version", "mrisr_unet_create", "mrisr_controlnet_create", "mrisr_model_destroy",
    "mrisr_model_set_param", "mrisr_model_set_lora_scale", "mrisr_model_finalize", "mrisr_model_num_params",
    "mrisr_model_workspace_bytes", "mrisr_unet_forward", "mrisr_model_set_context", "mrisr_model_num_skips",
    "mrisr_model_skip_shape", "mrisr_controlnet_forward", "mrisr_controlnet_set_cond", "mrisr_adapter_create",
    "mrisr_adapter_destroy", "mrisr_adapter_set_param", "mrisr_adapter_finalize", "mrisr_adapter_forward",
    "mrisr_resshift_forward", "mrisr_sampler_create", "mrisr_sampler_destroy", "mrisr_sampler_run", "mrisr_sampler_set_range", "mrisr_sampler_set_clip",
    "mrisr_adapter_train_prepare", "mrisr_adapter_train_num_trainable", "mrisr_adapter_train_num_tensors",
    "mrisr_adapter_train_tensor_info", "mrisr_adapter_train_bind", "mrisr_adapter_train_refresh", "mrisr_adapter_backward", "mrisr_adapter_backward_level", "mrisr_adapter_train_level_range",
    "mrisr_vae_create", "mrisr_vae_destroy", "mrisr_vae_set_param", "mrisr_vae_num_params", "mrisr_vae_finalize",
    "mrisr_vae_encode", "mrisr_vae_decode",
    "mrisr_image_metrics",
    "mrisr_resize_scratch_bytes", "mrisr_resize_slices", "mrisr_gaussian_blur_slices", "mrisr_low_field_scratch_bytes",
    "mrisr_simulate_low_field",
    "mrisr_train_prepare", "mrisr_train_num_trainable", "mrisr_train_num_tensors", "mrisr_train_tensor_info",
    "mrisr_train_bind", "mrisr_train_refresh", "mrisr_train_step", "mrisr_train_set_intrablock_grads", "mrisr_train_set_controlnet_residuals", "mrisr_controlnet_train_prepare", "mrisr_controlnet_train_num_trainable", "mrisr_controlnet_train_num_tensors", "mrisr_controlnet_train_tensor_info", "mrisr_controlnet_train_bind", "mrisr_controlnet_train_refresh", "mrisr_controlnet_train_forward", "mrisr_controlnet_train_backward", "mrisr_optim_sumsq", "mrisr_optim_adamw", "mrisr_optim_ema",
    "mrisr_prof_enable", "mrisr_prof_reset", "mrisr_prof_report",
    "mrisr_op_conv3x3", "mrisr_op_linear", "mrisr_op_ln_linear", "mrisr_op_linear_fp8", "mrisr_op_mlp", "mrisr_op_groupnorm", "mrisr_op_layernorm", "mrisr_op_attention",
    "mrisr_op_attention_bwd",
]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MrisrError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                             "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.mrisr_last_error.restype = C.c_char_p
        L.mrisr_version.restype = C.c_char_p
        L.mrisr_model_num_params.restype = C.c_int64
        L.mrisr_model_workspace_bytes.restype = C.c_int64
        L.mrisr_model_destroy.restype = None
        L.mrisr_adapter_destroy.restype = None
        L.mrisr_sampler_destroy.restype = None
        L.mrisr_sampler_set_clip.argtypes = [C.c_void_p, C.c_float]
        L.mrisr_resize_scratch_bytes.restype = C.c_size_t
        L.mrisr_resize_scratch_bytes.argtypes = [C.c_int] * 6
        L.mrisr_resize_slices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_size_t, C.c_void_p]
        L.mrisr_gaussian_blur_slices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
        L.mrisr_low_field_scratch_bytes.restype = C.c_size_t
        L.mrisr_low_field_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
        L.mrisr_simulate_low_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p]
        L.mrisr_vae_destroy.restype = None
        L.mrisr_vae_destroy.argtypes = [C.c_void_p]
        L.mrisr_vae_num_params.restype = C.c_int64
        L.mrisr_vae_num_params.argtypes = [C.c_void_p]
        for name in ("mrisr_model_destroy", "mrisr_adapter_destroy", "mrisr_sampler_destroy",
                     "mrisr_model_num_params", "mrisr_model_workspace_bytes", "mrisr_model_num_skips"):
            getattr(L, name).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise MrisrError(f"libmrisr error {rc}: {lib().mrisr_last_error().decode()}")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def as_tensor(t: Optional[torch.Tensor], layout: int = MRISR_NCHW, shape: Optional[Sequence[int]] = None) -> Tensor:
    """Describe a contiguous CUDA torch tensor.  ``shape`` overrides the logical {B,C,H,W} (NHWC buffers)."""
    d = Tensor()
    if t is None:
        return d
    if not t.is_cuda:
        raise MrisrError("tensors must live on the GPU")
    if not t.is_contiguous():
        raise MrisrError("tensors must be contiguous")
    d.data = t.data_ptr()
    d.dtype = _DT[t.dtype]
    d.layout = layout
    shp = list(shape) if shape is not None else list(t.shape)
    d.ndim = len(shp)
    for i, s in enumerate(shp):
        d.shape[i] = int(s)
    return d


def tensor_array(ts: Sequence[Tensor]):
    arr = (Tensor * max(1, len(ts)))()
    for i, t in enumerate(ts):
        arr[i] = t
    return arr


def dtype_id(name) -> int:
    if isinstance(name, torch.dtype):
        return _DT[name]
    return {"f32": MRISR_F32, "fp32": MRISR_F32, "float32": MRISR_F32, "bf16": MRISR_BF16,
            "bfloat16": MRISR_BF16}[str(name)]


def torch_dtype(dt: int) -> torch.dtype:
    return {MRISR_F32: torch.float32, MRISR_BF16: torch.bfloat16, MRISR_F16: torch.float16}[dt]


def push_param(set_fn, handle, key: str, value: torch.Tensor):
    v = value.detach().to(torch.float32).contiguous()
    shape = (C.c_int64 * max(1, v.ndim))(*[int(s) for s in v.shape])
    check(set_fn(handle, key.encode(), C.c_void_p(v.data_ptr()), shape, C.c_int(v.ndim), C.c_int(1 if v.is_cuda else 0)))
