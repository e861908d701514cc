"""ctypes binding of libhsidm.so (the C ABI declared in include/hsidm.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If it cannot be loaded,
or a tensor is not on a ROCm device, the call raises.
"""
import ctypes as C
import os

import torch  # must be imported first: the library resolves libamdhip64.so.7 to the copy torch loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# HSIDM_LIB: diagnostic override (A/B builds, in-kernel stamp builds); the product library is libhsidm.so next to this file
LIB_PATH = os.environ.get("HSIDM_LIB") or os.path.join(_HERE, "libhsidm.so")

ABI_VERSION = 3          # include/hsidm.h: HSIDM_ABI_VERSION (2: the ConvDesc below has w_v2_ls / w_v2_li; 3: hsidm_conv1x1_pair)
BF16, F32X3, F16, F32H = 0, 1, 2, 3      # (F32H: hsidm_conv2d only - ops.PackedConv sets it; as a STORAGE type "fp32h" is F32X3)
XF_NONE, XF_AFFINE, XF_AFFINE_SILU = 0, 1, 2
ACT_NONE, ACT_LEAKY = 0, 1

_vp, _i32, _i64, _u32, _u64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float


class ConvPhase(C.Structure):
    _fields_ = [("src0", _vp), ("src1", _vp), ("gn_ab", _vp), ("C0", _i32), ("C1", _i32),
                ("transform", _i32), ("ntaps", _i32)]


class ConvDesc(C.Structure):
    _fields_ = [("ph", ConvPhase * 2), ("nphase", _i32), ("w_hi", _vp), ("w_lo", _vp), ("w_v2", _vp), ("bias", _vp),
                ("film", _vp), ("film_stride", _i32), ("res", _vp), ("res_scale", _f32), ("out", _vp),
                ("stats", _vp), ("B", _i32), ("Hin", _i32), ("Win", _i32), ("Hout", _i32), ("Wout", _i32),
                ("Cout", _i32), ("ksize", _i32), ("stride", _i32), ("ups", _i32), ("act", _i32),
                ("out_nchw", _i32), ("prec", _i32), ("bn", _i32), ("workspace", _vp), ("workspace_bytes", _i64), ("w_v2_lo", _vp),
                ("w_v2_ls", _vp), ("w_v2_li", _vp)]


class WgradItem(C.Structure):
    _fields_ = [("ws", _vp), ("dw", _vp), ("nsplit", _i32), ("NT", _i32), ("Cout_pad", _i32), ("Cin_pad", _i32), ("Cout_w", _i32),
                ("Cin_w", _i32), ("block0", _i32), ("layout", _i32)]


# name -> argtypes, exactly the prototypes of include/hsidm.h
SIGNATURES = {
    "hsidm_version": [],
    "hsidm_debug_switch": [C.c_char_p, _i32],
    "hsidm_debug_query": [C.c_char_p],
    "hsidm_conv_bk": [_i32],
    "hsidm_conv2d": [C.POINTER(ConvDesc), _vp],
    "hsidm_conv_stats_nsplit": [C.POINTER(ConvDesc)],
    "hsidm_conv_kernel_id": [C.POINTER(ConvDesc)],
    "hsidm_conv_workspace_bytes": [C.POINTER(ConvDesc)],
    "hsidm_conv1x1_pair": [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp],
    "hsidm_gn_partial": [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "hsidm_gn_finalize": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _vp],
    "hsidm_noise_film": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "hsidm_film_affine": [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "hsidm_attention": [_i32, _vp, _vp, _i32, _i32, _i32, _vp],
    "hsidm_nchw_to_nhwc": [_i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp],
    "hsidm_nhwc_to_nchw": [_i32, _vp, _vp, _i32, _i32, _i32, _vp],
    "hsidm_p_sample_update": [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _u64, _i64, _vp, _i32, _vp],
    "hsidm_step_advance": [_vp, _i32, _vp],
    "hsidm_philox_normal": [_vp, _i64, _u64, _u32, _vp],
    "hsidm_q_sample": [_vp, _vp, _vp, _vp, _i32, _i64, _vp],
    "hsidm_loss_workspace_bytes": [],
    "hsidm_loss_sum": [_vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "hsidm_ca_vector": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "hsidm_ca_apply": [_i32, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _i32, _vp],
    "hsidm_overlap_average": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "hsidm_hsi_metrics_workspace_bytes": [_i32, _i32, _i32],
    "hsidm_hsi_metrics": [_vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp],
    "hsidm_hsi_mssim_workspace_bytes": [_i32, _i32, _i32, _i32],
    "hsidm_hsi_mssim": [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "hsidm_resample_axis": [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp],
    "hsidm_minmax_workspace_bytes": [_i32],
    "hsidm_minmax_normalize": [_vp, _vp, _i32, _i64, _vp, _vp],
    "hsidm_augment": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "hsidm_color_correction_workspace_bytes": [_i32, _i32],
    "hsidm_color_correction": [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    # training step
    "hsidm_gn_act_apply": [_i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _f32, _u64, _vp, _u32, _vp, _vp],
    "hsidm_gn_act_bwd_workspace_floats": [_i32, _i32, _i32, _i32],
    "hsidm_gn_act_bwd": [_i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _u64, _vp, _u32, _i32, _vp, _vp, _vp,
                         _vp, _vp, _vp, _vp],
    "hsidm_conv_wgrad_workspace_bytes": [_i32] * 11,
    "hsidm_conv_wgrad": [_i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32,
                         _i32, _vp, _vp, _i64, _vp],
    "hsidm_conv_wgrad_plan": [_i32] * 11 + [_vp],
    "hsidm_wgrad_reduce_all": [_vp, _i32, _i32, _vp],
    "hsidm_add": [_i32, _vp, _vp, _vp, _i64, _vp],
    "hsidm_zero_insert2": [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "hsidm_sum2x2": [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "hsidm_colsum": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "hsidm_loss_grad": [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "hsidm_noise_film_bwd_workspace_floats": [_i32, _i32, _i32],
    "hsidm_noise_film_bwd": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hsidm_bgemm": [_vp, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp],
    "hsidm_softmax_rows": [_vp, _i64, _i32, _vp],
    "hsidm_softmax_bwd_rows": [_vp, _vp, _i64, _i32, _f32, _vp],
    "hsidm_gather_pack": [_vp, _vp, _i64, _vp, _vp, _vp],
    "hsidm_adam_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp],
}
RESTYPE_I64 = {"hsidm_conv_wgrad_workspace_bytes", "hsidm_conv_workspace_bytes"}

_lib = None


def lib():
    """Load libhsidm.so once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "hsidm: %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _i64 if name in RESTYPE_I64 else _i32
        L.hsidm_error_string.argtypes = [_i32]
        L.hsidm_error_string.restype = C.c_char_p
        got = L.hsidm_version()
        if got != ABI_VERSION:      # a library built against another header would read the structs above past their end
            raise RuntimeError("hsidm: %s implements ABI version %d, this binding is written against %d - rebuild it "
                               "(python -c 'import __graft_entry__ as g; g.build()')" % (LIB_PATH, got, ABI_VERSION))
        _lib = L
    return _lib


class debug_switch:
    """Context manager around hsidm_debug_switch (diagnostic A/B dispatch switches, include/hsidm.h):
    ``with debug_switch("NO_V3", 1): ...`` restores the previous value on exit."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.old = lib().hsidm_debug_switch(self.name, self.value)
        if self.old < 0:
            check(self.old, "debug_switch")
        return self

    def __exit__(self, *exc):
        lib().hsidm_debug_switch(self.name, self.old)
        return False


def check(code, what):
    if code != 0:
        raise RuntimeError("hsidm %s failed: %s (code %d)" % (what, lib().hsidm_error_string(code).decode(), code))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the product has no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hsidm: tensor is on %s; the HIP path needs a ROCm device tensor" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("hsidm: tensor must be contiguous")
    return t.data_ptr()


def prec_id(precision):
    if precision == "bf16":
        return BF16
    if precision in ("fp32", "f32x3", "fp32h"):
        return F32X3            # ("fp32h": fp32 tensors everywhere; only its convolutions differ - include/hsidm.h, HSIDM_F32H)
    if precision in ("fp16", "fp16x1", "fp16x2") or (isinstance(precision, str) and precision.startswith("fp16d") and precision[5:].isdigit()):
        return F16              # ("fp16d<k>": the dithered one-pass sets of the "fp16" policy; precision.dither_phase validates k)
    raise ValueError("precision must be 'bf16', 'fp16' or 'fp32', got %r" % (precision,))


def act_dtype(precision):
    return {BF16: torch.bfloat16, F16: torch.float16, F32X3: torch.float32}[prec_id(precision)]
