"""ctypes binding of libcenternet_gfx950.so (C ABI: include/centernet_gfx950.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p)

_LIB_NAME = "libcenternet_gfx950.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", _LIB_NAME)

CNL_RELU = 1 << 0
CNL_SIGMOID = 1 << 1
CNL_UPSAMPLE_IN = 1 << 2
CNL_UPSAMPLE_OUT_ADD = 1 << 3
CNL_RELU6 = 1 << 4
CNL_W_SPLIT = 1 << 5          # cnl_conv_params.w is a cnl_conv_split_weights_f32 buffer (fp32 weights + their fp16 split)

# cnl_conv_params.algo: the arithmetic class a launch may use (include/centernet_gfx950.h)
CNL_ALGO_AUTO, CNL_ALGO_F2, CNL_ALGO_F32, CNL_ALGO_LATENCY, CNL_ALGO_F43, CNL_ALGO_FORCE = 0, 1, 2, 4, 5, 100
CNL_WINO_F32, CNL_WINO_F16X2 = 2, 5

CNL_E_BAD_ARG, CNL_E_UNSUPPORTED, CNL_E_WORKSPACE, CNL_E_HIP = -1, -2, -3, -4
ABI_VERSION = 13         # CNL_ABI_VERSION of include/centernet_gfx950.h


class ConvParams(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("y", c_void_p),
                ("N", c_int32), ("H_in", c_int32), ("W_in", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
                ("ldx", c_int32), ("ldy", c_int32), ("ldr", c_int32), ("flags", c_uint32),
                ("x_absmax", c_void_p), ("y_absmax", c_void_p), ("w_absmax", c_void_p), ("algo", c_uint32),
                ("splitk", c_int32), ("splitk_scratch", c_void_p), ("splitk_scratch_bytes", c_size_t),
                ("fuse_w", c_void_p), ("fuse_part", c_void_p), ("w_up", c_void_p)]


class DeconvParams(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("y", c_void_p),
                ("N", c_int32), ("H_in", c_int32), ("W_in", c_int32), ("Cin", c_int32), ("Cout", c_int32), ("K", c_int32),
                ("ldx", c_int32), ("ldy", c_int32), ("ldr", c_int32), ("flags", c_uint32)]


class DecodeParams(Structure):
    _fields_ = [("heat", c_void_p), ("heat_sn", c_int64), ("heat_sc", c_int64), ("heat_sh", c_int64), ("heat_sw", c_int64),
                ("box", c_void_p), ("box_sn", c_int64), ("box_sc", c_int64), ("box_sh", c_int64), ("box_sw", c_int64),
                ("reid", c_void_p), ("reid_sn", c_int64), ("reid_sc", c_int64), ("reid_sh", c_int64), ("reid_sw", c_int64),
                ("N", c_int32), ("C", c_int32), ("H", c_int32), ("W", c_int32), ("E", c_int32),
                ("k", c_int32), ("nms_kernel", c_int32), ("normalize_boxes", c_int32), ("box_log", c_int32),
                ("box_multiplier", c_float), ("stride", c_float),
                ("scores", c_void_p), ("indices", c_void_p), ("labels", c_void_p), ("boxes", c_void_p), ("emb", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


_SIGNATURES = {
    "cnl_version": (ctypes.c_int, []),
    "cnl_sizeof_params": (c_size_t, [ctypes.c_int32]),
    "cnl_absmax_stride": (ctypes.c_int, []),
    "cnl_last_error": (c_size_t, [c_char_p, c_size_t]),
    "cnl_conv2d_nhwc_f32": (ctypes.c_int, [POINTER(ConvParams), c_void_p]),
    "cnl_fused_out_pack_weights_f32": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    "cnl_fused_out_reduce_f32": (ctypes.c_int, [c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, c_void_p, c_void_p, ctypes.c_int32, ctypes.c_uint32, c_void_p]),
    "cnl_conv2d_out_hw": (ctypes.c_int, [POINTER(ConvParams), POINTER(c_int32), POINTER(c_int32)]),
    "cnl_conv3x3_winograd_f32": (ctypes.c_int, [POINTER(ConvParams), c_void_p]),
    "cnl_conv3x3_winograd_kernel": (ctypes.c_int, [POINTER(ConvParams)]),
    "cnl_winograd_up_weight_floats": (c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "cnl_winograd_transform_weights_up_f32": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    "cnl_conv3x3_winograd_variant": (ctypes.c_int, [POINTER(ConvParams)]),
    "cnl_conv2d_kernel": (ctypes.c_int, [POINTER(ConvParams)]),
    "cnl_up2_weight_floats": (c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "cnl_up2_pack_weights_f32": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    "cnl_conv3x3_up2_nhwc_f32": (ctypes.c_int, [POINTER(ConvParams), c_void_p]),
    "cnl_conv3x3_up2_kernel": (ctypes.c_int, [POINTER(ConvParams)]),
    "cnl_absmax_per_image_f32": (ctypes.c_int, [c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_void_p, c_void_p]),
    "cnl_winograd_weight_floats": (c_size_t, [c_int32, c_int32]),
    "cnl_winograd_transform_weights_f32": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "cnl_deconv2x_nhwc_f32": (ctypes.c_int, [POINTER(DeconvParams), c_void_p]),
    "cnl_deconv_phase_geometry": (ctypes.c_int, [c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "cnl_deconv_weight_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "cnl_conv2d_splitk_scratch_bytes": (c_size_t, [POINTER(ConvParams)]),
    "cnl_fuse_sum_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                             c_int32, c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_int32, c_void_p]),
    "cnl_upsample2x_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_int32, c_int32, c_void_p]),
    "cnl_depthwise3x3_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                 c_int32, c_uint32, c_void_p]),
    "cnl_deform_sample_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                  c_int32, c_int32, c_void_p]),
    "cnl_normalize_u8_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(c_float), c_void_p]),
    "cnl_resize_bilinear_u8": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_stem_conv7x7_u8": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, POINTER(c_float), POINTER(c_float), c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_stem_packed_weight_floats": (c_size_t, []),
    "cnl_stem_pack_weights_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    "cnl_stem_conv7x7_f32": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int32, c_int32, c_int32, c_uint32, c_void_p]),
    "cnl_stem_conv7x7_maxpool_f32": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int32, c_int32, c_int32, c_void_p]),
    "cnl_maxpool3x3s2_nhwc_f32": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_decode_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "cnl_decode_f32": (ctypes.c_int, [POINTER(DecodeParams), c_void_p]),
    "cnl_gather_boxes_f32": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                            c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "cnl_gather_embeddings_f32": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                                 c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_pack_detections_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_unpack_detections_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_boxes_xyxy_to_xywh_f32": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "cnl_track_costs_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int32,
                                           c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cnl_track_costs_metric_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int32,
                                                  c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cnl_track_apply_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                           c_void_p, c_void_p, c_void_p]),
    "cnl_conv_split_weight_floats": (ctypes.c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "cnl_conv_split_weights_f32": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "cnl_track_frame_bytes": (ctypes.c_int64, [c_int32, c_int32, c_int32]),
    "cnl_track_frame_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                           c_int32, c_int32, c_int32, c_int32, c_void_p, ctypes.c_int64, c_void_p]),
    "cnl_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(c_void_p)]),
    "cnl_host_free": (ctypes.c_int, [c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("CENTERNET_GFX950_LIB", _LIB_PATH)


def load():
    """Load (once) and return the ctypes handle.  Raises HipLibraryError when the .so is missing —
    build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipLibraryError(f"{_LIB_NAME} not found at {path}; the HIP extension is required (no CPU fallback). "
                              "Build it with __graft_entry__.build().")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.cnl_version() != ABI_VERSION:
        raise HipLibraryError(f"{path} has ABI version {lib.cnl_version()}, this binding expects {ABI_VERSION}: rebuild it "
                              "(__graft_entry__.build())")
    for which, struct in enumerate((ConvParams, DecodeParams, DeconvParams)):       # the binding's struct layouts against the library's
        if lib.cnl_sizeof_params(which) != ctypes.sizeof(struct):
            raise HipLibraryError(f"{path}: sizeof({struct.__name__}) is {lib.cnl_sizeof_params(which)} in the library, {ctypes.sizeof(struct)} in this binding")
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().cnl_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what=""):
    """Map the C ABI's error convention onto Python exceptions (reference: bare asserts / exceptions)."""
    if rc == 0:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc in (CNL_E_BAD_ARG, CNL_E_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)


def absmax_stride():
    """Floats between the per-image slots of an x_absmax / y_absmax array (one 128-byte line per image: include/centernet_gfx950.h)."""
    return load().cnl_absmax_stride()


def absmax_buffer(n, device="cuda"):
    """A zeroed per-image maxima array for n images (n * absmax_stride() floats)."""
    import torch
    return torch.zeros((n * absmax_stride(),), device=device, dtype=torch.float32)


def absmax_pack(values):
    """Per-image maxima (a tensor of n floats) -> the strided array the kernels read."""
    buf = absmax_buffer(values.numel(), values.device)
    buf[::absmax_stride()] = values.to(buf.dtype)
    return buf


def absmax_values(buf, n=None):
    """The n per-image values of a strided maxima array."""
    v = buf[::absmax_stride()]
    return v if n is None else v[:n]
