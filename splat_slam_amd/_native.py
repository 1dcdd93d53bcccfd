"""ctypes binding of libsplat_hip.so (C ABI declared in include/splat_hip.h).

There is deliberately no fallback: if the HIP library is missing or cannot be loaded the import of the
product path fails with a clear error (a silent CPU path would void every parity claim).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPLAT_HIP_LIB: another BUILD of the same library, for A/B measurements of kernel changes -- scripts/micro/ab_hash.py)
LIB_PATH = os.environ.get("SPLAT_HIP_LIB") or os.path.join(_HERE, "lib", "libsplat_hip.so")

SGR_OPT_FUSED_BLEND = 0
SGR_OPT_UPSTREAM_POSE_JACOBIAN = 1
SGR_OPT_SEGMENT_TEST = 2
SGR_OK, SGR_ERR_INVALID, SGR_ERR_WORKSPACE, SGR_ERR_CAPACITY, SGR_ERR_HIP = 0, -1, -2, -3, -4

_fp = C.c_void_p


class SgrSettings(C.Structure):
    _fields_ = [("num_gaussians", C.c_int32), ("image_height", C.c_int32), ("image_width", C.c_int32),
                ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("bg", _fp), ("viewmatrix", _fp), ("projmatrix", _fp), ("projmatrix_raw", _fp), ("campos", _fp)]


class SgrInputs(C.Structure):
    _fields_ = [("means3D", _fp), ("opacities", _fp), ("shs", _fp), ("colors_precomp", _fp), ("scales", _fp),
                ("rotations", _fp), ("cov3D_precomp", _fp)]


class SgrOutputs(C.Structure):
    _fields_ = [("color", _fp), ("depth", _fp), ("opacity", _fp), ("radii", _fp), ("n_touched", _fp)]


class SgrWorkspace(C.Structure):
    _fields_ = [("saved", _fp), ("saved_bytes", C.c_size_t), ("scratch", _fp), ("scratch_bytes", C.c_size_t),
                ("capacity", C.c_int64), ("counters_clean", C.c_int32), ("max_list_hint", C.c_int32)]


class SgrGradOutputs(C.Structure):
    _fields_ = [("dL_dcolor", _fp), ("dL_ddepth", _fp)]


class SgrGradInputs(C.Structure):
    _fields_ = [("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp), ("dL_dopacities", _fp), ("dL_dshs", _fp),
                ("dL_dcolors_precomp", _fp), ("dL_dscales", _fp), ("dL_drotations", _fp), ("dL_dcov3D_precomp", _fp),
                ("dL_dtau", _fp), ("accumulate", C.c_int32), ("stat_grad_accum", _fp), ("stat_denom", _fp),
                ("stat_max_radii", _fp)]


class SgrMapView(C.Structure):
    _fields_ = [("settings", SgrSettings), ("out", SgrOutputs), ("ws", SgrWorkspace), ("gt_image", _fp), ("gt_depth", _fp),
                ("exposure_a", _fp), ("exposure_b", _fp), ("loss", _fp), ("dL_dimage", _fp), ("dL_ddepth", _fp),
                ("dL_dexposure", _fp), ("dL_dtau", _fp), ("loss_scratch", _fp), ("loss_scratch_bytes", C.c_size_t)]


class SgrBackwardView(C.Structure):
    _fields_ = [("settings", SgrSettings), ("radii", _fp), ("ws", SgrWorkspace), ("dL_dcolor", _fp), ("dL_ddepth", _fp),
                ("dL_dmeans2D", _fp), ("dL_dtau", _fp)]


class SgrAdamGroup(C.Structure):
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("lr", C.c_float), ("skip", C.c_int32),
                ("step", C.c_int64)]


class SgrAdamTensor(C.Structure):
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("n", C.c_int64), ("step", C.c_int64)]


class SgrMapStep(C.Structure):
    _fields_ = [("num_gaussians", C.c_int64), ("scaling", _fp), ("rotation", _fp), ("opacity", _fp), ("scales_out", _fp),
                ("rot_out", _fp), ("opac_out", _fp), ("num_views", C.c_int32), ("forward_only", C.c_int32),
                ("views", C.POINTER(SgrMapView)), ("in_", C.POINTER(SgrInputs)), ("grads", C.POINTER(SgrGradInputs)),
                ("alpha", C.c_float), ("rgb_boundary_threshold", C.c_float), ("adam_groups", C.POINTER(SgrAdamGroup)),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("iso_weight", C.c_float),
                ("exp_rows", C.c_int32), ("exp_row_width", C.c_int32), ("exp_param", _fp), ("exp_grad", _fp),
                ("exp_avg", _fp), ("exp_avg_sq", _fp), ("exp_step", _fp), ("exp_active", _fp), ("exp_lr", C.c_float),
                ("exp_beta1", C.c_float), ("exp_beta2", C.c_float), ("exp_eps", C.c_float), ("grads_clean", C.c_int32)]


class SgrMapRun(C.Structure):
    _fields_ = [("step", SgrMapStep), ("num_iters", C.c_int32), ("num_window", C.c_int32), ("window", C.POINTER(SgrMapView)),
                ("pool_size", C.c_int32), ("picks_per_iter", C.c_int32), ("pool", C.POINTER(SgrMapView)),
                ("picks", C.POINTER(C.c_int32)), ("lr0", C.POINTER(C.c_float)), ("adam_groups", C.POINTER(SgrAdamGroup)),
                ("pool_exp_row", C.POINTER(C.c_int32)), ("n_touched_last_only", C.c_int32), ("pick_ws", C.POINTER(SgrWorkspace))]


class SgrDeformFrame(C.Structure):
    _fields_ = [("frame_idx", C.c_int32), ("rigid", C.c_int32), ("w2c_old", C.c_float * 16), ("c2w_old", C.c_float * 16),
                ("transform", C.c_float * 16), ("quat_wxyz", C.c_float * 4), ("intrinsics", C.c_float * 9),
                ("height", C.c_int32), ("width", C.c_int32), ("depth_new", _fp), ("depth_old", _fp)]


class SgrRowTensor(C.Structure):
    _fields_ = [("in_", _fp), ("out", _fp), ("row_bytes", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/splat_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "sgr_abi_version": (C.c_int, []),
    "sgr_last_error": (C.c_char_p, []),
    "sgr_saved_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "sgr_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "sgr_forward": (C.c_int, [C.POINTER(SgrSettings), C.POINTER(SgrInputs), C.POINTER(SgrOutputs),
                              C.POINTER(SgrWorkspace), C.POINTER(C.c_int64), _fp]),
    "sgr_backward": (C.c_int, [C.POINTER(SgrSettings), C.POINTER(SgrInputs), _fp, C.POINTER(SgrGradOutputs),
                               C.POINTER(SgrGradInputs), C.POINTER(SgrWorkspace), _fp]),
    "sgr_backward_views": (C.c_int, [C.c_int32, C.POINTER(SgrBackwardView), C.POINTER(SgrInputs), C.POINTER(SgrGradInputs), _fp]),
    "sgr_densify_stats": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sgr_query": (C.c_int, [_fp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _fp]),
    "sgr_query_header": (C.c_int, [_fp, C.POINTER(C.c_uint32), _fp]),
    "sgr_header_to_host": (C.c_int, [_fp, _fp, _fp]),
    "sgr_query_stats": (C.c_int, [C.POINTER(SgrWorkspace), C.c_int32, C.c_int32, C.c_int32, _fp, C.POINTER(C.c_int64), _fp]),
    "sgr_query_depth_keys": (C.c_int, [C.POINTER(SgrWorkspace), C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp]),
    "sgr_query_list_histogram": (C.c_int, [C.POINTER(SgrWorkspace), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _fp]),
    "sgr_set_option": (C.c_int, [C.c_int32, C.c_int32]),
    "sgr_get_option": (C.c_int, [C.c_int32]),
    "sgr_profile_enable": (C.c_int, [C.c_uint32]),
    "sgr_profile_read": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "sgr_mapping_loss": (C.c_int, [C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float,
                                   C.c_float, _fp, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "sgr_adam_step": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int64, _fp]),
    "sgr_adam_step_multi": (C.c_int, [C.c_int32, C.POINTER(SgrAdamTensor), C.c_float, C.c_float, C.c_float, C.c_float, _fp]),
    "sgr_activate": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "sgr_gaussian_adam_step": (C.c_int, [C.c_int64, C.POINTER(SgrAdamGroup), C.c_float, C.c_float, C.c_float, C.c_float, _fp]),
    "sgr_gaussian_adam_shard": (C.c_int, [C.c_int64, C.POINTER(SgrAdamGroup), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.c_float, C.c_float, C.c_float, C.c_float, _fp]),
    "sgr_map_views": (C.c_int, [C.c_int32, C.POINTER(SgrMapView), C.POINTER(SgrInputs), C.POINTER(SgrGradInputs),
                                C.c_float, C.c_float, C.c_int32, _fp]),
    "sgr_map_step": (C.c_int, [C.POINTER(SgrMapStep), _fp]),
    "sgr_map_run": (C.c_int, [C.POINTER(SgrMapRun), _fp]),
    "sgr_masked_adam": (C.c_int, [C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float,
                                  C.c_float, _fp]),
    "sgr_deform_points": (C.c_int, [C.c_int64, _fp, C.POINTER(SgrDeformFrame), _fp, _fp, _fp, _fp]),
    "sgr_compact_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "sgr_keep_list": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "sgr_gather_rows": (C.c_int, [C.c_int64, _fp, C.c_int32, C.POINTER(SgrRowTensor), _fp]),
    "sknn_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "sknn_dist2": (C.c_int, [_fp, C.c_int32, _fp, _fp, C.c_size_t, _fp]),
    "se3_exp": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "se3_log": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "se3_inv": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "se3_mul": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp]),
    "se3_act": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp]),
    "se3_adjT": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp]),
    "se3_matrix": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
}

_lib = None


def lib():
    """Loads (once) and returns the ctypes handle.  Raises, never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). splat_slam_amd has no CPU fallback by design.")
        # The caller's device memory and streams are torch's: load torch FIRST so that this process ends up with ONE HIP
        # runtime (torch's bundled libamdhip64). Loading /opt/rocm's copy first gives a second runtime that cannot see
        # the device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.sgr_abi_version() != 10:
            raise ImportError("libsplat_hip.so ABI version mismatch")
        _lib = h
    return _lib


def last_error():
    return lib().sgr_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
