"""ctypes binding of libmachisplin_hip.so (the C ABI declared in include/machisplin_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible when a compute entry point is reached, the call raises.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmachisplin_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_NODEVICE, ERR_NUMERIC, ERR_ALLOC = range(6)
F64, F32, I16 = 0, 1, 2
GCV_FIELDS, GCV_CONVERGED = 0, 1


class MhsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[mhs status {code}] {msg}")
        self.code = code


class Grid(C.Structure):
    """struct mhs_grid"""
    _fields_ = [("xmin", C.c_double), ("ymax", C.c_double), ("xres", C.c_double),
                ("yres", C.c_double), ("nrow", C.c_int64), ("ncol", C.c_int64)]


class Stack(C.Structure):
    """struct mhs_stack"""
    _fields_ = [("data", C.c_void_p), ("n_layers", C.c_int32), ("dtype", C.c_int32),
                ("plane_stride", C.c_int64), ("ld", C.c_int64), ("nodata", C.c_double)]


class TiffInfo(C.Structure):
    """struct mhs_tiff_info"""
    _fields_ = [("width", C.c_int64), ("height", C.c_int64), ("bits", C.c_int32), ("sample_format", C.c_int32),
                ("compression", C.c_int32), ("n_ifd", C.c_int32), ("dtype", C.c_int32), ("has_geo", C.c_int32),
                ("nodata", C.c_double), ("xmin", C.c_double), ("ymax", C.c_double), ("xres", C.c_double),
                ("yres", C.c_double)]


class MltpsInfo(C.Structure):
    """struct mhs_mltps_info"""
    _fields_ = [("rsq_model", C.c_double), ("rsq_final", C.c_double), ("lambda_", C.c_double), ("n_knots", C.c_int64),
                ("tiles_rows", C.c_int64), ("tiles_cols", C.c_int64), ("used_tps", C.c_int32), ("n_slots", C.c_int32),
                ("collective", C.c_int32), ("reserved_", C.c_int32), ("band_r0", C.c_int64 * 16), ("band_r1", C.c_int64 * 16),
                ("band_ms", C.c_double * 16), ("tiles_ms", C.c_double * 16), ("fit_ms", C.c_double), ("step_ms", C.c_double),
                ("upload_ms", C.c_double), ("download_ms", C.c_double), ("suggested_slot0_share", C.c_double),
                ("tiles_pulled_bytes", C.c_int64 * 16), ("tiles_owned", C.c_int32 * 16)]


class Unit(C.Structure):
    """struct mhs_unit"""
    _fields_ = [("models", C.POINTER(C.c_void_p)), ("weights", C.POINTER(C.c_double)), ("n_models", C.c_int32),
                ("reserved_", C.c_int32), ("wt_total", C.c_double), ("X", C.c_void_p), ("resp", C.c_void_p), ("n", C.c_int64)]


class UnitsInfo(C.Structure):
    """struct mhs_units_info"""
    _fields_ = [("n_slots", C.c_int32), ("reserved_", C.c_int32), ("n_units", C.c_int64), ("step_ms", C.c_double),
                ("unit_ms_sum", C.c_double), ("unit_ms_max", C.c_double), ("slot_ms", C.c_double * 16)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); every symbol include/machisplin_hip.h declares
SIGNATURES = {
    "mhs_last_error": (C.c_char_p, []),
    "mhs_version": (C.c_char_p, []),
    "mhs_init": (C.c_int, [C.c_int]),
    "mhs_shutdown": (C.c_int, []),
    "mhs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mhs_sync": (C.c_int, [_vp]),
    "mhs_tps_reduction_cache": (C.c_int, [C.c_int]),
    "mhs_fit_reserve_cus": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mhs_timer_start": (C.c_int, [_vp]),
    "mhs_timer_stop": (C.c_int, [_vp, _dp]),
    "mhs_tps_fit": (C.c_int, [_vp, _vp, _i64, C.c_double, C.c_int, C.POINTER(_vp)]),
    "mhs_tps_fit_many": (C.c_int, [_vp, _vp, _vp, _i64, C.c_double, C.c_int, _vp, _vp]),
    "mhs_host_gcv_tridiag": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_double, C.c_double, C.c_int,
                                       _dp, _dp, _dp, _vp]),
    "mhs_host_gcv_band": (C.c_int, [_vp, C.c_int, _vp, _i64, _i64, _i64, C.c_double, C.c_double, C.c_int,
                                    _dp, _dp, _dp, _vp]),
    "mhs_band32_reduce": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "mhs_band32_gcv_terms": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "mhs_band32_solve": (C.c_int, [_vp, _vp, _i64, C.c_double, _vp]),
    "mhs_tps_from_coef": (C.c_int, [_vp, _vp, _vp, _i64, C.c_double, _vp, _vp, C.POINTER(_vp)]),
    "mhs_tps_size": (C.c_int, [_vp, C.POINTER(_i64)]),
    "mhs_tps_get": (C.c_int, [_vp, _vp, _vp, _vp, _dp, _vp, _vp, _dp, _dp]),
    "mhs_tps_free": (C.c_int, [_vp]),
    "mhs_tps_predict_grid": (C.c_int, [_vp, C.POINTER(Grid), _i64, _i64, _i64, _i64, _vp]),
    "mhs_tps_predict_grid_dev": (C.c_int, [_vp, C.POINTER(Grid), _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "mhs_tps_predict_rows_dev": (C.c_int, [_vp, C.POINTER(Grid), _i64, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "mhs_tps_predict_points": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mhs_tps_eval_mode": (C.c_int, [C.c_int]),
    "mhs_tps_eval_plan": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(_i64)]),
    "mhs_lm_load": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "mhs_lm_fit": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "mhs_svr_fit": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "mhs_nnet_fit": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp, _vp, _vp]),
    "mhs_nnet_load": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(_vp)]),
    "mhs_earth_load": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "mhs_svr_load": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_double, C.c_double, _vp, _vp, C.c_double,
                               C.c_double, C.POINTER(_vp)]),
    "mhs_gbm_load": (C.c_int, [C.c_double, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "mhs_rf_load": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "mhs_model_free": (C.c_int, [_vp]),
    "mhs_predict_dev": (C.c_int, [_vp, C.POINTER(Grid), C.POINTER(Stack), _i64, _i64, _i64, _i64,
                                  C.c_double, C.c_int, _vp, _i64, _vp]),
    "mhs_members_predict_dev": (C.c_int, [C.POINTER(_vp), _dp, C.c_int, C.POINTER(Grid), C.POINTER(Stack), C.c_int64, C.c_int64,
                                          C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mhs_ensemble_predict_dev": (C.c_int, [C.POINTER(_vp), _dp, C.c_int, C.c_double, C.POINTER(Grid),
                                           C.POINTER(Stack), _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "mhs_ensemble_predict": (C.c_int, [C.POINTER(_vp), _dp, C.c_int, C.c_double, C.POINTER(Grid),
                                       C.POINTER(Stack), _i64, _i64, _i64, _i64, _vp]),
    "mhs_predict_points": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mhs_gbm_staged_points": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "mhs_model_info": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mhs_gbm_probe_last": (C.c_int, [_vp, _vp, _vp]),
    "mhs_residual_points": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, _i64, _vp]),
    "mhs_scale_add_dev": (C.c_int, [_vp, C.c_double, _vp, _vp, _i64, _vp]),
    "mhs_crop_window": (C.c_int, [C.POINTER(Grid), _vp, _vp]),
    "mhs_step3_tile_windows": (C.c_int, [C.POINTER(Grid), _i64, C.c_double, C.c_double, C.POINTER(_i64),
                                         C.POINTER(_i64), _vp, _vp, _i64]),
    "mhs_tiles_create_windows": (C.c_int, [C.POINTER(Grid), _i64, _i64, C.c_double, _vp, _vp]),
    "mhs_seam_count": (C.c_int, [_i64, _i64, C.POINTER(_i64)]),
    "mhs_cells_from_xy": (C.c_int, [C.POINTER(Grid), _vp, _i64, _vp, _vp]),
    "mhs_mosaic_feather_dev": (C.c_int, [C.POINTER(Grid), _i64, _i64, _vp, C.POINTER(_vp), C.c_int, _vp, _i64,
                                         _vp, _vp]),
    "mhs_mosaic_feather": (C.c_int, [C.POINTER(Grid), _i64, _i64, _vp, C.POINTER(_vp), C.c_int, _vp]),
    "mhs_gather_cells_dev": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "mhs_tiff_info_read": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(TiffInfo)]),
    "mhs_tiff_read_host": (C.c_int, [C.c_char_p, C.c_int, _vp, _i64]),
    "mhs_tiff_read_dev": (C.c_int, [C.c_char_p, C.c_int, _vp, _i64, _vp]),
    "mhs_tiff_write_f32_host": (C.c_int, [C.c_char_p, C.POINTER(Grid), _vp, C.c_double, C.c_int]),
    "mhs_tiff_write_f32_dev": (C.c_int, [C.c_char_p, C.POINTER(Grid), _vp, _i64, C.c_double, C.c_int, _vp]),
    "mhs_tfw_read": (C.c_int, [C.c_char_p, _dp]),
    "mhs_tps_surface": (C.c_int, [C.POINTER(Grid), _vp, _vp, _i64, _vp, _i64, C.c_double, C.c_int, _vp, _vp]),
    "mhs_tps_tiles_dev": (C.c_int, [C.POINTER(Grid), _vp, _vp, _i64, _vp, _i64, C.c_double, C.c_int, _vp, _i64, _vp]),
    "mhs_tps_surface_dev": (C.c_int, [C.POINTER(Grid), _vp, _vp, _i64, _vp, _i64, C.c_double, C.c_int, _vp, _i64,
                                      _vp, _vp]),
    "mhs_init_devices": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mhs_device_slots": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mhs_multi_stack_create": (C.c_int, [C.POINTER(Grid), C.POINTER(Stack), C.c_double, C.POINTER(_vp)]),
    "mhs_multi_stack_free": (C.c_int, [_vp]),
    "mhs_plan_row_bands": (C.c_int, [_i64, C.c_int, C.c_double, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "mhs_multi_trim": (C.c_int, []),
    "mhs_multi_stack_bands": (C.c_int, [_vp, C.POINTER(C.c_int), _vp, _vp]),
    "mhs_mltps_grid_multi_dev": (C.c_int, [C.POINTER(_vp), _dp, C.c_int, C.c_double, _vp, _vp, _vp, _i64, _i64, C.c_double,
                                           C.c_int, C.c_int, C.POINTER(MltpsInfo)]),
    "mhs_multi_final_download": (C.c_int, [_vp, _vp]),
    "mhs_multi_final_dev": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_vp)]),
    "mhs_mltps_grid_multi": (C.c_int, [C.POINTER(_vp), _dp, C.c_int, C.c_double, C.POINTER(Grid), C.POINTER(Stack), _vp, _vp,
                                       _i64, _i64, C.c_double, C.c_int, C.c_double, _vp, C.POINTER(MltpsInfo)]),
    "mhs_tiles_units_multi": (C.c_int, [C.POINTER(Grid), C.POINTER(Stack), _i64, _i64, C.c_double, C.c_int, C.POINTER(Unit),
                                        C.c_int, _i64, C.c_double, C.c_int, C.POINTER(_vp), _vp, C.POINTER(UnitsInfo)]),
}

_lock = threading.Lock()
_lib = None
_inited_device = None


def _hip_runtimes_mapped():
    libs = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    libs.add(line.split()[-1])
    except OSError:
        pass
    return libs


def load() -> C.CDLL:
    """dlopen the library (once) and attach the prototypes.  No GPU needed."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C machisplin_amd/csrc`).  machisplin_amd has no CPU fallback.")
        # torch ships its own libamdhip64 (same SONAME): import it first so this library
        # binds to the runtime torch's allocator and streams live in.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        mapped = _hip_runtimes_mapped()
        if len(mapped) > 1:
            raise ImportError(f"two HIP runtimes are mapped into this process: {sorted(mapped)}")
        _lib = lib
        return lib


def check(rc: int) -> None:
    if rc != OK:
        raise MhsError(rc, load().mhs_last_error().decode(errors="replace"))


def init(device: int | None = None) -> int:
    """Select the GPU (default: LOCAL_RANK or 0) and bring the library up on it."""
    global _inited_device
    lib = load()
    if device is None:
        if _inited_device is not None:   # an explicit earlier init() wins
            return _inited_device
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _inited_device == device:
        return device
    check(lib.mhs_init(int(device)))
    if _inited_device is None:
        # the library's streams (the CU-masked ones of mhs_fit_reserve_cus among them) are destroyed while the HIP
        # runtime is still whole: left to process teardown they crash rocprofv3's finalisation
        atexit.register(_shutdown)
    _inited_device = device
    return device


def _shutdown() -> None:
    global _inited_device
    if _lib is not None and _inited_device is not None:
        _lib.mhs_shutdown()
        _inited_device = None


def lib() -> C.CDLL:
    """The initialised library; raises MhsError(ERR_NODEVICE) without a gfx950 GPU."""
    if _inited_device is None:
        init()
    return load()


def ptr(a) -> int:
    """Raw address of a numpy array (host) or torch tensor (host or device)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data
