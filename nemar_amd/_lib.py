"""ctypes binding of the C-ABI library (include/nemar_hip.h -> nemar_amd/lib/libnemar_hip.so).

There is NO fallback: if the gfx950 library is missing or fails to load, importing the operator layer
raises.  `load(path)` with an explicit path exists only so the CPU test tier can bind the same signatures to
the host-emulated build of the same kernel sources (tests/emu) — the product never passes a path.

The product library has no measurement switch (nemar_tune*): those entry points (include/nemar_hip_ab.h) exist only in
nemar_amd/lib/libnemar_hip_ab.so, the -DNEMAR_AB build of the same sources, which tools/ and the A/B tests select with
NEMAR_AB_LIBRARY=1 (or NEMAR_TUNE=...) in the environment before the first load.
"""
import ctypes as C
import os

# torch must be imported BEFORE the library is dlopen'ed: PyTorch-ROCm bundles its own libamdhip64, and whichever
# copy is mapped first serves the whole process.  Loading ours first would bring in the system runtime instead of the
# one torch was built against ("no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_PATH = os.path.join(_HERE, "lib", "libnemar_hip.so")
AB_PATH = os.path.join(_HERE, "lib", "libnemar_hip_ab.so")
DEFAULT_PATH = AB_PATH if (os.environ.get("NEMAR_AB_LIBRARY") == "1" or os.environ.get("NEMAR_TUNE")) else PRODUCT_PATH

_f = C.POINTER(C.c_float)
_vp = C.c_void_p
_i = C.c_int
_fl = C.c_float
_sz = C.c_size_t
_ll = C.c_longlong
_db = C.c_double
_u64 = C.c_ulonglong
_u32 = C.c_uint

# name -> (restype, argtypes); mirrors include/nemar_hip.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "nemar_version": (_i, []),
    "nemar_last_error": (C.c_char_p, []),
    "nemar_grid_sample_fwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "nemar_grid_sample_bwd_workspace": (_sz, [_i, _i, _i, _i]),
    "nemar_grid_sample_bwd_zeroed_bytes": (_sz, [_i, _i, _i, _i]),
    "nemar_grid_sample_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nemar_smoothness_workspace": (_sz, [_i, _i, _i]),
    "nemar_smoothness_fwd": (_i, [_vp, _vp, _i, _fl, _fl, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "nemar_smoothness_bwd": (_i, [_vp, _vp, _i, _fl, _vp, _fl, _vp, _i, _i, _i, _i, _vp]),
    "nemar_conv2d_fwd_workspace": (_sz, [_i] * 9),
    "nemar_conv2d_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _vp, _sz, _i, _vp]),
    "nemar_conv2d_bwd_data_workspace": (_sz, [_i] * 10),
    "nemar_conv2d_bwd_data": (_i, [_vp, _vp, _vp, _i, _fl, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                   _i, _vp, _sz, _i, _vp]),
    "nemar_conv2d_bwd_weight_workspace": (_sz, [_i] * 11),
    "nemar_conv2d_bwd_weight": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz,
                                     _vp]),
    "nemar_conv2d_scratch": (_sz, [_i] * 9),
    "nemar_conv2d_gy_planes_bytes": (_sz, [_i] * 10),
    "nemar_absmax": (_i, [_vp, _ll, _vp, _vp]),
    "nemar_absmax_samples": (_i, [_vp, _i, _ll, _vp, _vp]),
    "nemar_kernel_timer": (_i, [_i]),
    "nemar_kernel_timer_read": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "nemar_bias_grad_workspace": (_sz, [_i, _i, _i]),
    "nemar_bias_grad": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "nemar_last_route": (_i, []),
    "nemar_last_gy_planes": (_i, []),
    "nemar_config_epoch": (_i, []),
    "nemar_set_max_words_lazy": (_i, [_i]),
    "nemar_instnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _fl, _i, _fl, _vp]),
    "nemar_instnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _fl, _vp]),
    "nemar_instnorm_fwd_max": (_i, [_vp, _vp, _vp, _vp, _i, _i, _fl, _i, _fl, _vp, _i, _vp]),
    "nemar_instnorm_fwd_planes": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _fl, _i, _fl, _fl, _u64, _u32, _vp, _vp, _vp, _vp, _vp]),
    "nemar_instnorm_bwd_planes": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _fl, _fl, _u64, _u32, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nemar_bias_from_partials": (_i, [_vp, _i, _i, _vp, _vp]),
    "nemar_conv2d_bwd_data_fusable": (_i, [_i] * 10),
    "nemar_conv2d_x_planes_bytes": (_sz, [_i] * 5),
    "nemar_set_dropout_base": (_i, [_vp]),
    "nemar_store_words": (_i, [_vp, _vp, _i, _vp]),
    "nemar_pack_plan_record": (_i, [_i]),
    "nemar_pack_plan_jobs": (_i, [_i]),
    "nemar_pack_plan_bytes": (_sz, [_i]),
    "nemar_pack_plan_dirty": (_i, [_i]),
    "nemar_pack_plan_commit": (_i, [_i, _vp, _sz, _vp]),
    "nemar_pack_plan_run": (_i, [_i, _vp]),
    "nemar_pack_plan_reset": (_i, [_i]),
    "nemar_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _ll, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "nemar_conv2d_fwd_ex": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _vp, _sz, _i, _vp, _vp]),
    "nemar_conv2d_bwd_data_ex": (_i, [_vp, _vp, _vp, _i, _fl, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp, _vp]),
    "nemar_conv2d_bwd_weight_ex": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "nemar_instnorm_bwd_max": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _fl, _vp, _i, _vp]),
    "nemar_act_bwd": (_i, [_vp, _vp, _vp, _ll, _i, _fl, _vp]),
    "nemar_act_fwd": (_i, [_vp, _vp, _ll, _i, _fl, _vp]),
    "nemar_conv2d_bwd_data_addend_ok": (_i, [_i] * 10),
    "nemar_max_words_finalize": (_i, [_vp, _i, _vp]),
    "nemar_concat_pieces": (_i, [_vp, _vp, _i, _vp, _vp]),
    "nemar_add2": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "nemar_maxpool2_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "nemar_maxpool2_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "nemar_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nemar_bilinear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nemar_crop_flip_normalize": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _fl, _vp]),
    "nemar_dropout": (_i, [_vp, _vp, _ll, _fl, _u64, _u32, _vp]),
    "nemar_dropout_max": (_i, [_vp, _vp, _i, _ll, _fl, _u64, _u32, _vp, _vp]),
    "nemar_loss_workspace": (_sz, []),
    "nemar_l1_loss_fwd": (_i, [_vp, _vp, _ll, _fl, _vp, _i, _vp, _sz, _vp]),
    "nemar_l1_loss_bwd": (_i, [_vp, _vp, _ll, _vp, _fl, _vp, _i, _vp]),
    "nemar_gan_loss_fwd": (_i, [_vp, _ll, _i, _i, _fl, _vp, _i, _vp, _sz, _vp]),
    "nemar_gan_loss_bwd": (_i, [_vp, _ll, _i, _i, _vp, _fl, _vp, _vp]),
    "nemar_adam_step": (_i, [_vp, _vp, _vp, _vp, _ll, _db, _db, _db, _db, _i, _vp]),
}


# include/nemar_hip_ab.h: what the measurement build exports on top
AB_SIGNATURES = {
    "nemar_tune": (_i, [_i, _i]),
    "nemar_grid_sample_tune": (_i, [_i]),
    "nemar_tune_ptr": (_i, [_vp]),
}


class NemarHipError(RuntimeError):
    pass


class ConvExtras(C.Structure):
    """include/nemar_hip.h nemar_conv_extras"""
    _fields_ = [("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t), ("src_max_words", C.c_void_p), ("src_max_count", C.c_int),
                ("src2_max_words", C.c_void_p), ("src2_max_count", C.c_int), ("src_planes", C.c_void_p),
                ("gy_planes_out", C.c_void_p), ("gy_planes_bytes", C.c_size_t), ("src2_planes", C.c_void_p),
                ("addend", C.c_void_p), ("out_max_words", C.c_void_p), ("bias_partials", C.c_void_p)]


class Library:
    """Thin wrapper: attribute access returns a checked callable for int-returning entry points."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise NemarHipError(
                "gfx950 operator library not found at %s — build it with `python -m nemar_amd.csrc.build` "
                "(there is no CPU/PyTorch fallback for the NeMAR hot path)" % path)
        self.path = path
        self._dll = C.CDLL(path)
        self._fns = {}
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise NemarHipError("%s does not export %s" % (path, name)) from e
            fn.restype = res
            fn.argtypes = args
            self._fns[name] = fn
        self.has_switches = all(hasattr(self._dll, name) for name in AB_SIGNATURES)
        if self.has_switches:
            for name, (res, args) in AB_SIGNATURES.items():
                fn = getattr(self._dll, name)
                fn.restype = res
                fn.argtypes = args
                self._fns[name] = fn

    def raw(self, name):
        return self._fns[name]

    def last_error(self):
        return self._fns["nemar_last_error"]().decode("utf-8", "replace")

    def __getattr__(self, name):
        fns = self.__dict__.get("_fns", {})
        full = name if name.startswith("nemar_") else "nemar_" + name
        if full not in fns:
            if full in AB_SIGNATURES:
                raise NemarHipError("%s has no %s: the measurement switches exist only in the -DNEMAR_AB build "
                                    "(libnemar_hip_ab.so; set NEMAR_AB_LIBRARY=1 before nemar_amd is imported)" % (self.__dict__.get("path"), full))
            raise AttributeError(name)
        fn = fns[full]
        if {**SIGNATURES, **AB_SIGNATURES}[full][0] is not _i or full in ("nemar_version", "nemar_last_route", "nemar_last_gy_planes", "nemar_config_epoch", "nemar_set_max_words_lazy", "nemar_pack_plan_jobs", "nemar_pack_plan_dirty", "nemar_conv2d_bwd_data_fusable", "nemar_conv2d_bwd_data_addend_ok"):
            return fn

        if os.environ.get("NEMAR_DEBUG_SYNC"):
            # debugging aid: name + integer arguments of every call on stderr, device synchronised after each one, so that a
            # GPU memory fault is attributed to the launch that caused it
            def checked(*a):
                import sys
                sys.stderr.write("[nemar] %s %s\n" % (full, [x for x in a if isinstance(x, (int, float))]))
                sys.stderr.flush()
                rc = fn(*a)
                torch.cuda.synchronize()
                if rc != 0:
                    raise NemarHipError("%s failed (%d): %s" % (full, rc, self.last_error()))
                return rc
            checked.__name__ = full
            self.__dict__[name] = checked
            return checked

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                raise NemarHipError("%s failed (%d): %s" % (full, rc, self.last_error()))
            return rc

        checked.__name__ = full
        self.__dict__[name] = checked
        return checked


_default = None


def load(path=None):
    """Load (once) and return the library.  `path` is for the emulated test build only."""
    global _default
    if path is not None:
        return Library(path)
    if _default is None:
        _default = Library(DEFAULT_PATH)
        # measurement hook: NEMAR_TUNE="21=1,15=0" applies nemar_tune switches once at load (tools/, A/B runs of bench.py)
        for kv in os.environ.get("NEMAR_TUNE", "").split(","):
            if "=" in kv:
                k, v = kv.split("=")
                _default.tune(int(k), int(v))
    return _default
