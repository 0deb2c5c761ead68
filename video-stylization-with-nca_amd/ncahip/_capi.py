"""ctypes binding of libncahip.so (C ABI: include/ncahip.h).  No torch types cross this boundary:
only raw device pointers, sizes and a hipStream_t.  Importing this module never touches the GPU;
the library is loaded on first use and a missing library is a hard error (no fallback)."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_uint64, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("NCAHIP_LIB", os.path.join(_PKG_DIR, "libncahip.so"))

PAD_MODES = {"constant": 0, "zeros": 0, "replicate": 1, "circular": 2, "reflect": 3}

_P, _I, _F, _U64 = c_void_p, c_int, c_float, c_uint64

# name -> argtypes, exactly the prototypes of include/ncahip.h
SIGNATURES = {
    "ncahip_version": [],
    "ncahip_last_error": [],
    "ncahip_limits": [_P, _P, _P],
    "ncahip_selftest": [_P, _P],
    "ncahip_check_errors": [_P, _I],
    "ncahip_debug_inject_error": [ctypes.c_uint],
    "ncahip_debug_force_generic": [_I],
    "ncahip_image_encoder_front_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ncahip_edge_extractor_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ncahip_dynca_perceive_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ncahip_cond_perceive_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ncahip_dynca_step_fwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P],
    "ncahip_dynca_nsteps_fwd_f32": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P],
    "ncahip_dynca_step_fwd_ms_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P],
    "ncahip_dynca_nsteps_fwd_ms_f32": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P],
    "ncahip_dynca_step_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P, _P, _P, _P, _P],
    "ncahip_cond_step_fwd_f32": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F,
                                 _F, _F, _U64, _U64, _P],
    "ncahip_cond_finalize_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P],
    "ncahip_cond_alive_u8": [_P, _P, _I, _I, _I, _I, _I, _F, _P],
    "ncahip_cond_grow_fwd_f32": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F,
                                 _F, _F, _F, _U64, _U64, _P],
    "ncahip_cond_precision": [_I],
    "ncahip_dynca_step_fwd_bf16": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P],
    "ncahip_dynca_nsteps_fwd_bf16": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P],
    "ncahip_cond_step_fwd_bf16": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F,
                                  _F, _F, _U64, _U64, _P],
    "ncahip_cond_finalize_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P],
    "ncahip_cond_grow_fwd_bf16": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F,
                                  _F, _F, _F, _U64, _U64, _P],
    "ncahip_philox_uniform_f32": [_P, _I, _I, _I, _U64, _U64, _P],
    "ncahip_pack_fire_mask_u32": [_P, _P, _I, _I, _I, _I, _F, _I, _P],
    "ncahip_dynca_step_bwd_w2_workspace": [_I, _I, _I, _I, _I],
    "ncahip_dynca_step_bwd_w2_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P, _P, _P, _P, _I, _P,
                                     ctypes.c_size_t, _P],
    "ncahip_dynca_nsteps_bwd_workspace": [_I, _I, _I, _I, _I, _I],
    "ncahip_dynca_nsteps_persist_workspace": [_I, _I, _I, _I, _I, _I],
    "ncahip_debug_persist_drop_tiles": [_I],
    "ncahip_dynca_nsteps_fwd_persist_ms_f32": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, ctypes.c_size_t,
                                               ctypes.c_uint, _P],
    "ncahip_dynca_nsteps_fwd_persist_f32": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, ctypes.c_size_t,
                                            ctypes.c_uint, _P],
    "ncahip_dynca_nsteps_bwd_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P, _P, _P, _P, _P, _P,
                                    _P, ctypes.c_size_t, _P],
    "ncahip_dynca_nsteps_bwd_bf16_workspace": [_I, _I, _I, _I, _I, _I],
    "ncahip_dynca_nsteps_bwd_bf16": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P, _P, _P, _P, _P, _P,
                                     _P, ctypes.c_size_t, _P],
    "ncahip_dynca_nsteps_bwd_ms_workspace": [_I, _I, _I, _I, _I, _I],
    "ncahip_dynca_nsteps_bwd_ms_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _U64, _P, _P, _P, _P, _P, _P, _P,
                                       _P, ctypes.c_size_t, _P],
    "ncahip_gram_rows_workspace": [_I, _I, _I, _I],
    "ncahip_gram_rows_f32": [_P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, ctypes.c_size_t, _P],
    "ncahip_cond_grow_bwd_workspace": [_I, _I, _I, _I, _I],
    "ncahip_cond_grow_bwd_f32": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F,
                                 _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "ncahip_cond_grow_bwd_bf16": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F,
                                  _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
}
_RESTYPES = {"ncahip_last_error": c_char_p, "ncahip_cond_grow_bwd_workspace": ctypes.c_size_t,
             "ncahip_gram_rows_workspace": ctypes.c_size_t, "ncahip_dynca_step_bwd_w2_workspace": ctypes.c_size_t,
             "ncahip_dynca_nsteps_bwd_workspace": ctypes.c_size_t, "ncahip_dynca_nsteps_bwd_ms_workspace": ctypes.c_size_t,
             "ncahip_dynca_nsteps_bwd_bf16_workspace": ctypes.c_size_t, "ncahip_dynca_nsteps_persist_workspace": ctypes.c_size_t}

_lib = None


class NcaHipError(RuntimeError):
    pass


EINVAL, ERANGE, EDEVICE = -1, -2, 100001     # include/ncahip.h return codes
SEED_U_IS_BITS = 0x5354494255     # include/ncahip.h NCAHIP_SEED_U_IS_BITS: `u` holds bit-packed fire masks


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NcaHipError(
                f"libncahip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C video-stylization-with-nca_amd`).  There is no CPU fallback for the NCA step.")
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  The runtime that is loaded FIRST serves the whole
        # process: if libncahip.so pulled in /opt/rocm's before torch loaded its own, there would be two runtimes and ours would see
        # "no ROCm-capable device" once torch has initialised the GPU.  Importing torch first makes its runtime the one we bind to.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header and library out of sync
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ncahip_last_error()
        raise NcaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def version():
    return lib().ncahip_version()


def limits():
    c, f, h = c_int(), c_int(), c_int()
    lib().ncahip_limits(ctypes.byref(c), ctypes.byref(f), ctypes.byref(h))
    return c.value, f.value, h.value
