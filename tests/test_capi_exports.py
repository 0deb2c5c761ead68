"""CPU-side checks of the drop-in boundary: libncahip.so loads and exports exactly what
include/ncahip.h declares, the ctypes table matches the header's arity, argument validation
rejects bad shapes without launching (no GPU needed: validation happens before any HIP call)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ncahip.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*(ncahip_\w+)\s*\(([^)]*)\)\s*;", src):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
    return protos


def test_header_declares_the_path():
    p = header_prototypes()
    for name in ("ncahip_dynca_perceive_f32", "ncahip_dynca_step_fwd_f32", "ncahip_dynca_nsteps_fwd_f32",
                 "ncahip_cond_step_fwd_f32", "ncahip_cond_finalize_f32", "ncahip_cond_grow_fwd_f32",
                 "ncahip_cond_alive_u8", "ncahip_cond_perceive_f32", "ncahip_version", "ncahip_last_error"):
        assert name in p, name


def test_library_exports_every_declared_symbol():
    from ncahip import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_capi.LIB_PATH)
    protos = header_prototypes()
    for name, nargs in protos.items():
        assert hasattr(L, name), f"{name} declared in ncahip.h but not exported"
        assert name in _capi.SIGNATURES, f"{name} missing from the ctypes table"
        assert len(_capi.SIGNATURES[name]) == nargs, (name, len(_capi.SIGNATURES[name]), nargs)
    assert set(_capi.SIGNATURES) == set(protos)
    assert _capi.version() == 300
    assert _capi.limits()[0] >= 16


def test_argument_validation_without_gpu():
    """Bad arguments are refused on the host before any launch (negative codes, message set)."""
    from ncahip import _capi
    L = _capi.lib()
    one = ctypes.c_void_p(0x1000)  # never dereferenced: validation fails first
    rc = L.ncahip_dynca_perceive_f32(None, one, 1, 4, 8, 8, 1, None)
    assert rc == -1 and b"null" in L.ncahip_last_error()
    rc = L.ncahip_dynca_perceive_f32(one, ctypes.c_void_p(0x2000), 1, 4, 8, 8, 7, None)
    assert rc == -1
    rc = L.ncahip_dynca_step_fwd_f32(one, ctypes.c_void_p(0x2000), None, None, one, one, one, one,
                                     1, 64, 8, 8, 96, 0, 1, 0.5, 0, 0, None)
    assert rc == -2 and b"exceeds" in L.ncahip_last_error()  # C=64 outside the instantiated range
    rc = L.ncahip_dynca_step_fwd_f32(one, one, None, None, one, one, one, one, 1, 12, 8, 8, 96, 0, 1, 0.5, 0, 0, None)
    assert rc == -1 and b"alias" in L.ncahip_last_error()
    rc = L.ncahip_cond_step_fwd_f32(one, None, ctypes.c_void_p(0x2000), one, None, 3, None, one, one, one, one, one,
                                    one, 1, 12, 8, 8, 64, 3, 0.1, 0.5, -10.0, 10.0, 0, 0, None)
    assert rc == -1 and b"goal" in L.ncahip_last_error()
    with pytest.raises(_capi.NcaHipError):
        _capi.check(rc, "cond_step")
    # bf16-storage entry points: W % 4 and 8-byte alignment are refused up front (there is no any-shape bf16 kernel)
    two = ctypes.c_void_p(0x2000)
    rc = L.ncahip_cond_step_fwd_bf16(one, None, two, one, None, 0, None, one, one, one, one, one, one,
                                     1, 16, 8, 10, 64, 3, 0.1, 0.5, -10.0, 10.0, 0, 0, None)
    assert rc == -2 and b"W % 4" in L.ncahip_last_error()
    rc = L.ncahip_cond_step_fwd_bf16(ctypes.c_void_p(0x1002), None, two, one, None, 0, None, one, one, one, one, one, one,
                                     1, 16, 8, 8, 64, 3, 0.1, 0.5, -10.0, 10.0, 0, 0, None)
    assert rc == -2
    # DyNCA forward and backward accept C <= 32 (BASELINE configs[4]), nothing beyond
    rc = L.ncahip_dynca_step_fwd_f32(one, two, None, None, one, one, one, one, 1, 33, 8, 8, 96, 0, 1, 0.5, 0, 0, None)
    assert rc == -2
    rc = L.ncahip_dynca_step_bwd_f32(one, None, None, one, one, one, one, 1, 33, 8, 8, 96, 0, 1, 0.5, 0, 0,
                                     one, two, one, one, one, None)
    assert rc == -2 and b"exceeds" in L.ncahip_last_error()
    # the T-step backward driver: workspace size is a pure function of the shape; a short workspace is refused
    need = L.ncahip_dynca_nsteps_bwd_workspace(2, 32, 64, 64, 256, 3)
    assert need > 2 * 32 * 64 * 64 * 4 * (2 + 4 + 4) and L.ncahip_dynca_nsteps_bwd_workspace(0, 32, 64, 64, 256, 3) == 0
    rc = L.ncahip_dynca_nsteps_bwd_f32(one, 2, two, None, one, one, one, one, 2, 32, 64, 64, 256, 3, 1, 0.5, 0, 0, one, None,
                                       two, one, one, one, one, ctypes.c_void_p(0x4000), need - 1, None)
    assert rc == -1 and b"workspace" in L.ncahip_last_error()
    # wide hidden layers (fc > 128) are an fp32-forward feature (one launch per 128-wide slice): bf16 storage and the
    # backward refuse them, and so does the forward beyond 1024
    rc = L.ncahip_dynca_step_fwd_bf16(one, two, None, None, one, one, one, one, 1, 16, 8, 8, 256, 0, 1, 0.5, 0, 0, None)
    assert rc == -2 and b"exceeds" in L.ncahip_last_error()
    rc = L.ncahip_dynca_step_fwd_f32(one, two, None, None, one, one, one, one, 1, 16, 8, 8, 2048, 0, 1, 0.5, 0, 0, None)
    assert rc == -2 and b"exceeds" in L.ncahip_last_error()
    rc = L.ncahip_dynca_step_bwd_f32(one, None, None, one, one, one, one, 1, 16, 8, 8, 256, 0, 1, 0.5, 0, 0,
                                     one, two, one, one, one, None)
    assert rc == -2
    # two-scale entry points: odd sizes and C > 16 are refused (the Python layer then composes)
    rc = L.ncahip_dynca_step_fwd_ms_f32(one, two, None, None, one, one, one, one, 1, 12, 9, 8, 96, 0, 1, 0.5, 0, 0, one, None)
    assert rc == -2 and b"even" in L.ncahip_last_error()
    rc = L.ncahip_dynca_step_fwd_ms_f32(one, two, None, None, one, one, one, one, 1, 24, 8, 8, 96, 0, 1, 0.5, 0, 0, one, None)
    assert rc == -2
    need = L.ncahip_dynca_nsteps_bwd_ms_workspace(1, 12, 8, 8, 96, 2)
    rc = L.ncahip_dynca_nsteps_bwd_ms_f32(one, 1, two, None, one, one, one, one, 1, 12, 8, 7, 96, 2, 1, 0.5, 0, 0, one, None,
                                          two, one, one, one, one, ctypes.c_void_p(0x4000), need, None)
    assert rc == -2
    # bf16-history backward: 8-byte aligned states / goal, W % 4
    rc = L.ncahip_cond_grow_bwd_bf16(ctypes.c_void_p(0x1004), one, 1, None, 0, None, one, one, one, one, one, one, 1, 16, 8, 8, 64, 3, 0.1,
                                     0.5, -10.0, 10.0, 0, 0, two, ctypes.c_void_p(0x3000), None, one, one, one, one, one, one,
                                     ctypes.c_void_p(0x4000), 1 << 30, None)
    assert rc == -2 and b"aligned" in L.ncahip_last_error()
    # conditioning front ends
    assert L.ncahip_image_encoder_front_f32(one, one, one, two, 1, 9, 8, 8, None) == -2
    assert L.ncahip_edge_extractor_f32(one, one, one, 1, 8, 8, 1, None) == -1      # aliased in/out
    # error-word entry points exist and fail soft without a device (no GPU in the build container: nothing to map)
    assert L.ncahip_debug_inject_error(1) in (0, -1)
    assert isinstance(L.ncahip_check_errors(None, 1), int)
    # precision switch: only 0 (exact) and 1 (bf16x3)
    assert L.ncahip_cond_precision(0) == 0 and L.ncahip_cond_precision(5) == -1


def test_hot_path_refuses_cpu_tensors():
    """No CPU fallback: the op wrappers fail loudly for non-CUDA tensors."""
    import torch
    from ncahip import ops, _capi
    with pytest.raises(_capi.NcaHipError):
        ops.dynca_perceive(torch.zeros(1, 4, 8, 8))
