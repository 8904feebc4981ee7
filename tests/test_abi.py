"""The C-ABI library loads and exports every symbol include/dasr.h declares (no compute calls: no GPU here)."""
import ctypes
import os

from dasr_amd import _lib, build


def test_header_parses_and_is_nontrivial():
    fns = _lib.declared_functions()
    assert len(fns) >= 28
    for name in ("dasr_conv2d_fwd", "dasr_sean_fwd", "dasr_sean_bwd", "dasr_mask_compress", "dasr_weight_pack_fwd",
                 "dasr_region_pool_fwd", "dasr_instnorm_stats", "dasr_dynk_fwd", "dasr_conv2d_wgrad"):
        assert name in fns, name


def test_hip_library_exports_every_declared_symbol():
    path = build.build_hip(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in _lib.declared_functions():
        assert hasattr(lib, name), "libdasr_hip.so does not export %s" % name
    lib.dasr_version.restype = ctypes.c_int
    assert lib.dasr_version() >= 100
    lib.dasr_is_device_build.restype = ctypes.c_int
    assert lib.dasr_is_device_build() == 1
    lib.dasr_error_string.restype = ctypes.c_char_p
    assert lib.dasr_error_string(-2) == b"inconsistent or non-positive sizes"


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, so this is safe on a GPU-less machine."""
    lib = ctypes.CDLL(build.build_hip(verbose=False))
    lib.dasr_add.restype = ctypes.c_int
    lib.dasr_add.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    assert lib.dasr_add(None, None, None, 4, None) == -1          # DASR_E_NULL
    lib.dasr_instnorm_stats_workspace.restype = ctypes.c_size_t
    lib.dasr_instnorm_stats_workspace.argtypes = [ctypes.c_int] * 3
    assert lib.dasr_instnorm_stats_workspace(2, 1000, 64) > 0


def test_product_loader_refuses_emulator_library(monkeypatch):
    emu = build.build_emu()
    monkeypatch.delenv("DASR_HIPEMU_LIB", raising=False)
    monkeypatch.setattr(_lib, "LIB_PATH", emu)
    _lib.reset_for_tests()
    try:
        try:
            _lib.get()
            raised = False
        except RuntimeError as e:
            raised = "emulator" in str(e)
        assert raised, "the product loader must not accept the CPU emulator build"
    finally:
        _lib.reset_for_tests()


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.delenv("DASR_HIPEMU_LIB", raising=False)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    _lib.reset_for_tests()
    try:
        try:
            _lib.get()
            raised = False
        except RuntimeError as e:
            raised = "no fallback" in str(e)
        assert raised
    finally:
        _lib.reset_for_tests()
