"""ctypes loader for libdasr_hip.so — the only way the product reaches its kernels.

The signatures are read from ``include/dasr.h`` (the C ABI is the contract; this file adds no
second copy of it).  There is NO fallback: if the shared object is missing or lacks a declared
symbol, the first kernel call raises.

Unit tests that exercise kernel sources on the CPU set ``DASR_HIPEMU_LIB`` to the emulator build
(``tests/hipemu/libdasr_emu.so``, see tests/hipemu/hipemu.h); that library reports
``dasr_is_device_build() == 0`` and is refused unless that variable names it explicitly.
"""
import ctypes
import os
import re
import threading

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
HEADER = os.path.join(ROOT, "include", "dasr.h")
LIB_PATH = os.path.join(PKG_DIR, "libdasr_hip.so")

_lock = threading.Lock()
_lib = None
_is_device = None

_CTYPES = {
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "void*": ctypes.c_void_p,
    "const void*": ctypes.c_void_p, "int": ctypes.c_int, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
    "const char*": ctypes.c_char_p, "const unsigned char*": ctypes.c_void_p, "unsigned char*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p, "int*": ctypes.c_void_p,
    "const unsigned short*": ctypes.c_void_p, "unsigned short*": ctypes.c_void_p,      # bf16 bits
    "const dasr_pack_job*": ctypes.c_void_p, "const dasr_split_job*": ctypes.c_void_p,  # job tables (structs below)
}


def declared_functions(header_path=HEADER):
    """Parse ``include/dasr.h`` -> {name: (restype, [argtypes])} (type names as strings)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = {}
    for m in re.finditer(r"(int|size_t|const char\*)\s+(dasr_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                ty = a.rsplit(" ", 1)[0] if not a.endswith("*") else a
                ty = ty.replace(" *", "*")
                if ty not in _CTYPES:
                    raise RuntimeError("dasr.h: cannot map argument %r of %s" % (a, name))
                argtypes.append(ty)
        out[name] = (ret, argtypes)
    return out


def _load():
    global _lib, _is_device
    emu = os.environ.get("DASR_HIPEMU_LIB")
    path = emu if emu else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "dasr_amd: %s not found. The HIP extension is required (there is no fallback path); "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'`." % path)
    lib = ctypes.CDLL(path)
    for name, (ret, argtypes) in declared_functions().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError("dasr_amd: %s does not export %s declared in include/dasr.h" % (path, name))
        fn.restype = _CTYPES[ret]
        fn.argtypes = [_CTYPES[a] for a in argtypes]
    dev = bool(lib.dasr_is_device_build())
    if not dev and not emu:
        raise RuntimeError("dasr_amd: %s is a CPU-emulator build; refusing to use it as the product library" % path)
    _lib, _is_device = lib, dev
    return lib


def get():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _load()
    return _lib


def is_device_build():
    get()
    return _is_device


def reset_for_tests():
    """Drop the cached handle (tests switch between the emulator and the device library)."""
    global _lib, _is_device
    _lib, _is_device = None, None


def stream():
    """Current HIP stream of the caller's device as an integer handle (0 for the emulator).  (The raw-stream query: a step
    issues ~2000 kernels from Python and torch.cuda.current_stream() builds a Stream object every time.)"""
    if _is_device is None:
        get()
    if not _is_device:
        return 0
    return _raw_stream(torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda dev: torch.cuda.current_stream(dev).cuda_stream)


def ptr(t, allow_none=False, dtype=torch.float32):
    """Raw device pointer of a contiguous tensor (fp32 unless told otherwise) that lives where the library computes."""
    if t is None:
        if allow_none:
            return None
        raise ValueError("dasr_amd: required tensor is None")
    if t.dtype != dtype:
        raise TypeError("dasr_amd: expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("dasr_amd: tensor must be contiguous")
    if _is_device is None:
        get()
    if _is_device != t.is_cuda:
        raise RuntimeError("dasr_amd: tensor on %s but library is a %s build"
                           % (t.device, "device" if _is_device else "CPU-emulator"))
    return t.data_ptr()


def check(rc, what):
    if rc != 0:
        msg = get().dasr_error_string(rc)
        raise RuntimeError("dasr_amd: %s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))
