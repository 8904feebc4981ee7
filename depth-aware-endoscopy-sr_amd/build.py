"""Build libdasr_hip.so (hipcc, gfx950) in-tree: ``python -m dasr_amd.build`` or ``build_hip()``.

The shared object lands next to the sources' package (``depth-aware-endoscopy-sr_amd/libdasr_hip.so``)
so that it travels with the repository snapshot to the GPU box; it is git-ignored.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdasr_hip.so")
ROOT = os.path.dirname(PKG_DIR)

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on",
               "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libdasr_hip.so")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=True, extra_flags=()):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "dasr.h")]
    if not force and not _stale(LIB_PATH, deps):
        return LIB_PATH
    objs = []
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    # DASR_HIPCC_EXTRA: extra compiler flags for experiments (e.g. -DDASR_V2_DEBUG); never set for the product build
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + os.environ.get("DASR_HIPCC_EXTRA", "").split()
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            cmd = [cc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def build_emu(force=False, verbose=False):
    """CPU kernel-emulator build of the same sources (unit tests only; see tests/hipemu/hipemu.h)."""
    emu_dir = os.path.join(ROOT, "tests", "hipemu")
    out = os.path.join(emu_dir, "libdasr_emu.so")
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(emu_dir, "hipemu.*")) + \
        [os.path.join(ROOT, "include", "dasr.h")]
    if not force and not _stale(out, deps):
        return out
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++") or cxx
    objdir = os.path.join(emu_dir, "build")
    os.makedirs(objdir, exist_ok=True)
    base = [cxx, "-std=c++17", "-O2", "-fPIC", "-DDASR_HIPEMU", "-ffp-contract=on", "-I", emu_dir, "-I", CSRC,
            "-Wno-unused-function", "-Wno-unused-value"]
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            cmd = base + ["-x", "c++", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    o = os.path.join(objdir, "hipemu.o")
    objs.append(o)
    procs.append(("hipemu.cpp", subprocess.Popen(base + ["-c", os.path.join(emu_dir, "hipemu.cpp"), "-o", o])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emulator build failed on %s" % s)
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    print(LIB_PATH)
