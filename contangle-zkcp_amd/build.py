"""Build the HIP extension (libzkcp_amd.so) for gfx950 in-tree, plus -- for the CPU test tier
only -- the emulator build under tests/emu/.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libzkcp_amd.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libzkcp_emu.so")

SOURCES = ["zk_api.cc"]
HEADERS = ["zk_params.h", "zk_field.h", "zk_curve.h", "zk_kernels.h", "zk_rt.h"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "zkcp_amd.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def build_hip(force=False, verbose=False):
    if not force and not _newer(LIB, _deps()):
        return LIB
    cmd = [hipcc_path(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-fgpu-rdc" if False else "-fno-gpu-rdc",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_emu(force=False, verbose=False, sanitize=False):
    """TEST INFRASTRUCTURE: the same sources against tests/emu/emu_hip.h (g++)."""
    out = EMU_LIB if not sanitize else EMU_LIB.replace(".so", "_ubsan.so")
    deps = _deps() + [os.path.join(EMU_DIR, "emu_hip.h"), os.path.join(EMU_DIR, "emu_hip.cpp")]
    if not force and not _newer(out, deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DZK_EMU", "-fvisibility=hidden",
           "-I" + EMU_DIR, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    if sanitize:
        cmd += ["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-g"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(EMU_DIR, "emu_hip.cpp"), "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    what = sys.argv[1:] or ["hip"]
    if "hip" in what:
        print(build_hip(force="-f" in what, verbose=True))
    if "emu" in what:
        print(build_emu(force="-f" in what, verbose=True))
