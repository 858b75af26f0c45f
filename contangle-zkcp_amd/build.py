"""Build the HIP extension (libzkcp_amd.so) for gfx950 in-tree, plus -- for the CPU test tier
only -- the emulator build under tests/emu/.  hipcc cross-compiles without a GPU.  One
translation unit per curve / field, compiled in parallel."""
import concurrent.futures
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

CURVES = ["Pallas", "Vesta", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"]
FIELDS = ["PallasFp", "PallasFq", "Bn254Fr", "Bls381Fr"]
# (source, extra define, object tag)
UNITS = [("zk_api.cc", None, "api"), ("zk_wire.cc", None, "wire")] + \
        [("zk_msm_inst.cc", "ZK_CURVE=" + c, "msm_" + c) for c in CURVES] + \
        [("zk_ntt_inst.cc", "ZK_FIELD=" + f, "ntt_" + f) for f in FIELDS]


def _all_sources():
    """every header / source under csrc/ plus the public header: the conservative dependency set (used for the library
    as a whole, and for an object whose compiler-written .d file does not exist yet)"""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inl", ".cc"))) + \
        [os.path.join(ROOT, "include", "zkcp_amd.h")]


def _deps(obj=None):
    """dependencies of one object = what the compiler recorded the last time it built it (-MMD -MF obj.d): an edit to
    the lazy-limb bucket arithmetic rebuilds the curve units, an NTT edit does not."""
    if obj is not None and os.path.exists(obj + ".d"):
        txt = open(obj + ".d").read().replace("\\\n", " ")
        names = txt.split(":", 1)[1].split() if ":" in txt else []
        return [n for n in names if not n.startswith("/opt/") and not n.startswith("/usr/")]
    return _all_sources()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def _compile_all(base_cmd, objdir, extra_deps, verbose, jobs):
    os.makedirs(objdir, exist_ok=True)
    objs, todo = [], []
    for src, define, tag in UNITS:
        obj = os.path.join(objdir, tag + ".o")
        objs.append(obj)
        if _newer(obj, _deps(obj) + extra_deps):
            cmd = base_cmd + (["-D" + define] if define else []) + ["-MMD", "-MF", obj + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
            todo.append(cmd)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda c: _run(c, verbose), todo))
    return objs


def build_hip(force=False, verbose=False, jobs=None):
    deps = _deps()
    if not force and not _newer(LIB, deps):
        return LIB
    jobs = jobs or min(len(UNITS), os.cpu_count() or 4)
    objdir = os.path.join(PKG, "build", "hip")
    if force:
        shutil.rmtree(objdir, ignore_errors=True)
    base = [hipcc_path(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
            "-fno-gpu-rdc", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs = _compile_all(base, objdir, [], verbose, jobs)
    _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB + ".tmp"] + objs + ["-lhiprtc", "-ldl"], verbose)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_emu(force=False, verbose=False, sanitize=False, jobs=None):
    """TEST INFRASTRUCTURE: the same sources against tests/emu/emu_hip.h (g++)."""
    out = EMU_LIB if not sanitize else EMU_LIB.replace(".so", "_ubsan.so")
    deps = _deps() + [os.path.join(EMU_DIR, "emu_hip.h"), os.path.join(EMU_DIR, "emu_hip.cpp")]
    if not force and not _newer(out, deps):
        return out
    jobs = jobs or min(len(UNITS), os.cpu_count() or 4)
    objdir = os.path.join(PKG, "build", "emu_ubsan" if sanitize else "emu")
    if force:
        shutil.rmtree(objdir, ignore_errors=True)
    # (the sanitized build is -O1 without debug info: at -O2 -g the largest unit alone compiles for six minutes)
    base = ["g++", "-O1" if sanitize else "-O2", "-std=c++17", "-fPIC", "-DZK_EMU", "-fvisibility=hidden", "-I" + EMU_DIR,
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    if sanitize:
        base += ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    objs = _compile_all(base, objdir, [os.path.join(EMU_DIR, "emu_hip.h")], verbose, jobs)
    link = ["g++", "-shared", "-fPIC"] + (["-fsanitize=undefined"] if sanitize else []) + \
           ["-I" + EMU_DIR, "-O2", "-std=c++17", os.path.join(EMU_DIR, "emu_hip.cpp")] + objs + ["-o", out + ".tmp"]
    _run(link, verbose)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    what = sys.argv[1:] or ["hip"]
    if "hip" in what:
        print(build_hip(force="-f" in what, verbose=True))
    if "emu" in what:
        print(build_emu(force="-f" in what, verbose=True))
