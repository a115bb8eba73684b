"""Build libope.so (HIP, gfx950) in-tree with hipcc. No torch dependency: the library is a plain C-ABI .so.

    python -m offpolicy_amd.build        # or: python off-policy_amd/build.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libope.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


SAN_LIB = os.path.join(HERE, "libope_asan.so")
SAN_FLAGS = ["-O1", "-g", "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer"]


def sanitizer_runtime():
    """The AddressSanitizer runtime that must be LD_PRELOADed into a Python process that loads libope_asan.so."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


EXP_LIB = os.path.join(HERE, "libope_exp.so")


def build(force=False, verbose=True, extra_flags=(), sanitize=False, experiments=False):
    """sanitize=True: the HOST side of the same sources (argument checks, layout / workspace planning, launch orchestration: the
    C-ABI shim) instrumented with AddressSanitizer + UndefinedBehaviorSanitizer into libope_asan.so (objects under csrc/asan/);
    device code is compiled as usual. tests/test_sanitizer_host.py drives every entry point that returns before its first launch
    through it ("sanitizer build of the shim")."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ope.h"))
    objs = []
    procs = []
    # experiments=True: libope_exp.so, the same sources with -DOPE_EXPERIMENTS -- the timing-only kernel variants behind OPE_WGRAD_EXP /
    # OPE_W2_EXP / OPE_T4_EXP / OPE_WIDE_EXP (loops with the loads, the MFMAs or the reductions left out: WRONG results, announced on stderr). They exist
    # only there; load it with OPE_LIB_PATH for a decomposition run (tools/, profiles/r05_*). libope.so has none of them compiled in.
    lib = SAN_LIB if sanitize else (EXP_LIB if experiments else LIB)
    flags = [f for f in FLAGS if f != "-O3"] + SAN_FLAGS if sanitize else (FLAGS + ["-DOPE_EXPERIMENTS"] if experiments else FLAGS)
    odir = os.path.join(CSRC, "asan") if sanitize else (os.path.join(CSRC, "exp") if experiments else CSRC)
    os.makedirs(odir, exist_ok=True)
    for src in sources():
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + flags + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + (["-fsanitize=address,undefined"] if sanitize else []) + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print("built", build(force="--force" in sys.argv, sanitize="--sanitize" in sys.argv, experiments="--experiments" in sys.argv))
