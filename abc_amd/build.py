"""Build libabc_hip.so (hipcc, gfx950) in-tree.  `python -m abc_amd.build` or abc_amd.build.build()."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libabc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
         "-I", os.path.join(HERE, "..", "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "abc_hip.h"))
    return hs


# probe kernels of tools/microbench.py (DESIGN.md section 4's issue-rate measurements): off in the product library
MICROBENCH = bool(os.environ.get("ABC_HIP_WITH_MICROBENCH"))


def _compile(src):
    probe = MICROBENCH and src == "abc_microbench.hip"
    obj = os.path.join(OBJDIR, src.replace(".hip", ".probes.o" if probe else ".o"))
    srcp = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(p) for p in [srcp] + _headers())
    if os.path.exists(obj) and os.path.getmtime(obj) > newest:
        return obj, False
    cmd = [HIPCC] + FLAGS + (["-DABC_HIP_WITH_MICROBENCH"] if probe else []) + ["-c", srcp, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in results]
    stamp = os.path.join(OBJDIR, "linked_with_probes" if MICROBENCH else "linked_without_probes")
    if any(changed for _, changed in results) or not os.path.exists(LIB) or not os.path.exists(stamp):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
        for f in ("linked_with_probes", "linked_without_probes"):
            if os.path.exists(os.path.join(OBJDIR, f)):
                os.remove(os.path.join(OBJDIR, f))
        open(stamp, "w").close()
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    if "--microbench" in sys.argv:
        MICROBENCH = True
    build(verbose=True)
    sys.exit(0)
