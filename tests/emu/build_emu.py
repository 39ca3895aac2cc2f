"""TEST INFRASTRUCTURE ONLY: compile the SAME HIP sources as the product (device intrinsics shimmed under `#ifdef SEG_EMU`: DPP, buffer_load ... lds,
inline asm, rocPRIM) with the host clang
against tests/emu/hip/hip_runtime.h (wave64 execution-model checker) -> tests/emu/_build/libsegengine_emu.so.
The product package never loads this library; tests inject it explicitly."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pytorchdeeplearing_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libsegengine_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
sys.path.insert(0, ROOT)
from pytorchdeeplearing_amd.build import SRCS          # noqa: E402  (the product's source list)


def build(force=False):
    import fcntl
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(os.path.join(os.path.dirname(OUT), ".lock"), "w") as lk:     # xdist workers / spawned ranks build once
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build(force)


def _build(force=False):
    srcs = [os.path.join(CSRC, s) for s in SRCS]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "conv3x_impl.h"), os.path.join(CSRC, "gn_fold.h"), os.path.join(CSRC, "engine_internal.h"),
                   os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "segengine.h")]
    hdr_t = max(os.path.getmtime(d) for d in deps[len(srcs):])
    # per-object freshness (an object compiled BEFORE an edit of its source must not hide behind a newer link step)
    stale = lambda s, o: force or not os.path.exists(o) or os.path.getmtime(o) <= max(os.path.getmtime(s), hdr_t)
    objs = [os.path.join(HERE, "_build", os.path.basename(s) + ".o") for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if stale(s, o)]
    if not todo and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(o) for o in objs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    procs = []
    for s, o in todo:
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-I", HERE, "-I", CSRC, "-Wno-unused-value",
               "-Wno-vla-cxx-extension", "-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            raise RuntimeError("emu build failed for %s:\n%s" % (s, out))
    subprocess.check_call([CLANG, "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
