"""Build libsegengine.so for gfx950 with hipcc (in-tree, so it travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsegengine.so")
SRCS = ["conv.hip", "conv3.hip", "conv3x.hip", "conv3x_f16_3d.hip", "conv3x_f16_2d.hip", "conv3x_bf16_3d.hip", "conv3x_bf16_2d.hip", "wgrad.hip", "stemx.hip", "norm.hip", "misc.hip", "lovasz.hip", "ssim.hip", "cldice.hip", "prepost.hip", "engine.hip", "engine_plan.hip", "capi_ops.hip"]
ID_UNIT = "engine.hip"          # compiled with -DSEG_BUILD_ID
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -munsafe-fp-atomics: fp32 / fp64 atomicAdd as the hardware instruction instead of a compare-and-swap loop (every buffer the library adds into is ordinary
# device memory).  No -ffp-contract=fast since round 5: the compiler's default for device code (fast-honor-pragmas) contracts the kernels' own a * b + c the
# same way but leaves the math library's carefully ordered sequences alone - the f32 (parity-exact) run dtype should not inherit a global relaxation.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def build_id():
    """sha256 over every source the library is built from (+ the compiler flags), first 12 hex digits: seg_build_info() carries it, the rocprofv3 / PMC
    summaries under profiles/ are stamped with it and bench.py drops a `traffic` figure measured on another binary."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in sorted(_deps()):
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


def _deps():
    d = [os.path.join(CSRC, f) for f in SRCS + ["common.h", "kernels.h", "conv3x_impl.h", "gn_fold.h", "engine_internal.h"]]
    d.append(os.path.join(os.path.dirname(HERE), "include", "segengine.h"))
    return d


def up_to_date():
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _deps())


def _dep_time(obj, fallback):
    """newest mtime among the project headers an object was compiled from (the compiler's own dependency file, written next to the object); the newest of
    ALL headers when there is none yet"""
    try:
        with open(obj + ".d") as f:
            deps = f.read().replace("\\\n", " ").split(":", 1)[1].split()
        root = os.path.dirname(HERE)
        t = [os.path.getmtime(d) for d in deps if os.path.abspath(d).startswith(root) and os.path.exists(d)]
        return max(t) if t else fallback
    except (OSError, IndexError):
        return fallback


def _embedded_id(obj):
    """the build id compiled INTO an object (engine.hip.o carries -DSEG_BUILD_ID as a string constant): compared with the id of the current sources, so
    that a cached object of another checkout is never linked under a newer stamp (a tracked side file used to make that possible)"""
    try:
        with open(obj, "rb") as f:
            data = f.read()
        i = data.find(b"segengine gfx950 ")
        return data[i + 17:i + 29].decode("ascii", "replace") if i >= 0 else ""
    except OSError:
        return ""


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "conv3x_impl.h", "gn_fold.h", "engine_internal.h")] + [os.path.join(os.path.dirname(HERE), "include", "segengine.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    procs, objs = [], []
    bid = build_id()
    for s in SRCS:
        o = os.path.join(objdir, s + ".o")
        objs.append(o)
        # per-object freshness: an object compiled before an edit of its source must not hide behind a newer link step
        stale_id = s == ID_UNIT and _embedded_id(o) != bid          # the unit that carries the build id follows every source change
        if not force and not stale_id and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(os.path.join(CSRC, s)), _dep_time(o, hdr_t)):
            continue
        cmd = [HIPCC] + FLAGS + (['-DSEG_BUILD_ID="%s"' % bid] if s == ID_UNIT else []) + ["-MD", "-MF", o + ".d", "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    if not procs and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(o) for o in objs):
        return LIB
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, out))
        if verbose and out.strip():
            print(out)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
