"""Tuning builds of libsegengine (never the product library): recompile ONE source with extra -D flags and link it with the
other objects of the last product build into pytorchdeeplearing_amd/lib/variants/libsegengine_<tag>.so; select it with
SEGENGINE_LIB=<path> for an A/B run on the GPU box (the variants directory travels with gpurun, it is git-ignored).

    python tools/build_variant.py occ4 conv3.hip -DSEG_C3_OCC=4 -DSEG_W3_OCC=4 -DSEG_STEM_OCC=4
    python tools/build_variant.py c3xtrace conv3x.hip,conv3x_f16_3d.hip -DSEG_C3X_TRACE"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorchdeeplearing_amd import build as B

tag, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]          # several sources: comma-separated (a flag that changes a shared struct)
B.build()
objdir = os.path.join(B.HERE, "lib", "obj")
vdir = os.path.join(B.HERE, "lib", "variants")
os.makedirs(vdir, exist_ok=True)
new, procs = [], []
for src in srcs:
    obj = os.path.join(vdir, "%s_%s.o" % (src, tag))
    new.append(obj)
    procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj]))
for p_ in procs:
    assert p_.wait() == 0
objs = [os.path.join(objdir, s_ + ".o") for s_ in B.SRCS if s_ not in srcs] + new
out = os.path.join(vdir, "libsegengine_%s.so" % tag)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
