"""Per-workgroup phase times of c3x::conv3x_kernel (diagnostic build: python tools/build_variant.py c3xtrace conv3x.hip,conv3x_f16_3d.hip -DSEG_C3X_TRACE; the library
prints one line per launch to stderr): stage = start -> halo in LDS (after the barrier), taps = the tap loops, epilogue = bias + statistics + stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SEGENGINE_LIB", os.path.join(ROOT, "pytorchdeeplearing_amd", "lib", "variants", "libsegengine_c3xtrace.so"))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import ops
dev = torch.device("cuda")
for (N, S, C, cfgs) in ((4, 48, 32, [17, 51]), (4, 24, 64, [-1]), (4, 12, 128, [-1]), (4, 6, 256, [-1])):
    x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    wf = ops.pack(w, "conv_fwd", "f16", frag="all")
    out = ops.aligned_like(torch.empty(N, S, S, S, C, device=dev).half())
    for cfg in cfgs:
        for _ in range(3):
            ops.conv3x(x, wf, "f16", 3, C, want_stats=True, out=out, cfg=cfg)
        torch.cuda.synchronize()
