"""Per-workgroup phase sums of wgrad3_kernel (diagnostic build: python tools/build_variant.py w3trace conv3.hip -DSEG_W3_TRACE; prints to stderr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SEGENGINE_LIB", os.path.join(ROOT, "pytorchdeeplearing_amd", "lib", "variants", "libsegengine_w3trace.so"))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import ops
dev = torch.device("cuda")
for (N, S, C) in ((4, 96, 16), (4, 48, 32), (4, 24, 64), (4, 12, 128)):
    x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
    dr = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
    for _ in range(3):
        ops.wgrad3(dr, x, "f16", 3)
    torch.cuda.synchronize()
