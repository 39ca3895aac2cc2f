"""Per-workgroup phase timeline of the halo conv kernel (SEG_CONV3_TRACE=1): prints one line per shape to stderr."""
import os, sys
os.environ["SEG_CONV3_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import ops
dev = torch.device("cuda")
for (N, S, C) in ((4, 48, 32), (4, 96, 16), (4, 24, 64), (4, 12, 128), (4, 6, 256)):
    x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    wp = ops.pack(w, "conv_fwd", "f16")
    o = torch.empty_like(x)
    for _ in range(3):
        ops.conv3(x, wp, "f16", 3, C, out=o)
    torch.cuda.synchronize()
