"""Op-level timing of the halo weight gradient (wgrad3_kernel vs the double-buffered wgrad3x_kernel, both + the partial-tile reduce)
on the VNet3d 4x96^3 layer shapes.  The kernel choice is read once per process (SEG_WGRAD3X), so each arm runs in a child."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pytorchdeeplearing_amd import ops
    dev = torch.device("cuda")
    res = {}
    for (N, S, C, dt) in ((4, 96, 16, "f16"), (4, 48, 32, "f16"), (4, 24, 64, "f16"), (4, 12, 128, "f16"), (4, 6, 256, "f16"), (2, 64, 32, "f16"), (1, 80, 32, "bf16")):
        x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).to(ops.TORCH_DTYPE[dt]))
        dr = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).to(ops.TORCH_DTYPE[dt]))
        fn = lambda: ops.wgrad3(dr, x, dt, 3)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10 * 1e3
        res["C%d@%d N%d %s" % (C, S, N, dt)] = [round(t, 1), round(2.0 * N * S ** 3 * 27 * C * C / t * 1e-6, 1)]
    print(json.dumps(res))
else:
    arms = [a.split(",") for a in sys.argv[1:]] or [["SEG_WGRAD3X=0"], ["SEG_WGRAD3X=1"]]
    for arm in arms:
        env = dict(os.environ)
        env.update(dict(kv.split("=") for kv in arm))
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(" ".join(arm), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
