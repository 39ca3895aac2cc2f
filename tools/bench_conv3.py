"""Op-level timing of the halo conv / wgrad kernels on the GPU (diagnostics)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from pytorchdeeplearing_amd import ops
    dev = torch.device("cuda")
    res = {}
    for (N, S, C) in ((4, 48, 32), (4, 96, 16), (4, 24, 64), (4, 12, 128), (4, 6, 256)):
        x = ops.aligned_like(torch.randn(N, S, S, S, C, device=dev).half())
        w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
        wp = ops.pack(w, "conv_fwd", "f16")
        o = torch.empty_like(x)
        for name, fn in (("conv3", lambda: ops.conv3(x, wp, "f16", 3, C, out=o)), ("wgrad3", lambda: ops.wgrad3(x, x, "f16", 3))):
            if name == "wgrad3" and (os.environ.get("SEG_CONV3_DBG", "0") != "0" or os.environ.get("SEG_CONV3_NT", "0") != "0"):
                continue
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize()
            res["%s C%d@%d" % (name, C, S)] = round(a.elapsed_time(b) / 10 * 1e3, 1)
    print(json.dumps(res))
else:
    for dbg, nt, wl in ((0, 0, -1), (0, 2, 1), (0, 4, 1), (0, 2, 0)):
        env = dict(os.environ, SEG_CONV3_DBG=str(dbg), SEG_CONV3_NT=str(nt))
        if wl >= 0: env["SEG_CONV3_WL"] = str(wl)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("dbg=%2d nt=%d wl=%d" % (dbg, nt, wl), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
