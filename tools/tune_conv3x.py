"""Per-layer timing of every conv3x tiling (and of conv3_kernel) on the halo-conv shapes of the BASELINE configs.
Run on the GPU box:  python tools/tune_conv3x.py [--sets c3,c4,c5,c2] [--iters 20]  > gpurun_out/tune_conv3x.jsonl
Prints one JSON line per (shape, tiling) and, at the end, the best tiling per shape in SEG_C3X_MAP syntax."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from pytorchdeeplearing_amd import ops  # noqa: E402

# (ndim, N, spatial, Cin, Cout): forward shapes; data-gradient shapes are (Cout -> Cin parts) and are listed explicitly
SETS = {
    "top": ("f16", [(3, 4, 96, 16, 16), (3, 4, 48, 32, 32)]),
    "c3": ("f16", [(3, 4, 48, 32, 32), (3, 4, 24, 64, 64), (3, 4, 12, 128, 128), (3, 4, 6, 256, 256)]),
    "c5": ("bf16", [(3, 1, 80, 32, 32), (3, 1, 40, 64, 64), (3, 1, 20, 128, 128), (3, 1, 10, 256, 256)]),
    "c4": ("f16", [(3, 2, 64, 32, 32), (3, 2, 64, 32, 16), (3, 2, 64, 64, 32), (3, 2, 32, 32, 64), (3, 2, 32, 64, 64), (3, 2, 32, 64, 32),
                   (3, 2, 32, 128, 64), (3, 2, 16, 64, 128), (3, 2, 16, 128, 128), (3, 2, 16, 128, 64), (3, 2, 16, 256, 128),
                   (3, 2, 8, 128, 256), (3, 2, 8, 256, 256), (3, 2, 8, 256, 128), (3, 2, 128, 32, 16)]),
    "c2": ("f16", [(2, 16, 256, 32, 32), (2, 16, 128, 64, 64), (2, 16, 64, 128, 128), (2, 16, 32, 256, 256)]),
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="c3,c5,c4,c2")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda")
    cfgs = ops.conv3x_cfgs(dev)
    best = {}
    for name in args.sets.split(","):
        dtype, shapes = SETS[name]
        tdt = ops.TORCH_DTYPE[dtype]
        for (ndim, N, S, cin, cout) in shapes:
            sp = (S, S, S) if ndim == 3 else (1, S, S)
            g = torch.Generator(device="cpu").manual_seed(S * 131 + cin)
            x = ops.aligned_like((torch.randn((N,) + sp + (cin,), generator=g)).to(dev).to(tdt))
            w = (torch.randn((cout, cin) + (3,) * ndim, generator=g) * 0.05).to(dev)
            wr = ops.pack(w, "conv_fwd", dtype)
            wf = ops.pack(w, "conv_fwd", dtype, frag="all")
            out0 = torch.empty((N,) + sp + (cout,), dtype=tdt, device=dev)
            out0 = ops.aligned_like(out0)
            flops = 2.0 * N * (S ** ndim) * (27 if ndim == 3 else 9) * cin * cout
            t_old = timed(lambda: ops.conv3(x, wr, dtype, ndim, cout, out=out0), args.iters)
            ref = out0.clone()
            key = "%dd N%d S%d %d->%d %s" % (ndim, N, S, cin, cout, dtype)
            print(json.dumps({"set": name, "shape": key, "cfg": "conv3_kernel", "us": round(t_old, 1), "tflops": round(flops / t_old * 1e-6, 1)}), flush=True)
            dflt = ops._capi.lib_for(dev).seg_op_conv3x_default_cfg(ndim, N, sp[0], sp[1], sp[2], cin, cout, ops._capi.DTYPE[dtype])
            for c in cfgs:
                if c["ndim"] != ndim or cout % c["bn"]:
                    continue
                out = ops.aligned_like(torch.zeros_like(out0))
                try:
                    t = timed(lambda: ops.conv3x(x, wf, dtype, ndim, cout, out=out, cfg=c["id"]), args.iters)
                except RuntimeError as ex:
                    print(json.dumps({"shape": key, "cfg": c["id"], "error": str(ex)[:80]}), flush=True)
                    continue
                same = bool(torch.equal(out, ref))
                md = float((out.float() - ref.float()).abs().max())
                rec = {"set": name, "shape": key, "cfg": c["id"], "name": c["name"], "us": round(t, 1), "tflops": round(flops / t * 1e-6, 1),
                       "same_as_conv3": same, "maxdiff": md, "default": c["id"] == dflt}
                print(json.dumps(rec), flush=True)
                if (same or md < 1e-2) and (key not in best or t < best[key][0]):
                    best[key] = (t, c["id"], cin, cout, S, t_old)
    m = ",".join("%d:%d:%d=%d" % (v[2], v[3], v[4], v[1]) for v in best.values())
    print(json.dumps({"best": {k: {"cfg": v[1], "us": round(v[0], 1), "conv3_kernel_us": round(v[5], 1)} for k, v in best.items()}, "SEG_C3X_MAP": m}))


if __name__ == "__main__":
    main()
