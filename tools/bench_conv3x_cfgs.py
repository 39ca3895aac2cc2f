"""Standalone time of halo-conv tilings on given shapes: python tools/bench_conv3x_cfgs.py "N,S,Cin,Cout:cfg,cfg,..." ...   (default: the 32-channel level of C3 / C4 / C5,
tiling 17 against the persistent tiling 51)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd import ops
dev = torch.device("cuda")
specs = sys.argv[1:] or ["4,48,32,32:17,51,14", "2,64,32,32:17,51", "1,80,32,32:17,51", "4,48,32,64:3,51"]
for spec in specs:
    shp, cfgs = spec.split(":")
    N, S, Cin, Cout = (int(v) for v in shp.split(","))
    x = ops.aligned_like(torch.randn(N, S, S, S, Cin, device=dev).half())
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
    wf = ops.pack(w, "conv_fwd", "f16", frag="all")
    out = ops.aligned_like(torch.empty(N, S, S, S, Cout, device=dev).half())
    res, ref = {}, None
    for cfg in (int(c) for c in cfgs.split(",")):
        for stats in (True, False):
            for _ in range(3):
                o, st = ops.conv3x(x, wf, "f16", 3, Cout, want_stats=stats, out=out, cfg=cfg) if stats else (ops.conv3x(x, wf, "f16", 3, Cout, out=out, cfg=cfg), None)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                ops.conv3x(x, wf, "f16", 3, Cout, want_stats=stats, out=out, cfg=cfg)
            b.record(); torch.cuda.synchronize()
            res["cfg%d%s" % (cfg, "+stats" if stats else "")] = round(a.elapsed_time(b) / 20 * 1e3, 1)
        cur = out.clone()
        if ref is None:
            ref = cur
        else:
            res["cfg%d_equal_first" % cfg] = bool(torch.equal(cur, ref))
    print(json.dumps({"shape": [N, S, Cin, Cout], "us": res}), flush=True)
