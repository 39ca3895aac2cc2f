cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6w; mkdir -p $O
cat > /tmp/noise.py <<'PY'
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from pytorchdeeplearing_amd import SegEngine, _capi
from oracle import seg_oracle as seg
from test_engine import CASES
dev = torch.device("cuda:0")
kind, ndim, shape, ncls, loss = CASES["vnet3d"]
x, y = seg.synthetic_batch(shape[0], (32, 32, 32), shape[1], ncls, seed=1)
xd, yd = x.to(dev), y.to(dev)
def run(mode):
    e = SegEngine(kind, ndim, shape[1], ncls, dtype="f16", device=dev)
    e.load_state_dict(seg.perturb_params(seg.init_params(kind, ndim, shape[1], ncls, seed=0), seed=7))
    logits = torch.empty((shape[0], ncls, 32, 32, 32), dtype=torch.float32, device=dev); probs = torch.empty_like(logits)
    curve = [float(e.train_step(xd, yd, loss, lr=1e-3, logits=logits, probs=probs, launch=mode)[0]) for _ in range(5)]
    torch.cuda.synchronize()
    p = e.params.detach().cpu().clone(); del e
    return curve, p
ref = run("stream")
for i in range(6):
    for mode in ("stream", "graph"):
        c, p = run(mode)
        d = (p - ref[1]).abs()
        print(os.environ.get("SEG_C3X16_REUSE", "1"), mode, i, "max %.5f frac>1e-4 %.4f loss diff %.2e" % (float(d.max()), float((d > 1e-4).float().mean()), max(abs(a - b) for a, b in zip(c, ref[0]))), flush=True)
PY
for r in 1 0; do SEG_C3X16_REUSE=$r SEG_VACT=1 timeout 300 python /tmp/noise.py >> $O/noise.log 2>&1; done
cat $O/noise.log | grep -v Warning
