"""soft-clDice forward+backward time on one GPU at the BASELINE configs[5] volume (1 x 1 x 160^3), loss alone."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd.synthetic import vessel_fields as cldice_inputs
from pytorchdeeplearing_amd.lossescldice import Binary_Soft_cldice_loss

SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 1, 160, 160, 160), (4, 1, 96, 96, 96), (16, 1, 512, 512)]
for shape in SHAPES:
    pred, target = cldice_inputs(shape, 3)
    pred, target = pred.cuda().requires_grad_(True), target.cuda()
    f = Binary_Soft_cldice_loss()
    for _ in range(2):
        pred.grad = None
        f(pred, target).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        pred.grad = None
        loss = f(pred, target)
        loss.backward()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    vox = pred.numel()
    # algorithmic traffic: 2 skeletons x 10 iterations x (read x, write e, read x+e, write x') + backward 10 x ~7 passes, fp32
    gb = vox * 4 * (2 * 10 * 5 + 10 * 9) / 1e9
    print(json.dumps({"shape": list(shape), "loss": float(loss.detach()), "ms_fwd_bwd": round(ms, 3), "approx_GB": round(gb, 2),
                      "approx_GBps": round(gb / ms * 1e3, 1)}))
