"""SEG_FORK_FLAG=1 on hardware (VERDICT r03 task 6): the completion-flag forks (gn_bwd_apply publishes a per-unit sequence number, a one-wave
kernel on the weight-gradient stream spins on it) against the event forks.
  1. equality: one forward + backward on the same weights / volume / dropout masks through a flag engine and an event engine: gradients to atomics noise;
  2. stress: K train steps with the flags on (fresh dropout every step); every 500 steps the same equality check from the stressed engine's weights;
     a spin-wait that timed out (bounded spin, ~0.3 s) would show as a wrong gradient or as a step that takes > 100 ms.
Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pytorchdeeplearing_amd import SegEngine, _capi, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda:0")
N, S = 4, 96
x, y = synthetic.synthetic_batch(N, (S, S, S), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)


def make(flag):
    os.environ["SEG_FORK_FLAG"] = flag
    e = SegEngine("vnet", 3, 1, 1, dtype="f16", device=dev)
    synthetic.init_engine(e, seed=0)
    return e


ef, ee = make("1"), make("0")


def equal_check(tag):
    ee.load_state_dict(ef.state_dict())
    res = []
    for e in (ef, ee):
        torch.manual_seed(0)
        logits, probs = e.forward(x, _capi.MASKS_EVAL, None)
        dl = e.loss_backward(logits, y, "BinaryDiceLoss")
        e.backward(dl)
        torch.cuda.synchronize()
        res.append({k: v.clone() for k, v in e.grad_dict().items()})
    worst = 0.0
    for k in res[0]:
        d = float((res[0][k].double() - res[1][k].double()).norm()) / (float(res[1][k].double().norm()) + 1e-30)
        worst = max(worst, d)
    return {"at": tag, "worst_rel_grad_diff": worst, "event_forks": [ef.lib.seg_plan_count(ef.h, 2), ee.lib.seg_plan_count(ee.h, 2)],
            "flag_waits": [ef.lib.seg_plan_count(ef.h, 3), ee.lib.seg_plan_count(ee.h, 3)]}


checks = [equal_check(0)]
slow, t_all = 0, time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    out3 = ef.train_step(x, y, "BinaryDiceLoss", lr=1e-4)
    if i % 50 == 49:
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) > 0.1:
            slow += 1
    if i % 500 == 499:
        checks.append(equal_check(i + 1))
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print(json.dumps({"steps": steps, "seconds": round(dt, 1), "final_loss": float(out3[0]), "slow_sync_points": slow, "checks": checks,
                  "max_worst": max(c["worst_rel_grad_diff"] for c in checks)}))
