"""Why is soft-clDice slow for the 2nd+ tensor shape in a process?  Allocator counters and host/GPU time per shape."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorchdeeplearing_amd.synthetic import vessel_fields
from pytorchdeeplearing_amd.lossescldice import Binary_Soft_cldice_loss
for shape in [(1, 1, 160, 160, 160), (4, 1, 96, 96, 96), (1, 1, 160, 160, 160)]:
    pred, target = vessel_fields(shape, 3)
    pred, target = pred.cuda().requires_grad_(True), target.cuda()
    f = Binary_Soft_cldice_loss()
    for _ in range(2):
        pred.grad = None
        f(pred, target).backward()
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(5):
        pred.grad = None
        loss = f(pred, target)
        loss.backward()
    b.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print(json.dumps({"shape": list(shape), "host_enqueue_ms": round((t1 - t0) / 5 * 1e3, 2), "wall_ms": round((t2 - t0) / 5 * 1e3, 2),
                      "gpu_event_ms": round(a.elapsed_time(b) / 5, 2),
                      "device_allocs": s1["num_device_alloc"] - s0["num_device_alloc"], "device_frees": s1["num_device_free"] - s0["num_device_free"],
                      "alloc_retries": s1["num_alloc_retries"] - s0["num_alloc_retries"],
                      "reserved_MB": round(s1["reserved_bytes.all.current"] / 1e6), "allocated_MB": round(s1["allocated_bytes.all.current"] / 1e6)}))
