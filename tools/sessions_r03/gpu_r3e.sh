cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
python - > $O/graph_dbg.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from pytorchdeeplearing_amd import SegEngine, synthetic
dev = torch.device('cuda:0')
e = SegEngine('vnet', 3, 1, 1, dtype='f16', device=dev)
synthetic.init_engine(e, seed=0)
x, y = synthetic.synthetic_batch(2, (32, 32, 32), 1, 1, seed=1)
x, y = x.to(dev), y.to(dev)
lg = torch.empty((2, 1, 32, 32, 32), device=dev); pr = torch.empty_like(lg)
for i in range(3):
    out = e.train_step(x, y, 'BinaryDiceLoss', logits=lg, probs=pr, launch='graph')
    torch.cuda.synchronize()
    print(i, float(out[0]), 'ready', e.lib.seg_train_graph_ready(e.h), 'err', getattr(e, 'graph_error', None))
PY
cat $O/graph_dbg.log
timeout 600 python -m pytest tests/test_wrappers.py -m gpu -x -q -k "reference_entry" 2>&1 | tail -40 > $O/entry.log; grep -E "Error|error|assert|passed|failed" $O/entry.log | head -20
