"""Synthetic workload of the benchmark and the tuning tools (no dataset or checkpoint ships with the reference): seeded inputs
of the shapes the datasets deliver (model/dataset.py:107-114: z-scored float32 image, int64 label) and random-init weights with
the distribution of `networks.initialize_weights` (networks/__init__.py:11-26).  Product-side on purpose: `bench.py` must not
touch `oracle/` outside its cpu_baseline leg."""
import torch


def synthetic_batch(n, spatial, in_ch=1, numclass=1, seed=1234):
    """x ~ N(0,1) float32 (n, in_ch, *spatial); binary label with ~20 % foreground, or uniform class ids."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, in_ch) + tuple(spatial), generator=g)
    if numclass == 1:
        y = (torch.rand((n,) + tuple(spatial), generator=g) > 0.8).long()
    else:
        y = torch.randint(0, numclass, (n,) + tuple(spatial), generator=g)
    return x, y


def init_engine(engine, seed=0):
    """kaiming_normal_(nonlinearity='relu') for conv / conv-transpose weights (fan_in = size(1) x kernel volume, as torch
    computes it for both), biases 0, GroupNorm gamma 1 / beta 0 — written straight into the engine's flat fp32 buffer."""
    if hasattr(engine, "engines"):                    # lanes.LaneEngine: the lanes share lane 0's parameters
        engine.load_state_dict(init_engine(engine.engines[0], seed).state_dict())
        return engine
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, (shape, _off) in engine.table.items():
            view = engine.param_view(name)
            if len(shape) > 1:
                fan_in = shape[1]
                for k in shape[2:]:
                    fan_in *= k
                view.copy_((torch.randn(shape, generator=g, dtype=torch.float64) * (2.0 / fan_in) ** 0.5).float())
            elif name.endswith("weight"):
                view.fill_(1.0)
            else:
                view.zero_()
    engine.packed = False
    return engine


def vessel_fields(shape, seed, numclass=0):
    """vessel-like probabilities for the soft-clDice tools: low-pass noise through a sigmoid / soft-max; the target is
    another such field thresholded (binary) or its arg-max (multi-class).  shape = (N, C, [D,] H, W)."""
    g = torch.Generator().manual_seed(seed)
    nd = len(shape) - 2
    pool = torch.nn.functional.avg_pool3d if nd == 3 else torch.nn.functional.avg_pool2d

    def field(ch):
        t = torch.randn((shape[0], ch) + tuple(shape[2:]), generator=g)
        return pool(pool(t, 3, 1, 1), 3, 1, 1) * 6.0
    if numclass:
        return torch.softmax(field(numclass), 1), field(numclass).argmax(1)
    return torch.sigmoid(field(shape[1])), (field(shape[1]) > 0.3).float()
