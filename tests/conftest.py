import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("GPU_STREAMOPS_CP_WAIT", "1")      # SEG_FORK=flag (test_flag_forks_equal_event_forks): hipStreamWaitValue32 as a command-processor wait; read when the HIP runtime starts

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_emu_lib = None


def emu_library():
    """Host-side wave64 checker build of the UNMODIFIED HIP sources (tests/emu) — CPU tests only."""
    global _emu_lib
    if _emu_lib is None:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        from pytorchdeeplearing_amd import _capi
        # SEG_TEST_EXPERIMENTS=1: the -DSEG_EXPERIMENTS twin of the checker (the measured-slower paths of rounds 2-4 and every tuning knob);
        # the default is the product's source list and flags
        _emu_lib = _capi.SegLib(build_emu.build(experiments=EXPERIMENTS))
        _capi.inject_library(_emu_lib)
    return _emu_lib


EXPERIMENTS = bool(os.environ.get("SEG_TEST_EXPERIMENTS"))
PERSISTENT_CFGS = (18, 19, 28, 29, 40, 58, 51, 52, 53, 59)      # conv3p / conv3p16 / conv3q / conv3q16 tilings: experiments build only


def needs_experiments(dev):
    """Tests of paths that are not in the product library (wgrad3x, GroupNorm in the consumer conv, persistent halo convs, flag forks, sub-batched levels,
    two weight-gradient streams): they run against the experiments build only - SEG_TEST_EXPERIMENTS=1 on the host checker, and on the GPU additionally
    SEGENGINE_LIB=pytorchdeeplearing_amd/lib/libsegengine_exp.so (python -m pytorchdeeplearing_amd.build --experiments)."""
    from pytorchdeeplearing_amd import _capi
    if "+experiments" not in _capi.lib_for(dev).build_info():
        pytest.skip("experiments build only (SEG_TEST_EXPERIMENTS=1; on the GPU also SEGENGINE_LIB=.../libsegengine_exp.so)")


@pytest.fixture(params=[pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    """'emu': kernel sources run on the host-side execution-model checker (logic / indexing check);
    'gpu': the real gfx950 library on cuda:0 (the parity tests proper)."""
    import torch
    if request.param == "emu":
        emu_library()
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from pytorchdeeplearing_amd import _capi
    _capi.product_library()      # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def checker_slow(dev, why="minutes on the host checker"):
    """Cases whose host-checker run takes a minute or more are left to the GPU run (`-m gpu` executes the same test on the
    device) so that `-m "not gpu"` stays a few-minute suite; SEG_TEST_FULL=1 runs them on the checker too."""
    if getattr(dev, "type", str(dev)) == "cpu" and not os.environ.get("SEG_TEST_FULL"):
        pytest.skip("%s; covered by the -m gpu run of this test (SEG_TEST_FULL=1 runs it here)" % why)
