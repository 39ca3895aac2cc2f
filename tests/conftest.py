import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_emu_lib = None


def emu_library():
    """Host-side wave64 checker build of the product's HIP sources (tests/emu) - CPU tests only.  The product package has no hook for it: this function
    REPLACES `_capi.lib_for` / `_capi.host_library` in the test process so that CPU tensors reach the checker library (GPU tensors keep reaching the
    product library)."""
    global _emu_lib
    if _emu_lib is None:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        from pytorchdeeplearing_amd import _capi
        if hasattr(_capi, "product_lib_for"):          # this file imported twice (`conftest` and `tests.conftest`): the routing is already in place
            _emu_lib = _capi.host_library()
            return _emu_lib
        _emu_lib = _capi.SegLib(build_emu.build())
        product_lib_for = _capi.lib_for

        def lib_for(device):
            import torch
            return _emu_lib if torch.device(device).type == "cpu" else product_lib_for(device)

        _capi.product_lib_for = product_lib_for        # (tests of the "CPU tensors raise" behaviour call the original)
        _capi.lib_for = lib_for
        _capi.host_library = lambda: _emu_lib
    return _emu_lib


def emu_active():
    return _emu_lib is not None


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
