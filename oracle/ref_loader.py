"""TEST INFRASTRUCTURE ONLY.  Import the real reference (read-only, /root/reference)
under alias package names so it never clashes with this repo's own packages.

Follows SURVEY.md Appendix B.  Shims (no reference file is edited):
  * networks/VNet3d.py:127 reads ``self.feature`` (typo for ``self.features``) -> class attr.
  * model/__init__.py pulls cv2/SimpleITK/torchsummary/tensorboard -> never executed; the
    leaf modules losses.py / metric.py / lossescldice.py are loaded by path.
  * model/metric.py:8 imports skimage -> stub module.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("SEG_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "networks", "VNet3d.py"))


def _load_pkg(alias, path):
    spec = importlib.util.spec_from_file_location(
        alias, os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load():
    """Returns (ref_networks, ref_losses, ref_metric)."""
    if "v" in _cache:
        return _cache["v"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    sys.dont_write_bytecode = True
    nets = _load_pkg("ref_networks", os.path.join(REF, "networks"))
    sys.modules["ref_networks.VNet3d"].VNet3d.feature = 16
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.metrics")
    skm.structural_similarity = None
    sys.modules.setdefault("skimage", sk)
    sys.modules.setdefault("skimage.metrics", skm)
    pkg = types.ModuleType("ref_model")
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules["ref_model"] = pkg

    def leaf(name):
        spec = importlib.util.spec_from_file_location(
            "ref_model." + name, os.path.join(REF, "model", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ref_model." + name] = mod
        spec.loader.exec_module(mod)
        return mod

    leaf("lovasz")
    leaf("lossesSSIM")
    losses = leaf("losses")
    metric = leaf("metric")
    _cache["v"] = (nets, losses, metric)
    return _cache["v"]
