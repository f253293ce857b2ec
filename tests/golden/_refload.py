"""Loader for individual reference source files (golden-vector generation ONLY).

Runs only in the build container where /root/reference exists; nothing under
tests/ that is collected by pytest imports this module.  The reference package
cannot be imported as a whole (cv2 / torchvision / numba / albumentations are
absent), so single files are loaded with importlib after registering INERT
stub modules for the missing third-party names.  The stubs contain no
arithmetic: numba.njit is the identity decorator, so lanms.py runs as plain
NumPy float64 — the same IEEE operations numba emits without fastmath.
"""
import importlib.util
import sys
import types
from pathlib import Path

REF = Path("/root/reference/src/manuscript")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Sub:
    """Dummy subscriptable / callable type object (numba signatures)."""

    def __getitem__(self, _):
        return self

    def __call__(self, *a, **k):
        return self


def _njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], (_Sub, str)) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


def install_stubs():
    if "numba" not in sys.modules:
        _stub("numba", njit=_njit, float64=_Sub(), int64=_Sub())
        _stub("numba.types", Tuple=_Sub())
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "shapely" not in sys.modules:
        _stub("shapely")
        _stub("shapely.geometry", Polygon=object)
    if "torchvision" not in sys.modules:
        _stub("torchvision")
        _stub("torchvision.ops", DropBlock2d=object)
        _stub(
            "torchvision.models",
            resnet50=None,
            ResNet50_Weights=None,
            resnet101=None,
            ResNet101_Weights=None,
        )
        _stub("torchvision.models.feature_extraction", create_feature_extractor=None)


def load_file(modname, relpath, package=None):
    install_stubs()
    spec = importlib.util.spec_from_file_location(modname, REF / relpath)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_lanms():
    return load_file("ref_lanms", "detectors/_east/lanms.py")


def ref_east_utils():
    return load_file("ref_east_utils", "detectors/_east/utils.py")


def ref_east_model():
    """DecoderBlock / FeatureMergingBranchResNet / OutputHead (plain torch).
    The torchvision backbone is NOT available (stubbed names are None)."""
    return load_file("ref_east_model", "detectors/_east/east.py")


def ref_trba_model():
    pkg = types.ModuleType("ref_trba_pkg")
    pkg.__path__ = [str(REF / "recognizers/_trba/model")]
    sys.modules["ref_trba_pkg"] = pkg
    load_file("ref_trba_pkg.seresnet31", "recognizers/_trba/model/seresnet31.py", "ref_trba_pkg")
    return load_file("ref_trba_pkg.model", "recognizers/_trba/model/model.py", "ref_trba_pkg")


def ref_types():
    return load_file("ref_types", "detectors/_types.py")
