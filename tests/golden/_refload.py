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


# ---- package-style loads (files that use relative imports) -------------------------------------
# A synthetic package tree "refman" mirrors the reference's package names; every node is an EMPTY
# module with a __path__ (so `from .x import y` resolves through sys.modules), the files on the hot
# path are executed from where they lie under /root/reference, and the training-only siblings they
# import at module level (dataset, train_utils, training.train) are inert stubs holding None names.

def _pkg(name, relpath):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = [str(REF / relpath)] if relpath is not None else []
    m.__package__ = name
    sys.modules[name] = m
    return m


class _Inert:
    """Base class / callable stand-in for third-party names that are only touched at import time."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self


def _install_pkg_stubs():
    install_stubs()
    if "gdown" not in sys.modules:
        _stub("gdown", download=None)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tr = _stub("torchvision.transforms", Compose=_Inert, ToTensor=_Inert, Normalize=_Inert)
        tv.transforms = tr
    if "albumentations" not in sys.modules:
        _stub("albumentations", ImageOnlyTransform=_Inert, Compose=_Inert, Normalize=_Inert)
        _stub("albumentations.pytorch", ToTensorV2=_Inert)
    if "tqdm" not in sys.modules:  # present in this image; kept for completeness
        _stub("tqdm", tqdm=lambda x, **k: x)


def _load_into(pkgname, modname, relpath):
    full = f"{pkgname}.{modname}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, REF / relpath)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = pkgname
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    setattr(sys.modules[pkgname], modname, mod)
    return mod


def ref_east_infer():
    """detectors/_east/infer.py (class EAST: the box tail :134-233 is plain NumPy).  dataset.py and
    train_utils.py (training, need skimage / tensorboard / torch_optimizer) are stubs."""
    _install_pkg_stubs()
    _pkg("refman", None)
    _pkg("refman.detectors", None)
    east = _pkg("refman.detectors._east", None)
    _load_into("refman.detectors", "_types", "detectors/_types.py")
    _stub("refman.detectors._east.dataset", EASTDataset=None)
    _stub("refman.detectors._east.train_utils", _run_training=None)
    for f in ("lanms", "utils", "east"):
        _load_into("refman.detectors._east", f, f"detectors/_east/{f}.py")
    mod = _load_into("refman.detectors._east", "infer", "detectors/_east/infer.py")
    east.EAST = mod.EAST
    return mod


def ref_pipeline():
    """_pipeline.py with the REAL utils.py functions behind `.detectors`; the default plugin classes
    (EAST, TRBA — only touched by Pipeline() without arguments) are None."""
    infer = ref_east_infer()
    utils = sys.modules["refman.detectors._east.utils"]
    det = sys.modules["refman.detectors"]
    det.EAST = infer.EAST
    for n in ("visualize_page", "read_image", "sort_boxes_reading_order", "sort_boxes_reading_order_with_resolutions"):
        setattr(det, n, getattr(utils, n))
    _stub("refman.recognizers", TRBA=None)
    return _load_into("refman", "_pipeline", "_pipeline.py")


def ref_trba_wrapper():
    """recognizers/_trba/__init__.py (class TRBA).  training/train.py is a stub (Config / run_training = None);
    data/transforms.py loads behind inert albumentations / cv2 stubs (load_charset and decode_tokens are pure Python)."""
    _install_pkg_stubs()
    base = "refman_trba"  # its own root: `refman.recognizers` stays ref_pipeline's stub
    _pkg(base, "recognizers/_trba")
    for sub in ("model", "data", "training"):
        _pkg(f"{base}.{sub}", None)
    _load_into(f"{base}.model", "seresnet31", "recognizers/_trba/model/seresnet31.py")
    _load_into(f"{base}.model", "model", "recognizers/_trba/model/model.py")
    _load_into(f"{base}.data", "transforms", "recognizers/_trba/data/transforms.py")
    _load_into(f"{base}.training", "utils", "recognizers/_trba/training/utils.py")
    _stub(f"{base}.training.train", Config=None, run_training=None)
    spec = importlib.util.spec_from_file_location(base + ".__ref_init__", REF / "recognizers/_trba/__init__.py")
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = base
    sys.modules[base + ".__ref_init__"] = mod
    spec.loader.exec_module(mod)
    return mod
