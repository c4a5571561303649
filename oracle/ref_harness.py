"""Stub-import harness for the *reference* scoring code (TEST INFRASTRUCTURE ONLY).

Runs ONLY in the build container, where /root/reference exists.  It lets
``cald_train.get_uncertainty`` / ``cls_kldiv`` and ``cald.cald_helper`` be
imported on a CPU-only torch without torchvision, so that golden vectors for
the scoring half of the hot path can be
captured from the reference itself (see ``oracle/make_golden.py``).

Nothing here travels to the GPU box in executable form: only the fixtures it
produces (``tests/golden/*.npz``) do.  No reference source is copied; the
reference modules are imported from where they lie.
"""
import importlib
import sys
import types
from argparse import Namespace
from unittest import mock

REF_ROOT = "/root/reference"

_MISSING = [
    "torchvision", "torchvision.models", "torchvision.models.detection",
    "torchvision.models.detection.mask_rcnn", "torchvision.models.detection.faster_rcnn",
    "torchvision.models.detection.retinanet", "torchvision.models.detection.generalized_rcnn",
    "torchvision.models.detection.backbone_utils", "torchvision.models.detection.rpn",
    "torchvision.models.detection.roi_heads", "torchvision.models.detection.transform",
    "torchvision.models.detection.anchor_utils", "torchvision.models.detection.image_list",
    "torchvision.models.detection._utils", "torchvision.models._utils", "torchvision.models.utils",
    "torchvision.models.resnet", "torchvision.models.mobilenet",
    "torchvision.ops", "torchvision.ops.feature_pyramid_network", "torchvision.ops.misc",
    "torchvision.ops.boxes", "torchvision.ops.focal_loss",
    "torchvision.transforms", "torchvision.transforms.functional",
    "torchvision.datasets", "torchvision.datasets.voc", "torchvision.datasets.coco",
    "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval",
    "terminaltables", "cv2", "mmcv", "mmcv.utils", "torch._six",
]


def _to_tensor(pic):
    """3-line restatement of torchvision's to_tensor for PIL RGB / uint8 HWC input
    (uint8 HWC -> float32 CHW, true division by 255)."""
    import numpy as np
    import torch
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def _to_pil_image(t):
    import numpy as np
    from PIL import Image
    arr = (t.mul(255).byte().permute(1, 2, 0).numpy())
    return Image.fromarray(np.ascontiguousarray(arr))


def _adjust(kind):
    """torchvision.transforms.functional.adjust_{brightness,contrast,saturation} for PIL input are one-liners over
    PIL.ImageEnhance (functional_pil.py); restated here because torchvision is not installed."""
    def f(img, factor):
        from PIL import ImageEnhance
        return getattr(ImageEnhance, kind)(img).enhance(factor)
    return f


def install_stubs():
    import torch

    class _Base(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mods = {}
    for name in _MISSING:
        if name in sys.modules and not isinstance(sys.modules[name], types.ModuleType):
            continue
        m = types.ModuleType(name)
        m.__path__ = []
        m.__all__ = []
        mods[name] = m

    def _mk_getattr(modname):
        def __getattr__(attr):
            full = modname + "." + attr
            if full in mods:
                return mods[full]
            if attr.startswith("__"):
                raise AttributeError(attr)
            return mock.MagicMock(name=full)
        return __getattr__

    for name, m in mods.items():
        m.__getattr__ = _mk_getattr(name)
        sys.modules[name] = m

    # real (empty) base classes where the reference subclasses torchvision types
    mods["torchvision.models.detection.faster_rcnn"].FasterRCNN = _Base
    mods["torchvision.models.detection.faster_rcnn"].TwoMLPHead = _Base
    mods["torchvision.models.detection.faster_rcnn"].FastRCNNPredictor = _Base
    mods["torchvision.models.detection.roi_heads"].RoIHeads = _Base
    mods["torchvision.models.detection.transform"].GeneralizedRCNNTransform = _Base
    mods["torchvision.models.detection.generalized_rcnn"].GeneralizedRCNN = _Base
    mods["torchvision.models.detection.rpn"].RegionProposalNetwork = _Base
    mods["torchvision.models.detection.rpn"].RPNHead = _Base
    mods["torchvision.models.detection.rpn"].AnchorGenerator = _Base
    mods["torchvision.models.detection.anchor_utils"].AnchorGenerator = _Base
    mods["torchvision.ops"].MultiScaleRoIAlign = _Base
    mods["torchvision.datasets"].VOCDetection = object
    mods["torchvision.datasets"].CocoDetection = object
    mods["torchvision.datasets.voc"].VOCDetection = object
    mods["torchvision.datasets.coco"].CocoDetection = object
    mods["torchvision.transforms.functional"].to_tensor = _to_tensor
    mods["torchvision.transforms.functional"].to_pil_image = _to_pil_image
    mods["torchvision.transforms.functional"].adjust_brightness = _adjust("Brightness")
    mods["torchvision.transforms.functional"].adjust_contrast = _adjust("Contrast")
    mods["torchvision.transforms.functional"].adjust_saturation = _adjust("Color")
    mods["torch._six"].string_classes = (str,)
    mods["torch._six"].container_abcs = importlib.import_module("collections.abc")
    mods["torch._six"].int_classes = (int,)

    # the reference calls .cuda() unconditionally; on this CPU-only torch make it a no-op
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None


def load_reference(bp=1.3, uniform=False):
    """Returns (cald_train module, cald_helper module) imported from /root/reference."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    cald_train = importlib.import_module("cald_train")
    cald_helper = importlib.import_module("cald.cald_helper")
    cald_train.args = Namespace(bp=bp, uniform=uniform)
    return cald_train, cald_helper


def load_baselines():
    """lt_c_train / ls_c_train.  ls_c_train.py:52 imports `cal4od.cal4od_helper`, a module
    that does not exist in the reference tree (the script is broken as shipped); the
    harness aliases it to cald.cald_helper, which defines the same helper names (GaussianNoise, ...)."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    helper = importlib.import_module("cald.cald_helper")
    pkg = types.ModuleType("cal4od"); pkg.__path__ = []
    sys.modules["cal4od"] = pkg
    sys.modules["cal4od.cal4od_helper"] = helper
    return importlib.import_module("lt_c_train"), importlib.import_module("ls_c_train")
