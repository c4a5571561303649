"""Drop-in detector objects behind the reference's factory signatures.

``fasterrcnn_resnet50_fpn_feature(pretrained, progress, num_classes, pretrained_backbone, **kwargs)``
mirrors detection/frcnn_la.py:278-289 (call sites cald_train.py:340-347): the returned object exposes
``.eval() / .to() / .load_state_dict() / __call__(list[Tensor[3,H,W] float 0..1])`` and returns the
result dicts of detection/frcnn_la.py:131-141 (``boxes, labels, scores, props, prob_max,
scores_cls``; ``features`` is omitted, detection/engine.py:109-111 checks for its presence).
All arithmetic runs in libcaldhip.so on the MI355X; there is no CPU path.  Training is out of
scope: ``.train()`` raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi

_CTX = {}


def get_ctx(device_index=None):
    """One library context per device, bound to torch's current stream on that device."""
    if not torch.cuda.is_available():
        raise RuntimeError("cald_amd needs a visible MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _CTX:
        h = C.c_void_p()
        stream = torch.cuda.current_stream(device_index).cuda_stream
        _ffi.check(_ffi.lib().cald_ctx_create(device_index, C.c_void_p(stream), C.byref(h)))
        _CTX[device_index] = h
    return _CTX[device_index]


_SIDE_CTX = {}


def get_side_ctx(device_index, stream):
    """A second library context of the same device bound to `stream` (a torch.cuda.Stream): lets independent kernels (the weight
    gradients of the training backward) run beside the main stream's.  The caller orders the two streams with events."""
    key = (device_index, stream.cuda_stream)
    if key not in _SIDE_CTX:
        h = C.c_void_p()
        _ffi.check(_ffi.lib().cald_ctx_create(device_index, C.c_void_p(stream.cuda_stream), C.byref(h)))
        _SIDE_CTX[key] = h
    return _SIDE_CTX[key]


_U8_TABLE = {}


def _u8_over_255(device):
    """float32 table k / 255 (numpy: correctly rounded division, the same values as the kernel's `(float)u8 / 255.0f`)."""
    key = str(device)
    if key not in _U8_TABLE:
        _U8_TABLE[key] = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(device)
    return _U8_TABLE[key]


class HipDetector:
    def __init__(self, num_classes, depth=50, min_size=800, max_size=1333, box_score_thresh=0.05, box_nms_thresh=0.5,
                 box_detections_per_img=100, rpn_pre_nms_top_n_test=1000, rpn_post_nms_top_n_test=1000,
                 rpn_nms_thresh=0.7, arch=0, precision="fp32", **unused):
        """precision: "fp32" (exact, bit-identical to the oracle; default) or "f16x3" (split-fp16 MFMA path, 2x faster, fp32-grade
        but not bit-identical to fp32 and not reproducible on a CPU) -- include/cald_hip.h, DESIGN.md section 6."""
        self.arch = arch
        self.cfg = _ffi.ModelCfg(self.arch, depth, num_classes, int(min_size), int(max_size), box_score_thresh,
                                 box_nms_thresh, box_detections_per_img, rpn_pre_nms_top_n_test, rpn_post_nms_top_n_test,
                                 rpn_nms_thresh, _ffi.PRECISION[precision])
        self.num_classes = num_classes
        self.training = False
        self._state = None
        self._handle = None
        self._device = None
        self._depth = depth
        self._trainer = None          # cald_amd.train.FasterRCNNTrainer while the weights are being trained
        self._train_fn = None

    # ---- nn.Module-like surface used by cald_train.py ----
    def eval(self):
        if self.training and self._trainer is not None:
            self._sync_from_trainer()
        self.training = False
        return self

    def train(self, mode=True):
        """task_model.train() (cald_train.py:41): the Faster R-CNN training step runs on the HIP operators of cald_amd/train.py
        (SURVEY 8f rank 4); ``model(images, targets)`` then returns the loss dict (Faster R-CNN: four losses, frcnn_la.py; RetinaNet:
        'classification' / 'bbox_regression', retinanet_cal.py:50-55)."""
        if not mode:
            return self.eval()
        if self.cfg.precision != _ffi.PRECISION["fp32"]:
            raise NotImplementedError("the training step computes in fp32; build the detector with precision='fp32' to train it "
                                      "(the f16x3 mode is inference-only)")
        self._ensure_trainer()
        self.training = True
        return self

    def _ensure_trainer(self):
        if self._trainer is None:
            if self._state is None:
                raise RuntimeError("no weights: call load_state_dict() first (cald_train.py:356)")
            from . import train as _train
            dev = "cuda:%d" % (self._device if self._device is not None else torch.cuda.current_device())
            if self.arch == 0:
                self._trainer = _train.FasterRCNNTrainer(self._state, self.num_classes, depth=self._depth, min_size=self.cfg.min_size,
                                                         max_size=self.cfg.max_size, device=dev, rpn_nms_thresh=self.cfg.rpn_nms_thresh)
            else:
                self._trainer = _train.RetinaNetTrainer(self._state, self.num_classes, depth=self._depth, min_size=self.cfg.min_size,
                                                        max_size=self.cfg.max_size, device=dev)
            self._train_fn = _train.TrainableDetector(self._trainer)
        return self._trainer

    def _sync_from_trainer(self):
        """The trained parameters become the inference engine's weights (the native model is rebuilt on the next forward)."""
        self._state = {k: v.detach().cpu().numpy() for k, v in self._trainer.state_dict().items()}
        self._destroy()

    def parameters(self):
        """task_model.parameters() (cald_train.py:396): the trainable tensors (``requires_grad`` True), torch Parameters."""
        return self._ensure_trainer().parameters()

    def named_parameters(self):
        return self._ensure_trainer().named_parameters()

    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("cald_amd detectors live on the MI355X only")
        self._device = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else "cuda:%d" % device)

    def state_dict(self):
        """{name: CPU float32 tensor} in torchvision's key layout, like ``nn.Module.state_dict()``: ``torch.save({'model':
        task_model.state_dict()})`` (cald_train.py:421) and ``stock_torch_model.load_state_dict(hip_model.state_dict())`` both work."""
        from collections import OrderedDict
        if self._trainer is not None and self.training:
            return OrderedDict((k, v.detach().cpu()) for k, v in self._trainer.state_dict().items())
        return OrderedDict((k, torch.from_numpy(np.array(v, dtype=np.float32))) for k, v in (self._state or {}).items())

    def load_state_dict(self, sd, strict=True):
        self._state = {k: (v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, np.float32))
                       for k, v in sd.items() if not k.endswith("num_batches_tracked")}
        self._destroy()
        self._trainer = self._train_fn = None
        if self.training:          # train mode: a trainer on the new weights (parameters() are NEW tensors -- build the optimizer after
            self._ensure_trainer()  # loading, as cald_train.py:356 / :396 does)
        return self

    # ---- native handle ----
    def _destroy(self):
        if self._handle is not None:
            _ffi.lib().cald_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def handle(self):
        if self._handle is None:
            if self._state is None:
                raise RuntimeError("no weights: call load_state_dict() first (cald_train.py:356)")
            L = _ffi.lib()
            ctx = self._ctx if getattr(self, "_ctx", None) is not None else get_ctx(self._device)      # _ctx: a context of its own (second stream)
            h = C.c_void_p()
            _ffi.check(L.cald_model_create(ctx, C.byref(self.cfg), C.byref(h)))
            for k, v in self._state.items():
                a = np.ascontiguousarray(v, dtype=np.float32)
                if a.ndim == 0:
                    continue
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _ffi.check(L.cald_model_load_tensor(h, k.encode(), _ffi.ptr(a), shape, a.ndim))
            _ffi.check(L.cald_model_finalize(h))
            self._handle = h
        return self._handle

    # ---- inference ----
    def forward_views(self, views):
        """views: list of (uint8 HWC cuda tensor, flip, rects[, noise]) -- noise: optional float32 CHW cuda tensor added to
        image/255 (cald_view.noise_dev).  Returns list of result dicts (device tensors)."""
        L = _ffi.lib()
        h = self.handle()
        n = len(views)
        dev = views[0][0].device
        Cn = self.num_classes
        cap = self.cfg.detections_per_img * (Cn if self.arch == 1 else 1)
        out = dict(boxes=torch.empty((n, cap, 4), device=dev), scores=torch.empty((n, cap), device=dev),
                   labels=torch.empty((n, cap), dtype=torch.int64, device=dev), props=torch.empty((n, cap, 4), device=dev),
                   prob_max=torch.empty((n, cap), device=dev), scores_cls=torch.empty((n, cap, Cn), device=dev),
                   count=torch.zeros((n,), dtype=torch.int32, device=dev))
        arr = (_ffi.View * n)()
        for i, view in enumerate(views):
            img, flip, rects = view[:3]
            noise = view[3] if len(view) > 3 else None
            assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous() and img.shape[2] == 3
            arr[i].image_dev = img.data_ptr(); arr[i].H = img.shape[0]; arr[i].W = img.shape[1]
            if noise is not None:
                assert noise.dtype == torch.float32 and noise.is_cuda and noise.is_contiguous() and tuple(noise.shape) == (3, img.shape[0], img.shape[1])
                arr[i].noise_dev = noise.data_ptr()
            arr[i].flip = int(bool(flip)); arr[i].nrect = 0 if rects is None else len(rects)
            if rects is not None:
                for j, r in enumerate(np.asarray(rects, np.int32).reshape(-1)):
                    arr[i].rects[j] = int(r)
        d = _ffi.Dets(out["boxes"].data_ptr(), out["scores"].data_ptr(), out["labels"].data_ptr(), out["props"].data_ptr(),
                      out["prob_max"].data_ptr(), out["scores_cls"].data_ptr(), out["count"].data_ptr(), cap)
        _ffi.check(L.cald_forward(h, n, arr, C.byref(d)))
        counts = out["count"].cpu().tolist()
        res = []
        for i, k in enumerate(counts):
            keys = ("boxes", "labels", "scores", "props", "prob_max", "scores_cls") if self.arch == 0 else \
                   ("boxes", "scores", "labels", "scores_cls", "prob_max")
            res.append({key: out[key][i, :k] for key in keys})
        return res

    def __call__(self, images, targets=None):
        """model(list[Tensor[3,H,W] float32]) -> list[dict] (detection/frcnn_la.py:237-275).
        A float tensor is split EXACTLY into the nearest uint8 grid image g and the float32 remainder x - g/255 (Sterbenz:
        the subtraction is exact and fl(g/255 + (x - g/255)) == x), so to_tensor outputs (remainder all zero, no extra
        traffic) and arbitrary floats -- e.g. a caller-made GaussianNoise image -- both reach the kernels bit for bit."""
        if self.training:
            if targets is None:
                raise ValueError("In training mode, targets should be passed")      # GeneralizedRCNN.forward (frcnn_la.py:247-248)
            return self._train_fn(images, targets)
        views = []
        for img in images:
            if img.dtype == torch.uint8:
                u8 = img if img.shape[-1] == 3 else img.permute(1, 2, 0)
                views.append((u8.contiguous().cuda(), False, None))
                continue
            x = img.detach().to(torch.float32).cuda().contiguous()
            if not bool(torch.isfinite(x).all()):
                raise ValueError("non-finite pixel values in the input image")
            g = (x * 255.0).round().clamp(0, 255).to(torch.uint8)
            base = _u8_over_255(x.device)[g.long()]            # the kernel's (float)u8 / 255.0f, IEEE division
            rem = x - base
            if not bool(torch.equal(base + rem, x)):             # cannot happen for finite inputs; fail loudly rather than drift
                raise RuntimeError("float input is not representable as uint8 grid + float32 remainder")
            noise = rem.contiguous() if bool((rem != 0).any()) else None
            views.append((g.permute(1, 2, 0).contiguous(), False, None, noise))
        return self.forward_views(views)

    def set_rpn_prune(self, on):
        """Certified RPN pruning of the exact sweeps (include/cald_hip.h cald_model_set_rpn_prune): returns the previous state."""
        was = C.c_int(0)
        _ffi.check(_ffi.lib().cald_model_set_rpn_prune(self.handle(), int(bool(on)), C.byref(was)))
        return bool(was.value)

    def set_rpn_prune_capture(self, on):
        """Test hook (cald_model_set_rpn_prune_capture): forward_views then takes the pruned RPN path and keeps the look-ahead's maps."""
        _ffi.check(_ffi.lib().cald_model_set_rpn_prune_capture(self.handle(), int(bool(on))))

    def rpn_prune_bound(self):
        """(c1[3], c0[3]) of the pruning's per-anchor bound B_a(p) = c1[a] * |patch(p)|_2 + c0[a]."""
        import numpy as np
        c1 = np.zeros(3, np.float32); c0 = np.zeros(3, np.float32)
        _ffi.check(_ffi.lib().cald_model_rpn_prune_bound(self.handle(), _ffi.ptr(c1), _ffi.ptr(c0)))
        return c1, c0

    def debug_tensor(self, name, view=0):
        shape = (C.c_int64 * 3)()
        cap = 1 << 26
        buf = np.empty(cap, np.float32)
        _ffi.check(_ffi.lib().cald_debug_tensor(self.handle(), name.encode(), view, _ffi.ptr(buf), cap, shape))
        n = shape[0] * shape[1] * shape[2]
        return buf[:n].reshape(shape[0], shape[1], shape[2]).copy()


def fasterrcnn_resnet50_fpn_feature(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True, **kwargs):
    """detection/frcnn_la.py:278-289.  No network here: `pretrained*` cannot download anything; weights arrive
    through load_state_dict() exactly as in cald_train.py:349-356."""
    return HipDetector(num_classes, depth=50, **kwargs)


def fasterrcnn_resnet101_fpn_feature(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True, **kwargs):
    """BASELINE.json config 5 (ResNet-101): same factory with torchvision's 'resnet101' body."""
    return HipDetector(num_classes, depth=101, **kwargs)


def retinanet_resnet50_fpn_cal(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True,
                               score_thresh=0.05, nms_thresh=0.5, detections_per_img=300, **kwargs):
    """detection/retinanet_cal.py:584-625 (constructor defaults :322-333).  Result dict keys of :479-485."""
    return HipDetector(num_classes, depth=50, box_score_thresh=score_thresh, box_nms_thresh=nms_thresh,
                       box_detections_per_img=detections_per_img, arch=1, **kwargs)


def from_torch_module(task_model):
    """Builds the HIP detector that mirrors a reference torch model (detection/frcnn_la.py FRCNN_Feature or
    detection/retinanet_cal.py RetinaNet): constructor arguments are read off the module's attributes
    (transform.min_size / max_size, roi_heads.*, rpn.*), the weights off ``state_dict()`` -- so
    ``get_uncertainty(task_model, ...)`` accepts the model cald_train.py already has (cald_train.py:436)."""
    sd = task_model.state_dict()
    tr = getattr(task_model, "transform", None)
    min_size = getattr(tr, "min_size", 800)
    min_size = int(min_size[0] if isinstance(min_size, (tuple, list)) else min_size)
    max_size = int(getattr(tr, "max_size", 1333))
    depth = 101 if any(k.startswith("backbone.body.layer3.22.") for k in sd) else 50
    if any(k.startswith("head.classification_head.") for k in sd):
        ncls = int(sd["head.classification_head.cls_logits.weight"].shape[0]) // 9
        m = HipDetector(ncls, depth=depth, min_size=min_size, max_size=max_size, arch=1,
                        box_score_thresh=float(getattr(task_model, "score_thresh", 0.05)),
                        box_nms_thresh=float(getattr(task_model, "nms_thresh", 0.5)),
                        box_detections_per_img=int(getattr(task_model, "detections_per_img", 300)))
    else:
        ncls = int(sd["roi_heads.box_predictor.cls_score.weight"].shape[0])
        rh, rpn = getattr(task_model, "roi_heads", None), getattr(task_model, "rpn", None)
        pre = getattr(rpn, "_pre_nms_top_n", {"testing": 1000}); post = getattr(rpn, "_post_nms_top_n", {"testing": 1000})
        m = HipDetector(ncls, depth=depth, min_size=min_size, max_size=max_size,
                        box_score_thresh=float(getattr(rh, "score_thresh", 0.05)), box_nms_thresh=float(getattr(rh, "nms_thresh", 0.5)),
                        box_detections_per_img=int(getattr(rh, "detections_per_img", 100)),
                        rpn_pre_nms_top_n_test=int(pre["testing"]), rpn_post_nms_top_n_test=int(post["testing"]),
                        rpn_nms_thresh=float(getattr(rpn, "nms_thresh", 0.7)))
    m.to("cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else "cuda")
    m.load_state_dict(sd)
    return m.eval()
