"""Training step of the Faster R-CNN detector on the MI355X.

What the reference does per iteration (cald_train.py:40-74 ``train_one_epoch``; detection/engine.py:19-61):

    loss_dict = task_model(images, targets); losses = sum(loss_dict.values())
    task_optimizer.zero_grad(); losses.backward(); task_optimizer.step()

with ``task_model`` = detection/frcnn_la.py ``FRCNN_Feature`` in train mode -- i.e. torchvision 0.8.2's GeneralizedRCNN training
forward (transform -> ResNet-FPN -> RPN losses -> sampled RoI-head losses) under torch autograd on cuDNN/cuBLAS.  Here the same
graph is strung from the ``cald_train_*`` device operators (cald_amd/csrc/train.hip): the forward runs on the inference MFMA
kernels with weights packed on the device every step, the backward is hand-written (data gradients on the same kernels with the
flipped filter, weight gradients on the split-K MFMA kernel, RoIAlign / FPN / ReLU / FrozenBatchNorm backward kernels), and
``SGD`` is one fused kernel per tensor.  torch supplies device memory, the stream, the Parameter / Optimizer / autograd-Function
*interfaces* (so ``losses.backward()`` and the reference's lr schedulers work unchanged) and the CPU random permutations of the
samplers; it performs none of the arithmetic.

torchvision semantics restated here (0.8.2, "TV-mem" in SURVEY Appendix A; in-repo copies cited where they exist):
  * trainable parameters: body layers 2-4 (``resnet_fpn_backbone(trainable_layers=3)``), FPN, RPN head, box head, predictor;
    every BatchNorm is FrozenBatchNorm2d; conv1 / layer1 are frozen                                    (frcnn_la.py:283)
  * RPN: anchors (32..512) x (0.5, 1, 2); Matcher(0.7, 0.3, allow_low_quality); BalancedPositiveNegativeSampler(256, 0.5);
    BCE-with-logits objectness loss; smooth-L1 (beta 1/9) box loss / #sampled; proposals from pre/post_nms_top_n_train = 2000
                                                                                                        (frcnn_la.py:154-203)
  * RoI heads: proposals + ground truth; Matcher(0.5, 0.5); sampler(512, 0.25); BoxCoder (10, 10, 5, 5); cross-entropy +
    smooth-L1 (beta 1/9) / #sampled                                                                     (frcnn_la.py:160-222)
The samplers draw from torch's CPU generator (the reference draws ``torch.randperm`` on its CUDA device: a different stream of
random numbers, the same distribution over subsets).  ``sampler="choose_k"`` (default) draws k distinct indices in O(k);
``sampler="randperm"`` consumes ``torch.randperm(n, generator=g)[:k]`` for the positives, then for the negatives, image by
image, RPN before the RoI heads -- exactly the calls torchvision's BalancedPositiveNegativeSampler makes, so a seeded CPU
generator reproduces torchvision's own samples (``tests/test_gpu_train.py::test_randperm_sampler_*``).
"""
import numpy as np
import torch

from . import train_ops as ops


# 1: 3x3 stride-2 data gradients as four phase convolutions on the un-dilated dY (4x fewer FLOPs; measured SLOWER at batch 4 -- 100.4
# vs 104.4 images/s on one box -- because the four small launches are latency-bound); default: one conv on the zero-stuffed grid
_S2_PHASES = __import__("os").environ.get("CALD_TRAIN_S2_PHASES", "0") != "0"
_PACK_PLAN = __import__("os").environ.get("CALD_TRAIN_PACK_PLAN", "1") != "0"      # all trainable layers re-packed in two launches per step


def _bn_fold(sd, prefix, eps=1e-5):
    """FrozenBatchNorm2d as y = x * scale + shift (torchvision.ops.misc.FrozenBatchNorm2d, eps 1e-5 in 0.8.x)."""
    w, b = sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double()
    rm, rv = sd[prefix + ".running_mean"].double(), sd[prefix + ".running_var"].double()
    scale = w * (rv + eps).rsqrt()
    return scale.float(), (b - rm * scale).float()


def choose_k(n, k, generator=None):
    """k distinct indices of range(n), every k-subset equally likely -- what ``torch.randperm(n)[:k]`` of torchvision's
    BalancedPositiveNegativeSampler selects, without permuting all n candidates (n is ~10^5 RPN negatives per image):
    sequential uniform draws from the CPU generator, repeats skipped."""
    if k >= n:
        return torch.arange(n)
    if 4 * k > n or n <= 8192:          # a few thousand candidates (the RoI sampler's): one permutation costs less than draw-and-dedupe
        return torch.randperm(n, generator=generator)[:k]
    got = np.empty(0, np.int64)
    while len(got) < k:
        draw = torch.randint(n, (2 * (k - len(got)) + 16,), generator=generator).numpy()
        allv = np.concatenate([got, draw])
        _, first = np.unique(allv, return_index=True)
        got = allv[np.sort(first)][:k]
    return torch.from_numpy(got)


class _LazyDict(dict):
    """A dict some of whose entries are computed on first use (zero-argument callables in `lazy`): what only tests and inspection read
    from the last forward (the proposals as a list, the sampled candidates per image, the RoI labels on the host) stays off the step's
    critical path -- the GPU is waiting while the forward's one host stop runs."""

    def __init__(self, lazy, **kw):
        super().__init__(**kw)
        self._lazy = dict(lazy)

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy.pop(key)()
            return self[key]
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class _Conv(object):
    """One conv / linear layer of the graph: forward on the packed weight, backward = weight gradient + data gradient."""

    def __init__(self, net, wnames, bnames=None, bn=None, stride=1, pad=0, cin_k=None, out_ld=None, mode=0, taps=None):
        self.net, self.stride, self.pad, self.mode, self.taps = net, stride, pad, mode, taps
        self.w = net.merged(wnames)                     # torch-layout view (several adjacent parameters merged along dim 0)
        self.b = net.merged(bnames) if bnames else None
        self.gw = net.merged_grad(wnames)
        self.gb = net.merged_grad(bnames) if bnames else None
        self.trainable = self.gw is not None
        self.scale, self.shift = (None, None) if bn is None else net.bn[bn]
        if mode == 2:
            self.Cout, self.Cin, self.K = self.w.shape[0], self.w.shape[1] // taps, 1
        elif self.w.dim() == 2:
            self.Cout, self.Cin, self.K = self.w.shape[0], self.w.shape[1], 1
        else:
            self.Cout, self.Cin, self.K = self.w.shape[0], self.w.shape[1], self.w.shape[2]
        self.cin_k = cin_k if cin_k is not None else self.Cin
        self.out_ld = out_ld if out_ld is not None else self.Cout
        self._pk = self._pkd = None
        self._pk_version = self._pkd_version = -1
        self.x = None
        net.convs.append(self)

    def _packed(self):
        if self._pk is None or (self.trainable and self._pk_version != self.net.version):
            buf = self._pk.buf if self._pk is not None else None
            self._pk = ops.PackedConv(self.w, self.b, self.scale, self.shift, CinK=self.cin_k, mode=self.mode, taps=self.taps, out=buf)
            self._pk_version = self.net.version
            self._pkd_version = -1
        return self._pk

    def _plan_packs(self):
        """The forward and data-gradient PackedConv of a trainable layer for a PackPlan (allocated, not packed, on first use)."""
        if self._pk is None:
            self._pk = ops.PackedConv(self.w, self.b, self.scale, self.shift, CinK=self.cin_k, mode=self.mode, taps=self.taps, pack=False)
        if self._pkd is None:
            self._pkd = ops.PackedConv(self.w, scale=self.scale, CinK=self.out_ld, mode=3 if self.mode == 2 else 1, taps=self.taps, pack=False)
        return [self._pk, self._pkd]

    def _is_s2(self):
        return _S2_PHASES and self.stride == 2 and self.K == 3 and self.pad == 1 and self.w.dim() == 4

    def _packed_grad(self):
        if self._is_s2():                                   # four phase sub-filters instead of one filter on the zero-stuffed grid
            if self._pkd is None or self._pkd_version != self.net.version:
                self._pkd = ops.pack_s2_grads(self.w, scale=self.scale, CinK=self.out_ld, outs=self._pkd)
                self._pkd_version = self.net.version
            return self._pkd
        if self._pkd is None or self._pkd_version != self.net.version:
            buf = self._pkd.buf if self._pkd is not None else None
            # FrozenBatchNorm layers: the scale rides in the packed filter, so the incoming gradient is the one wrt the BN output
            self._pkd = ops.PackedConv(self.w, scale=self.scale, CinK=self.out_ld, mode=3 if self.mode == 2 else 1, taps=self.taps, out=buf)
            self._pkd_version = self.net.version
        return self._pkd

    def _count(self, x, passes):
        """algorithmic FLOPs of `passes` GEMM passes (forward, data gradient, weight gradient) over input x"""
        net = self.net
        if net.flops is not None:
            if self.mode == 2 or self.w.dim() == 2:
                rows, k = x.shape[2], self.w.shape[1]
            else:
                ho, wo = (x.shape[1] + 2 * self.pad - self.K) // self.stride + 1, (x.shape[2] + 2 * self.pad - self.K) // self.stride + 1
                rows, k = x.shape[0] * ho * wo, (3 if self.Cin == 4 else self.Cin) * self.K * self.K
            net.flops += 2.0 * rows * self.Cout * k * passes

    def fwd(self, x, relu=False, residual=None, up=None, out=None):
        if self.trainable:
            self.x = x
        self._count(x, 1)
        return ops.conv(x, self._packed(), stride=self.stride, pad=self.pad, relu=relu, residual=residual, up=up, out=out, out_ld=self.out_ld)

    def bwd(self, g, need_dx=True, residual=None, accumulate=False, x=None, mask=None, wgrad=True):
        """g: gradient wrt this layer's output AFTER its FrozenBatchNorm (if any) and after the ReLU mask, row stride out_ld: the BN
        scale is folded into the packed data-gradient filter and into the weight-gradient reduction.  mask: the saved post-ReLU
        activation the returned data gradient flows into (its ReLU backward is applied in the conv epilogue)."""
        x = self.x if x is None else x
        accumulate = accumulate or self.net.accumulate_grads
        self._count(x, int(bool(need_dx)) + int(bool(wgrad)))
        side, prev = self.net.side, ops._WGRAD_CTX[0]
        if wgrad:
            if side is not None:        # the weight gradient depends on (x, g) only: issue it beside the data gradient
                side[0].wait_stream(torch.cuda.current_stream(self.net.dev))
                x.record_stream(side[0]); g.record_stream(side[0])
                ops._WGRAD_CTX[0] = side[1]
            try:
                if self.mode == 2 or self.w.dim() == 2:
                    ops.linear_wgrad(x.view(g.shape[2], -1), g.view(g.shape[2], -1), self.Cout, self.gw, self.gb, taps=self.taps or 1, accumulate=accumulate)
                else:
                    ops.conv_wgrad(x, g, self.Cin, self.Cout, self.K, self.K, self.stride, self.pad, self.gw, self.gb, accumulate=accumulate,
                                   row_scale=self.scale)
            finally:
                ops._WGRAD_CTX[0] = prev
        if not need_dx:
            return None
        pkd = self._packed_grad()
        if self.mode == 2 or self.w.dim() == 2:
            return ops.conv(g, pkd, residual=residual, mask=mask)
        if self._is_s2():
            assert residual is None
            return ops.conv_dgrad_s2(g, pkd, x.shape[1], x.shape[2], mask=mask)
        return ops.conv_dgrad(g, pkd, x.shape[1], x.shape[2], self.stride, self.pad, residual=residual, mask=mask)


class _Bottleneck(object):
    def __init__(self, net, prefix, stride, has_down, need_dx):
        self.c1 = _Conv(net, [prefix + ".conv1.weight"], bn=prefix + ".bn1")
        self.c2 = _Conv(net, [prefix + ".conv2.weight"], bn=prefix + ".bn2", stride=stride, pad=1)
        self.c3 = _Conv(net, [prefix + ".conv3.weight"], bn=prefix + ".bn3")
        self.down = _Conv(net, [prefix + ".downsample.0.weight"], bn=prefix + ".downsample.1", stride=stride) if has_down else None
        self.trainable, self.need_dx = self.c1.trainable, need_dx

    def fwd(self, x):
        self.a1 = self.c1.fwd(x, relu=True)
        self.a2 = self.c2.fwd(self.a1, relu=True)
        idt = self.down.fwd(x) if self.down is not None else x
        self.out = self.c3.fwd(self.a2, relu=True, residual=idt)
        return self.out

    def bwd(self, g, mask_input):
        """g: gradient wrt the block output with the block's final ReLU already applied backwards (g = dL/dout where out > 0, else 0).
        Returns the gradient wrt the block input x -- masked by x > 0 when `mask_input` (x is the previous block's post-ReLU output and
        nothing else is added to its gradient), unmasked otherwise -- or None when the input needs no gradient.  Every ReLU backward
        and FrozenBatchNorm scale of the block rides in a conv epilogue / packed filter: no elementwise pass."""
        ga2 = self.c3.bwd(g, mask=self.a2)
        ga1 = self.c2.bwd(ga2, mask=self.a1)
        x = self.c1.x
        if self.down is not None:
            if not self.need_dx:
                self.c1.bwd(ga1, need_dx=False); self.down.bwd(g, need_dx=False)
                return None
            gx = self.down.bwd(g)
            return self.c1.bwd(ga1, residual=gx, mask=x if mask_input else None)
        return self.c1.bwd(ga1, residual=g, mask=x if mask_input else None)


class _TrainerBase(object):
    """What the two detectors share: parameter storage, the ResNet body (forward + backward), input preparation."""
    LOSS_NAMES = ()
    MERGE_GROUPS = ()

    def _init_common(self, state_dict, num_classes, depth, min_size, max_size, device, trainable_layers, generator, sampler="choose_k"):
        if sampler not in ("choose_k", "randperm"):
            raise ValueError("sampler must be 'choose_k' or 'randperm', got %r" % (sampler,))
        self.sampler = sampler
        self.dev = torch.device(device)
        self.C, self.min_size, self.max_size = num_classes, int(min_size), int(max_size)
        self.generator = generator
        self.convs = []
        self.version = 0                                   # bumped whenever the parameters change: packed weights are rebuilt lazily
        sd = {k: (v.detach().float().cpu() if hasattr(v, "detach") else torch.from_numpy(np.asarray(v, np.float32)))
              for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        if not 0 <= trainable_layers <= 4:
            raise NotImplementedError("trainable_layers must be 0..4 (the reference uses torchvision's default 3; 5 would also train conv1)")
        frozen_layers = ["layer4", "layer3", "layer2", "layer1", "conv1"][trainable_layers:]
        def is_frozen(k):
            if ".bn" in k or "downsample.1" in k or k.startswith("backbone.body.bn1"):
                return True
            return k.startswith("backbone.body.") and any(k.startswith("backbone.body." + f) for f in frozen_layers + ["bn1"])
        # parameter storage: trainable tensors live in ONE flat buffer in state-dict order, so that tensors torchvision keeps
        # apart but the kernels treat as one matrix (RPN cls_logits + bbox_pred, predictor cls_score + bbox_pred) are adjacent
        self.names = [k for k in sd if not is_frozen(k)]
        self._merge_groups = [list(g) for g in self.MERGE_GROUPS]
        for a, b in self._merge_groups:
            self.names.remove(b)
            self.names.insert(self.names.index(a) + 1, b)
        sizes = [sd[k].numel() for k in self.names]
        pad = [(-s) % 4 for s in sizes]                    # keep every tensor 16-byte aligned...
        for grp in self._merge_groups:                     # ...except inside a merged group, which must be contiguous
            i = self.names.index(grp[0])
            assert self.names[i + 1] == grp[1], "state-dict order must keep %s adjacent" % grp
            pad[i] = 0
        offs, o = [], 0
        for s, p in zip(sizes, pad):
            offs.append(o); o += s + p
        self.flat = torch.zeros(o, dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros(o, dtype=torch.float32, device=self.dev)
        self._off = dict(zip(self.names, offs))
        self.params, self.grads, self.frozen = {}, {}, {}
        for k, off in zip(self.names, offs):
            v = self.flat[off:off + sd[k].numel()].view(sd[k].shape)
            v.copy_(sd[k])
            self.params[k] = torch.nn.Parameter(v, requires_grad=True)
            self.grads[k] = self.gflat[off:off + sd[k].numel()].view(sd[k].shape)
        for k in sd:
            if k not in self.params:
                self.frozen[k] = sd[k].to(self.dev)
        self.bn = {}
        for k in sd:
            if k.endswith(".running_var"):
                p = k[:-len(".running_var")]
                sc, sh = _bn_fold(sd, p)
                self.bn[p] = (sc.to(self.dev), sh.to(self.dev))
        # ---- graph ----
        w1 = self.frozen["backbone.body.conv1.weight"]
        self.frozen["__conv1_4ch"] = torch.cat([w1, torch.zeros_like(w1[:, :1])], dim=1).contiguous()      # the stem reads RGB0
        self.stem = _Conv(self, ["__conv1_4ch"], bn="backbone.body.bn1", stride=2, pad=3)
        nblocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
        self.layers = []
        for li, nb in enumerate(nblocks):
            name = "layer%d" % (li + 1)
            trainable = name not in frozen_layers
            prev_trainable = li > 0 and ("layer%d" % li) not in frozen_layers
            blocks = [_Bottleneck(self, "backbone.body.%s.%d" % (name, b), stride=(2 if (b == 0 and li > 0) else 1), has_down=(b == 0),
                                  need_dx=(b > 0 or prev_trainable)) for b in range(nb)]
            self.layers.append((trainable, blocks))
        self._anchors = {}
        self.last = None
        self.timing = None          # set to a list to collect (section, wall-clock) marks of forward(); each mark synchronizes
        self.flops = None           # set to 0.0 to accumulate the algorithmic GEMM FLOPs of forward() + backward()
        self.accumulate_grads = False   # True: backward() adds to the flat gradient buffer (autograd's semantics when .grad is already set)
        # weight gradients run on a second stream: at batch 4 most layers fill a fraction of the 256 CUs, and dW / dX of one layer
        # are independent.  CALD_TRAIN_SIDE_STREAM=0 keeps everything on one stream.
        self.side = self.aux = None
        self._pack_plan = None
        self._roi_pin = None
        if __import__("os").environ.get("CALD_TRAIN_SIDE_STREAM", "1") != "0":
            from .detector import get_side_ctx
            st = torch.cuda.Stream(device=self.dev)
            di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
            self.side = (st, get_side_ctx(di, st))
            # a third stream for what depends on the batch only (input upload, anchor matching, RPN sampling): its device->host copy
            # then waits for the match kernels, not for the previous step's backward still queued on the main stream
            st2 = torch.cuda.Stream(device=self.dev)
            self.aux = (st2, get_side_ctx(di, st2))

    def _join_side(self):
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side[0])


    # ---- parameter plumbing ----
    def _lookup(self, name):
        return self.params[name] if name in self.params else self.frozen[name]

    def merged(self, names):
        if names is None:
            return None
        if len(names) == 1:
            return self._lookup(names[0]).detach()
        first, n = self._lookup(names[0]).detach(), sum(self._lookup(k).numel() for k in names)
        rows = sum(self._lookup(k).shape[0] for k in names)
        return self.flat[self._off[names[0]]:self._off[names[0]] + n].view((rows,) + tuple(first.shape[1:]))

    def merged_grad(self, names):
        if names is None or names[0] not in self.params:
            return None
        first, n = self.params[names[0]], sum(self.params[k].numel() for k in names)
        rows = sum(self.params[k].shape[0] for k in names)
        return self.gflat[self._off[names[0]]:self._off[names[0]] + n].view((rows,) + tuple(first.shape[1:]))

    def parameters(self):
        return [self.params[k] for k in self.names]

    def named_parameters(self):
        return [(k, self.params[k]) for k in self.names]

    def state_dict(self):
        out = {k: v.detach().clone() for k, v in self.params.items()}
        out.update({k: v.clone() for k, v in self.frozen.items() if not k.startswith("__")})
        return out

    def parameters_changed(self):
        self.version += 1

    # ---- forward ----
    def _prepare_images(self, images):
        u8, rem = [], []
        for img in images:
            if img.dtype == torch.uint8:
                u8.append((img if img.shape[-1] == 3 else img.permute(1, 2, 0)).contiguous().to(self.dev)); rem.append(None)
                continue
            x = img.detach().to(torch.float32).to(self.dev).contiguous()
            g = (x * 255.0).round().clamp(0, 255).to(torch.uint8)
            from .detector import _u8_over_255
            r = x - _u8_over_255(x.device)[g.long()]
            u8.append(g.permute(1, 2, 0).contiguous()); rem.append(r.contiguous() if bool((r != 0).any()) else None)
        return u8, rem

    def anchors(self, Hp, Wp, level_hw):
        key = (Hp, Wp)
        if key not in self._anchors:
            if len(self._anchors) >= 16:                    # a few MB per padded batch size: keep the cache bounded over a long epoch
                self._anchors.pop(next(iter(self._anchors)))
            self._anchors[key] = ops.anchors(Hp, Wp, level_hw, self.dev, kind=self.ANCHOR_KIND)
        return self._anchors[key]

    def _sample(self, pos, neg, batch, frac):
        """BalancedPositiveNegativeSampler for one image: index tensors (CPU) of the sampled positives / negatives."""
        num_pos = min(int(batch * frac), pos.numel())
        num_neg = min(batch - num_pos, neg.numel())
        if self.sampler == "randperm":      # torchvision's own calls, in its order: positives first, then negatives
            perm_pos = torch.randperm(pos.numel(), generator=self.generator)[:num_pos]
            perm_neg = torch.randperm(neg.numel(), generator=self.generator)[:num_neg]
            return pos[perm_pos], neg[perm_neg]
        return pos[choose_k(pos.numel(), num_pos, self.generator)], neg[choose_k(neg.numel(), num_neg, self.generator)]

    def _fpn_out_fwd(self, inner):
        """The FPN output 3x3 convs (one weight per level) of all levels in one grouped launch."""
        for cv, x in zip(self.fout, inner):
            cv.x = x; cv._count(x, 1)
        return ops.conv_group(inner, [cv._packed() for cv in self.fout], pad=1)

    def _fpn_out_bwd(self, gP):
        """Weight gradients per level (side stream), data gradients of all levels in one grouped launch."""
        for cv, g in zip(self.fout, gP):
            cv.bwd(g, need_dx=False); cv._count(cv.x, 1)
        return ops.conv_group(gP, [cv._packed_grad() for cv in self.fout], pad=1)

    def _mark(self, name):
        if self.timing is not None:
            torch.cuda.synchronize(self.dev); self.timing.append((name, __import__("time").time()))

    def _repack(self):
        """Forward and data-gradient forms of every trainable weight, packed on the side stream while the main stream runs the
        frozen stem / layer 1 (the packs depend on the parameters only)."""
        if self.side is None:
            return
        st, ctx = self.side
        st.wait_stream(self._main)                          # the optimizer's update of the flat parameter buffer
        prev, ops._WGRAD_CTX[0] = ops._WGRAD_CTX[0], ctx
        try:
            with torch.cuda.stream(st):                     # torch-side helpers of the packers (sub-filter gathers) run on this stream too
                planned = [cv for cv in self.convs if cv.trainable and not cv._is_s2()] if _PACK_PLAN else []
                if planned:
                    # every layer's two forms in two launches: ~220 per-layer launches kept this stream busy for 1.5 ms, longer than
                    # the frozen layers cover (tools/train_event_timeline.py: layer 2 started 1.5 ms after the optimizer step)
                    if self._pack_plan is None:
                        self._pack_plan = ops.PackPlan([pk for cv in planned for pk in cv._plan_packs()], self.dev)
                    self._pack_plan.run()
                    for cv in planned:
                        cv._pk_version = cv._pkd_version = self.version
                for cv in self.convs:
                    if cv.trainable:
                        cv._packed(); cv._packed_grad()
        finally:
            ops._WGRAD_CTX[0] = prev
        self._packs_ready = torch.cuda.Event()
        self._packs_ready.record(st)

    def _begin_step(self):
        from .detector import get_ctx
        get_ctx(self.dev.index if self.dev.index is not None else torch.cuda.current_device())     # binds the main context to THIS stream on first use
        self._main = torch.cuda.current_stream(self.dev)
        self.version += 1                                  # whoever updated the parameters (any optimizer): repack the trainable layers
        self._repack()

    def _inputs(self, u8, rem, targets):
        """Host side of GeneralizedRCNNTransform: sizes, resized ground-truth boxes (device) and labels (host)."""
        sizes = [ops.transform_size(im.shape[0], im.shape[1], self.min_size, self.max_size) for im in u8]
        Hp, Wp = max(s[2] for s in sizes), max(s[3] for s in sizes)
        self.last_padded_hw = (Hp, Wp)
        img_sizes = [(s[0], s[1]) for s in sizes]
        gts, gt_labels = [], []
        for n_img, (im, s, t) in enumerate(zip(u8, sizes, targets)):      # resize_boxes: per-axis ratio in float32
            b = t["boxes"].detach().float().cpu().reshape(-1, 4)
            bad = (b[:, 2:] <= b[:, :2]).any(dim=1)                 # GeneralizedRCNN.forward's check (torchvision 0.8.2)
            if bool(bad.any()):
                j = int(torch.nonzero(bad)[0])
                raise ValueError("All bounding boxes should have positive height and width. Found invalid box %s for target at index %d."
                                 % (b[j].tolist(), n_img))
            rh = torch.tensor(s[0], dtype=torch.float32) / torch.tensor(im.shape[0], dtype=torch.float32)
            rw = torch.tensor(s[1], dtype=torch.float32) / torch.tensor(im.shape[1], dtype=torch.float32)
            gts.append(torch.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], dim=1).to(self.dev).contiguous())
            gt_labels.append(t["labels"].detach().long().cpu().reshape(-1))
        self._mark("inputs")
        return u8, rem, Hp, Wp, img_sizes, gts, gt_labels

    def _body(self, u8, rem, Hp, Wp, img_sizes):
        """normalize + resize + pad on the device, stem, the four body stages.  Returns feats C2..C5."""
        x = ops.preprocess(u8, img_sizes, Hp, Wp, rem)
        x = ops.maxpool(self.stem.fwd(x, relu=True))
        self._mark("stem")
        feats = []
        waited = self.side is None
        for li, (trainable, blocks) in enumerate(self.layers):
            if trainable and not waited:                    # first layer that reads a freshly packed weight
                self._main.wait_event(self._packs_ready)
                waited = True
            for blk in blocks:
                x = blk.fwd(x)
            feats.append(x)
            if li < 3:
                self._mark("layer%d" % (li + 1))
        if not waited:
            self._main.wait_event(self._packs_ready)
        self._mark("body")
        return feats

    def _inputs_and_body(self, images, targets):
        self._begin_step()
        u8, rem = self._prepare_images(images)
        u8, rem, Hp, Wp, img_sizes, gts, gt_labels = self._inputs(u8, rem, targets)
        return self._body(u8, rem, Hp, Wp, img_sizes), Hp, Wp, img_sizes, gts, gt_labels

    def _body_backward(self, gC):
        """gC[li]: gradient wrt the output of body layer li + 1 coming from the FPN laterals (None where there is none)."""
        g = None
        for li in (3, 2, 1, 0):
            trainable, blocks = self.layers[li]
            if not trainable:
                break
            if gC[li] is not None:
                g = gC[li] if g is None else ops.add(gC[li], g)
            ops.relu_bwd_(g, blocks[-1].out)                # the layer's last ReLU: the only elementwise pass of the layer
            if li < 3:
                self._mark("bwd layer%d" % (li + 2))
            for bi in range(len(blocks) - 1, -1, -1):
                # inside a layer the block input is the previous block's output and receives this gradient only: its ReLU backward is
                # fused; at the top of a layer the input also feeds an FPN lateral, whose gradient is added first (next iteration)
                g = blocks[bi].bwd(g, mask_input=bi > 0)

    def body_relu_decisions(self):
        out = {}
        nchw = lambda t: (t > 0).permute(0, 3, 1, 2).cpu()
        for li, (trainable, blocks) in enumerate(self.layers):
            for b, blk in enumerate(blocks):
                if trainable:
                    out.update({"layer%d.%d.a1" % (li + 1, b): nchw(blk.a1), "layer%d.%d.a2" % (li + 1, b): nchw(blk.a2), "layer%d.%d.out" % (li + 1, b): nchw(blk.out)})
        return out


class FasterRCNNTrainer(_TrainerBase):
    """The trainable mirror of detection/frcnn_la.py FRCNN_Feature (ResNet-50/101 FPN Faster R-CNN) on one MI355X."""
    LOSS_NAMES = ("loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg")
    MERGE_GROUPS = (("rpn.head.cls_logits.weight", "rpn.head.bbox_pred.weight"), ("rpn.head.cls_logits.bias", "rpn.head.bbox_pred.bias"),
                    ("roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.weight"),
                    ("roi_heads.box_predictor.cls_score.bias", "roi_heads.box_predictor.bbox_pred.bias"))
    ANCHOR_KIND = 0

    def __init__(self, state_dict, num_classes, depth=50, min_size=600, max_size=1000, device="cuda", trainable_layers=3,
                 rpn_pre_nms_top_n=2000, rpn_post_nms_top_n=2000, rpn_nms_thresh=0.7, rpn_fg_iou=0.7, rpn_bg_iou=0.3,
                 rpn_batch=256, rpn_pos_fraction=0.5, box_fg_iou=0.5, box_bg_iou=0.5, box_batch=512, box_pos_fraction=0.25,
                 bbox_reg_weights=(10.0, 10.0, 5.0, 5.0), generator=None, sampler="choose_k"):
        self._init_common(state_dict, num_classes, depth, min_size, max_size, device, trainable_layers, generator, sampler)
        self.cfg = dict(pre_n=rpn_pre_nms_top_n, post_n=rpn_post_nms_top_n, nms=rpn_nms_thresh, rpn_fg=rpn_fg_iou, rpn_bg=rpn_bg_iou,
                        rpn_batch=rpn_batch, rpn_pos=rpn_pos_fraction, box_fg=box_fg_iou, box_bg=box_bg_iou, box_batch=box_batch,
                        box_pos=box_pos_fraction, w=tuple(bbox_reg_weights))
        self.lat = [_Conv(self, ["backbone.fpn.inner_blocks.%d.weight" % i], ["backbone.fpn.inner_blocks.%d.bias" % i]) for i in range(4)]
        self.fout = [_Conv(self, ["backbone.fpn.layer_blocks.%d.weight" % i], ["backbone.fpn.layer_blocks.%d.bias" % i], pad=1) for i in range(4)]
        self.rpn_conv = _Conv(self, ["rpn.head.conv.weight"], ["rpn.head.conv.bias"], pad=1)
        self.rpn_head = _Conv(self, self._merge_groups[0], self._merge_groups[1], out_ld=16)
        self.fc6 = _Conv(self, ["roi_heads.box_head.fc6.weight"], ["roi_heads.box_head.fc6.bias"], mode=2, taps=49)
        self.fc7 = _Conv(self, ["roi_heads.box_head.fc7.weight"], ["roi_heads.box_head.fc7.bias"])
        self.pred_ld = ops.round_up(5 * num_classes, 4)
        self.pred = _Conv(self, self._merge_groups[2], self._merge_groups[3], out_ld=self.pred_ld)
        assert self.pred.Cout == 5 * num_classes, "box predictor does not match num_classes"
        # CALD_TRAIN_SPECULATE=1: the first half of the backward's RPN branch is enqueued during the forward's RoI-sampling window (see
        # forward).  Off by default: it paid (-0.8 ms) while the weight-gradient stream was the longer one of the backward; since that
        # stream got faster the step is bound by the data-gradient chain, which wants the RPN branch beside the box-head branch: +0.7 ms
        self.speculate = __import__("os").environ.get("CALD_TRAIN_SPECULATE", "0") != "0"
        self.grad_wanted = True         # TrainableDetector clears it under torch.no_grad()
        self._spec_grads = None

    def forward(self, images, targets, proposals_override=None):
        """Training forward.  Returns the four losses as 1-element device tensors (no autograd) and keeps what backward needs."""
        cfg, N, Ccls = self.cfg, len(images), self.C
        mark = self._mark
        mark("start")
        self._begin_step()
        u8, rem = self._prepare_images(images)              # on the main stream: the images may have just been produced there
        aux, prev = self.aux, ops._WGRAD_CTX[0]
        import contextlib
        with (torch.cuda.stream(aux[0]) if aux is not None else contextlib.nullcontext()):
            if aux is not None:
                ops._WGRAD_CTX[0] = aux[1]
            try:
                u8, rem, Hp, Wp, img_sizes, gts, gt_labels = self._inputs(u8, rem, targets)
                # The RPN targets depend on the anchors and the ground truth only: match + sample them BEFORE the network is enqueued, so the
                # device->host copy of the match results does not wait for (and the host-side sampling does not stall) the body's kernels.
                level_hw = [(Hp // 4, Wp // 4), (Hp // 8, Wp // 8), (Hp // 16, Wp // 16), (Hp // 32, Wp // 32)]
                level_hw.append(((level_hw[3][0] - 1) // 2 + 1, (level_hw[3][1] - 1) // 2 + 1))
                head_sizes = [N * h * w * 16 for h, w in level_hw]
                anchors = self.anchors(Hp, Wp, level_hw)
                A_img = anchors.shape[0]
                # ---- RPN targets and sampling (anchor order: level, y, x, anchor).  One device->host copy of all match results, host-side
                # sampling (torch CPU generator), one host->device copy of every index the loss kernels need. ----
                lvl_start = np.cumsum([0] + [h * w * 3 for h, w in level_hw])
                head_off = np.cumsum([0] + head_sizes)
                lvl_pix = np.array([h * w for h, w in level_hw])
                def head_offsets(img, idx):                                  # float offset of anchor idx's objectness logit in head_flat
                    l = np.searchsorted(lvl_start, idx, side="right") - 1
                    rel = idx - lvl_start[l]
                    pix, a = rel // 3, rel % 3
                    return head_off[l] + (img * lvl_pix[l] + pix) * 16 + a, a
                n_gt = [int(g.shape[0]) for g in gts]
                gt_off = np.cumsum([0] + n_gt)
                gt_labels_cat = np.ascontiguousarray(torch.cat(gt_labels).numpy()) if sum(n_gt) else np.zeros(1, np.int64)
                gts_all = torch.cat(gts + [torch.zeros((1, 4), device=self.dev)])      # last row: the "matched box" of images without ground truth
                matched_dev = torch.full((N, A_img), -1, dtype=torch.int32, device=self.dev)
                for i in range(N):
                    if n_gt[i]:
                        ops.match(anchors, gts[i], cfg["rpn_fg"], cfg["rpn_bg"], True, out=matched_dev[i])
                matched_all = matched_dev.cpu().numpy()
                obj_idx, obj_lab, box_idx, anc_idx, gt_idx = [], [], [], [], []
                rpn_samples, box_samples = [], []
                for i in range(N):
                    m = matched_all[i]
                    pos, neg = torch.from_numpy(np.flatnonzero(m >= 0)), torch.from_numpy(np.flatnonzero(m == -1))
                    sp, sn = self._sample(pos, neg, cfg["rpn_batch"], cfg["rpn_pos"])
                    sp, sn = np.sort(sp.numpy()), np.sort(sn.numpy())
                    rpn_samples.append((sp, sn))
                    op_, ap_ = head_offsets(i, sp); on_, _ = head_offsets(i, sn)
                    obj_idx += [op_, on_]; obj_lab += [np.ones(len(op_), np.float32), np.zeros(len(on_), np.float32)]
                    box_idx.append(op_ - ap_ + 3 + 4 * ap_)                  # channel 3 + 4a of the same pixel
                    anc_idx.append(sp); gt_idx.append(gt_off[i] + m[sp])
                obj_idx, box_idx, anc_idx, gt_idx = [np.concatenate(v).astype(np.int64) for v in (obj_idx, box_idx, anc_idx, gt_idx)]
                packed = torch.from_numpy(np.concatenate([obj_idx, box_idx, anc_idx, gt_idx])).to(self.dev)
                n_obj, n_pos = len(obj_idx), len(box_idx)
                obj_idx, box_idx = packed[:n_obj], packed[n_obj:n_obj + n_pos]
                anc_sel, gt_sel = packed[n_obj + n_pos:n_obj + 2 * n_pos], packed[n_obj + 2 * n_pos:]
                obj_lab = torch.from_numpy(np.concatenate(obj_lab)).to(self.dev)
                rpn_tgt = ops.box_encode(gts_all[gt_sel], anchors[anc_sel], (1.0, 1.0, 1.0, 1.0))
                mark("rpn targets")
            finally:
                ops._WGRAD_CTX[0] = prev
        if aux is not None:                                 # what the main stream consumes from the batch-only stream
            self._main.wait_stream(aux[0])
            for t in list(gts) + [gts_all, packed, obj_lab, rpn_tgt]:
                t.record_stream(self._main)
        feats = self._body(u8, rem, Hp, Wp, img_sizes)
        # FPN (top-down), LastLevelMaxPool
        inner = [None] * 4
        inner[3] = self.lat[3].fwd(feats[3])
        for i in (2, 1, 0):
            inner[i] = self.lat[i].fwd(feats[i], up=inner[i + 1])
        P = self._fpn_out_fwd(inner)
        P.append(ops.subsample2(P[3]))
        assert level_hw == [(p.shape[1], p.shape[2]) for p in P]
        # RPN head on the five levels: outputs in ONE buffer so that the loss kernels address (level, pixel, channel) by offset
        head_flat = torch.zeros(sum(head_sizes), dtype=torch.float32, device=self.dev)
        heads, o = [], 0
        tl = ops.conv_group(P, self.rpn_conv._packed(), pad=1, relu=True)            # shared 3x3 conv on the five levels: one launch
        for i, (h, w) in enumerate(level_hw):
            heads.append(ops.conv(tl[i], self.rpn_head._packed(), out=head_flat[o:o + head_sizes[i]].view(N, h, w, 16), out_ld=16))
            self.rpn_conv._count(P[i], 1); self.rpn_head._count(tl[i], 1)
            o += head_sizes[i]
        mark("fpn+rpn head")
        spec = None
        rpn_state = dict(N=N, P=P, tl=tl, head_flat=head_flat, head_sizes=head_sizes, level_hw=level_hw, obj_idx=obj_idx, obj_lab=obj_lab,
                         box_idx=box_idx, rpn_tgt=rpn_tgt)
        props_ready = None
        if proposals_override is None:
            props, counts = ops.rpn_proposals(heads, Hp, Wp, img_sizes, cfg["pre_n"], cfg["post_n"], cfg["nms"], 1e-3)
            if self.speculate and aux is not None and self.grad_wanted:
                # The host now needs the proposals (counts, then the match results) to draw the RoI samples: ~1 ms during which the main
                # stream would sit empty.  The RPN branch of the backward pass depends on nothing that comes later, so its first half
                # (loss gradients, the head's gradients, the 3x3 conv's weight gradients) is enqueued here, behind the proposal
                # kernels, for unit upstream gradients (losses.backward() of the plain sum, the reference's loop); backward() uses it
                # when its upstream gradients are 1 and recomputes otherwise.  The second half (the 3x3 conv's data gradients, ~2 ms)
                # stays in backward(): there it runs beside the box-head branch, which would otherwise be alone on the chip.
                props.record_stream(aux[0])
                counts_h = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
                counts_h.copy_(counts, non_blocking=True)
                props_ready = torch.cuda.Event(); props_ready.record(self._main)
                spec = self._rpn_branch_weights(rpn_state, 1.0, 1.0, speculative=True)
                props_ready.synchronize()
                counts = counts_h.tolist()
                proposals = [props[i, :counts[i]] for i in range(N)]
            else:
                # ONE host stop instead of two: the proposal counts are not waited for.  The RoI candidates are matched in a fixed-row
                # table (image i: its post_n proposal slots, used or not, then its ground truth) and the counts travel to the host in
                # the same copy as the match results; unused slots are dropped there.  Same candidates, same draws, same rows.
                proposals = None
        else:
            proposals = [p.to(self.dev).float().contiguous() for p in proposals_override]
        mark("proposals")
        # ---- RoI sampling (on the batch-only stream when the main stream is busy with the speculative branch: the device->host copy
        # of the match results then waits for the match kernels only) ----
        on_aux = props_ready is not None
        with (torch.cuda.stream(aux[0]) if on_aux else contextlib.nullcontext()):
            if on_aux:
                aux[0].wait_event(props_ready)
                ops._WGRAD_CTX[0] = aux[1]
            try:
                fixed_rows = proposals is None
                slots = [(int(props.shape[1]) if fixed_rows else int(proposals[i].shape[0])) for i in range(N)]
                n_pr = [slots[i] + n_gt[i] for i in range(N)]
                pr_off = np.cumsum([0] + n_pr)
                src = [props[i] for i in range(N)] if fixed_rows else proposals
                pr_all = torch.cat([t for i in range(N) for t in ((src[i], gts[i]) if n_gt[i] else (src[i],))]).contiguous()
                matched_dev = torch.full((int(pr_off[-1]) + (N if fixed_rows else 0),), -1, dtype=torch.int32, device=self.dev)
                for i in range(N):
                    if n_gt[i]:
                        ops.match(pr_all[pr_off[i]:pr_off[i + 1]], gts[i], cfg["box_fg"], cfg["box_bg"], False, out=matched_dev[pr_off[i]:pr_off[i + 1]])
                if fixed_rows:
                    matched_dev[int(pr_off[-1]):] = counts.to(torch.int32)
                matched_all = matched_dev.cpu().numpy()
                # ---- the GPU waits from here to the host->device copy below: everything in between is host time on the step's critical path ----
                T = int(pr_off[-1])
                if fixed_rows:
                    counts_np = matched_all[T:]
                    counts = None                           # as a list only where someone asks (self.last["proposals"], the randperm sampler)
                else:
                    counts_np, counts = None, list(slots)
                lazy = {}
                if fixed_rows:
                    lazy["proposals"] = lambda props=props, c=counts_np.copy(): [props[i, :int(c[i])] for i in range(N)]
                if self.sampler == "choose_k":
                    # one C call for the whole batch (cald_train_roi_sample_host): labels, the balanced sampler (the k smallest of iid
                    # uniform keys per class -- one generator call for all images), every index list of the loss kernels, written into
                    # pinned memory and uploaded as one block; one kernel (cald_train_roi_gather) then builds RoIAlign's rows and the
                    # regression targets.  The numpy loop of the other branch + eight small torch ops cost 0.6 ms of GPU idle per step.
                    # Keys are drawn per (image, post_n proposal slots + ground truth) whatever the table's shape, so that the fixed-row and
                    # the compact table (speculative path, proposals handed in) select the same candidates from the same generator state.
                    post = cfg["post_n"]
                    if all(sl <= post for sl in slots):
                        keys = torch.rand(sum(post + g for g in n_gt), generator=self.generator, dtype=torch.float64).numpy()
                        if any(sl != post for sl in slots):
                            ko = np.cumsum([0] + [post + g for g in n_gt])
                            keys = np.concatenate([keys[np.r_[ko[i]:ko[i] + slots[i], ko[i] + post:ko[i] + post + n_gt[i]]] for i in range(N)])
                    else:
                        keys = torch.rand(T, generator=self.generator, dtype=torch.float64).numpy()
                    cap = N * cfg["box_batch"]
                    if self._roi_pin is None or self._roi_pin.numel() < 6 * cap:
                        self._roi_pin = torch.empty(6 * cap, dtype=torch.int64, pin_memory=True)
                    pin_np = self._roi_pin.numpy()
                    _, _, R, n_posrows, per_img = ops.roi_sample_host(slots, n_gt, counts_np, matched_all, gt_labels_cat, keys, cfg["box_batch"], cfg["box_pos"],
                                                                      self.pred_ld, Ccls, out=pin_np)
                    packed2 = torch.empty(6 * cap, dtype=torch.int64, device=self.dev)
                    packed2.copy_(self._roi_pin[:6 * cap], non_blocking=True)       # the pinned block is rewritten only after the next step's stop
                    rois, box_tgt = ops.roi_gather(pr_all, gts_all, packed2, cap, R, n_posrows, cfg["w"])
                    keep_sel, labels_dev = packed2[:R], packed2[2 * cap:2 * cap + R]
                    pred_idx = packed2[3 * cap:3 * cap + n_posrows]
                    keep_np, roi_labels_np = pin_np[:R].copy(), pin_np[2 * cap:2 * cap + R].copy()
                    def box_samples_fn(keep=keep_np, lab=roi_labels_np, per_img=per_img, cn=None if counts_np is None else counts_np.copy(), slots=list(slots),
                                       pr_off=pr_off.copy()):
                        out, o = [], 0
                        for i in range(N):                      # table row -> compact candidate number (used proposals, then the ground truth)
                            r = keep[o:o + per_img[i]] - pr_off[i]
                            c = np.where(r < slots[i], r, r - slots[i] + (slots[i] if cn is None else int(cn[i])))
                            l = lab[o:o + per_img[i]]
                            out.append((c[l > 0], c[l == 0])); o += per_img[i]
                        return out
                    box_samples = None
                else:
                    if counts is None:
                        counts = [int(v) for v in counts_np]
                    # compact candidate number -> row of the fixed-row table (the used proposal slots, then the ground truth)
                    rowmap = [np.concatenate([np.arange(counts[i]), slots[i] + np.arange(n_gt[i])]).astype(np.int64) for i in range(N)]
                    keep_all, lab_all, gtsel_all, img_col = [], [], [], []
                    for i in range(N):
                        m = matched_all[pr_off[i]:pr_off[i + 1]][rowmap[i]]
                        if n_gt[i]:
                            labels = gt_labels[i].numpy()[np.maximum(m, 0)].copy()
                            labels[m == -1] = 0
                            labels[m == -2] = -1
                        else:
                            labels = np.zeros(len(m), np.int64)
                        pos, neg = torch.from_numpy(np.flatnonzero(labels >= 1)), torch.from_numpy(np.flatnonzero(labels == 0))
                        sp, sn = self._sample(pos, neg, cfg["box_batch"], cfg["box_pos"])
                        box_samples.append((np.sort(sp.numpy()), np.sort(sn.numpy())))
                        keep = np.sort(np.concatenate([sp.numpy(), sn.numpy()]))
                        keep_all.append(pr_off[i] + rowmap[i][keep]); lab_all.append(labels[keep])
                        gtsel_all.append(gt_off[i] + np.maximum(m[keep], 0) if n_gt[i] else np.full(len(keep), gt_off[-1], np.int64))
                        img_col.append(np.full(len(keep), float(i), np.float32))
                    roi_labels_np = np.concatenate(lab_all).astype(np.int64)
                    R = len(roi_labels_np)
                    pos_rows = np.flatnonzero(roi_labels_np > 0)
                    n_posrows = len(pos_rows)
                    pred_idx_np = pos_rows * self.pred_ld + Ccls + 4 * roi_labels_np[pos_rows]
                    packed2 = torch.from_numpy(np.concatenate([np.concatenate(keep_all), np.concatenate(gtsel_all), roi_labels_np, pred_idx_np, pos_rows]).astype(np.int64)).to(self.dev)
                    img_col_dev = torch.from_numpy(np.concatenate(img_col)).to(self.dev)
                    box_samples_fn = None
                    keep_sel, gt_sel2, labels_dev = packed2[:R], packed2[R:2 * R], packed2[2 * R:3 * R]
                    pred_idx, pos_sel = packed2[3 * R:3 * R + n_posrows], packed2[3 * R + n_posrows:3 * R + 2 * n_posrows]
                    boxes = pr_all[keep_sel]
                    rois = torch.cat([img_col_dev[:, None], boxes], dim=1).contiguous()
                    roi_gt = gts_all[gt_sel2]
                    box_tgt = ops.box_encode(roi_gt.contiguous(), boxes.contiguous(), cfg["w"])[pos_sel].contiguous()
                lazy["roi_labels"] = lambda l=roi_labels_np: torch.from_numpy(np.ascontiguousarray(l))
                if "proposals" not in lazy:
                    lazy["proposals"] = lambda p=proposals: p
            finally:
                ops._WGRAD_CTX[0] = prev
        if on_aux:
            self._main.wait_stream(aux[0])
            for t in (packed2, rois, box_tgt):
                t.record_stream(self._main)
        mark("roi sampling")
        # ---- box head ----
        roi_rows = ops.roi_align(P[:4], rois)
        f6 = self.fc6.fwd(roi_rows.view(1, 1, R, -1), relu=True)
        f7 = self.fc7.fwd(f6, relu=True)
        pred = self.pred.fwd(f7)
        self.last = _LazyDict(lazy, N=N, R=R, feats=feats, inner=inner, P=P, tl=tl, heads=heads, head_flat=head_flat, head_sizes=head_sizes, level_hw=level_hw,
                         obj_idx=obj_idx, obj_lab=obj_lab, box_idx=box_idx, rpn_tgt=rpn_tgt, rois=rois, roi_rows=roi_rows, f6=f6, f7=f7, pred=pred,
                         labels=labels_dev, pred_idx=pred_idx, box_tgt=box_tgt,
                         samples=_LazyDict({"box": box_samples_fn} if box_samples_fn is not None else {}, rpn=rpn_samples,
                                           **({} if box_samples_fn is not None else {"box": box_samples})), spec=spec)
        mark("box head")
        losses = {
            "loss_classifier": ops.softmax_ce(pred.view(R, -1), labels_dev, Ccls),
            "loss_box_reg": ops.smooth_l1(pred, pred_idx, box_tgt, 1.0 / 9, R),
            "loss_objectness": ops.bce_logits(head_flat, obj_idx, obj_lab),
            "loss_rpn_box_reg": ops.smooth_l1(head_flat, box_idx, rpn_tgt, 1.0 / 9, n_obj),
        }
        mark("losses")
        return losses

    def relu_decisions(self):
        """{name: bool NCHW / [R, C] CPU tensor}: which side every ReLU of the last forward's differentiated part took (for
        gradient checks against a higher-precision restatement, which must take the same branches)."""
        L, out = self.last, self.body_relu_decisions()
        nchw = lambda t: (t > 0).permute(0, 3, 1, 2).cpu()
        for i, t in enumerate(L["tl"]):
            out["rpn.%d" % i] = nchw(t)
        out["fc6"] = (L["f6"] > 0).view(L["R"], -1).cpu(); out["fc7"] = (L["f7"] > 0).view(L["R"], -1).cpu()
        return out

    def _rpn_branch_weights(self, L, g_obj, g_reg, speculative=False):
        """First half of the RPN branch: RPN losses -> gradient of the 1x1 head's output -> the weight gradients of the head and of the
        3x3 conv (shared weights: they accumulate over the five levels) and the head's data gradient (with the conv's ReLU backward).
        Returns that gradient per level -- what the 3x3 conv's data gradient (_rpn_branch_data) starts from.  speculative: the weight
        gradients go to buffers of their own (whether backward() must overwrite or add to the flat buffer is not known yet)."""
        N, P, level_hw = L["N"], L["P"], L["level_hw"]
        layers = (self.rpn_head, self.rpn_conv)
        saved = [(c.gw, c.gb) for c in layers]
        if speculative:
            if self._spec_grads is None:
                self._spec_grads = [(torch.zeros_like(c.gw), torch.zeros_like(c.gb)) for c in layers]
            for c, (gw, gb) in zip(layers, self._spec_grads):
                c.gw, c.gb = gw, gb
        acc_saved, self.accumulate_grads = self.accumulate_grads, (False if speculative else self.accumulate_grads)
        try:
            ghead_flat = torch.zeros_like(L["head_flat"])
            ops.bce_logits(L["head_flat"], L["obj_idx"], L["obj_lab"], grad=ghead_flat, gscale=g_obj)
            ops.smooth_l1(L["head_flat"], L["box_idx"], L["rpn_tgt"], 1.0 / 9, L["obj_idx"].numel(), grad=ghead_flat, gscale=g_reg)
            o, ghs = 0, []
            for i, (h, w) in enumerate(level_hw):
                ghs.append(ghead_flat[o:o + L["head_sizes"][i]].view(N, h, w, 16)); o += L["head_sizes"][i]
                self.rpn_head.bwd(ghs[i], need_dx=False, accumulate=i > 0, x=L["tl"][i])
                self.rpn_head._count(L["tl"][i], 1)
            gts_ = ops.conv_group(ghs, self.rpn_head._packed_grad(), masks=L["tl"])       # 1x1 head: data gradient of the five levels + ReLU backward, one launch
            for i in range(5):
                self.rpn_conv.bwd(gts_[i], need_dx=False, accumulate=i > 0, x=P[i])
        finally:
            self.accumulate_grads = acc_saved
            for c, (gw, gb) in zip(layers, saved):
                c.gw, c.gb = gw, gb
        return gts_

    def _rpn_branch_data(self, L, gts_):
        """Second half: the 3x3 conv's data gradient on the five levels -> gradients wrt P2..P5 and the pooled level."""
        P = L["P"]
        for i in range(5):
            self.rpn_conv._count(P[i], 1)
        g = ops.conv_group(gts_, self.rpn_conv._packed_grad(), pad=1)       # the five levels in one launch, like the forward
        return g[:4], g[4]

    def _commit_speculative(self):
        """The speculative weight gradients of the RPN branch become the real ones: moved (or added) into the flat gradient buffer, on
        the stream that computed them."""
        st = self.side[0] if self.side is not None else torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(st):
            for c, (gw, gb) in zip((self.rpn_head, self.rpn_conv), self._spec_grads):
                for dst, src in ((c.gw, gw), (c.gb, gb)):
                    if dst is None:
                        continue
                    if self.accumulate_grads:
                        dst.add_(src)
                    else:
                        dst.copy_(src)

    # ---- backward ----
    def backward(self, gscale=(1.0, 1.0, 1.0, 1.0)):
        """Gradients of sum_i gscale[i] * loss_i (order: classifier, box_reg, objectness, rpn_box_reg) into the flat gradient buffer."""
        L = self.last
        N, R, P, level_hw = L["N"], L["R"], L["P"], L["level_hw"]
        # The box-head branch (predictor -> fc7 -> fc6 -> RoIAlign backward) and the RPN branch meet only at the FPN outputs: the former
        # is issued on the batch-only stream (idle during the backward pass), the latter on the main stream.
        import contextlib
        aux, prev = self.aux, ops._WGRAD_CTX[0]
        main = torch.cuda.current_stream(self.dev)
        if aux is not None:
            aux[0].wait_stream(main)
        with (torch.cuda.stream(aux[0]) if aux is not None else contextlib.nullcontext()):
            if aux is not None:
                ops._WGRAD_CTX[0] = aux[1]
            try:
                gpred = torch.zeros_like(L["pred"])
                ops.softmax_ce(L["pred"].view(R, -1), L["labels"], self.C, grad=gpred, gscale=gscale[0])
                ops.smooth_l1(L["pred"], L["pred_idx"], L["box_tgt"], 1.0 / 9, R, grad=gpred, gscale=gscale[1])
                g7 = self.pred.bwd(gpred, mask=L["f7"])
                g6 = self.fc7.bwd(g7, mask=L["f6"])
                groi = self.fc6.bwd(g6)
                gP_roi = [torch.zeros_like(p) for p in P[:4]]
                ops.roi_align_bwd_(gP_roi, L["rois"], groi.view(R, 49, -1))
                self._mark("bwd box head (aux)")
            finally:
                ops._WGRAD_CTX[0] = prev
        spec = L.get("spec")
        if spec is not None and float(gscale[2]) == 1.0 and float(gscale[3]) == 1.0:
            self._commit_speculative()
            gts_ = spec
        else:
            gts_ = self._rpn_branch_weights(L, gscale[2], gscale[3])
        L["spec"] = None
        gP, gpool = self._rpn_branch_data(L, gts_)      # on the main stream, beside the box-head branch on the batch-only stream
        self._mark("bwd rpn branch")
        if aux is not None:
            main.wait_stream(aux[0])
            for t in gP_roi:
                t.record_stream(main)
        gP = [ops.add(gP[i], gP_roi[i]) for i in range(4)]
        gP[3] = ops.add(gP[3], ops.dilate(gpool, 2, P[3].shape[1], P[3].shape[2]))          # LastLevelMaxPool (kernel 1, stride 2)
        # FPN
        ginner = self._fpn_out_bwd(gP[:4])
        for i in range(1, 4):
            ops.upsample_bwd_(ginner[i - 1], ginner[i])
        gC = [None] * 4
        for i in range(4):
            need = self.layers[i][0]                                                       # the body layer producing feats[i] is trainable
            gC[i] = self.lat[i].bwd(ginner[i], need_dx=need)
        self._mark("bwd fpn")
        self._body_backward(gC)
        self._mark("bwd body dgrad")
        self._join_side()
        return self.grads


class RetinaNetTrainer(_TrainerBase):
    """The trainable mirror of detection/retinanet_cal.py RetinaNet (ResNet-50 FPN, P3-P7, 9 anchors per location): training forward
    :545-564, matcher :389-400 (IoU 0.5 / 0.4, low-quality matches), classification loss :100-133 (sigmoid focal loss, sum over the
    anchors outside the ignore band / max(1, #foreground), mean over images), regression loss :185-221 (L1 on the foreground
    anchors / max(1, #foreground), mean over images).  As in the reference, every image must carry at least one box."""
    LOSS_NAMES = ("classification", "bbox_regression")
    ANCHOR_KIND = 1

    def __init__(self, state_dict, num_classes, depth=50, min_size=600, max_size=1000, device="cuda", trainable_layers=3,
                 fg_iou_thresh=0.5, bg_iou_thresh=0.4, generator=None):
        self._init_common(state_dict, num_classes, depth, min_size, max_size, device, trainable_layers, generator)
        self.cfg = dict(fg=fg_iou_thresh, bg=bg_iou_thresh)
        cin = [None, 0, 1, 2]                                    # body layer index -> FPN block index (returned_layers = [2, 3, 4])
        self.lat = [_Conv(self, ["backbone.fpn.inner_blocks.%d.weight" % i], ["backbone.fpn.inner_blocks.%d.bias" % i]) for i in range(3)]
        self.fout = [_Conv(self, ["backbone.fpn.layer_blocks.%d.weight" % i], ["backbone.fpn.layer_blocks.%d.bias" % i], pad=1) for i in range(3)]
        self.p6 = _Conv(self, ["backbone.fpn.extra_blocks.p6.weight"], ["backbone.fpn.extra_blocks.p6.bias"], stride=2, pad=1)
        self.p7 = _Conv(self, ["backbone.fpn.extra_blocks.p7.weight"], ["backbone.fpn.extra_blocks.p7.bias"], stride=2, pad=1)
        tower = lambda head: [_Conv(self, ["head.%s.conv.%d.weight" % (head, 2 * j)], ["head.%s.conv.%d.bias" % (head, 2 * j)], pad=1) for j in range(4)]
        self.cls_tower, self.reg_tower = tower("classification_head"), tower("regression_head")
        self.K = num_classes
        self.cls_ld = ops.round_up(9 * num_classes, 4)
        self.cls_out = _Conv(self, ["head.classification_head.cls_logits.weight"], ["head.classification_head.cls_logits.bias"], pad=1, out_ld=self.cls_ld)
        self.reg_out = _Conv(self, ["head.regression_head.bbox_reg.weight"], ["head.regression_head.bbox_reg.bias"], pad=1)
        assert self.cls_out.Cout == 9 * num_classes and self.reg_out.Cout == 36, "RetinaNet heads must have 9 * num_classes / 36 outputs"

    def forward(self, images, targets):
        N, K, mark = len(images), self.K, self._mark
        mark("start")
        self._begin_step()
        u8, rem = self._prepare_images(images)              # on the main stream: the images may have just been produced there
        # Everything that depends on the batch only (ground-truth upload, anchor matching, regression targets) goes first, on its own
        # stream: the device->host copy of the match results then does not wait for the network's kernels.
        aux, prev = self.aux, ops._WGRAD_CTX[0]
        import contextlib
        with (torch.cuda.stream(aux[0]) if aux is not None else contextlib.nullcontext()):
            if aux is not None:
                ops._WGRAD_CTX[0] = aux[1]
            try:
                u8, rem, Hp, Wp, img_sizes, gts, gt_labels = self._inputs(u8, rem, targets)
                if any(g.shape[0] == 0 for g in gts):
                    raise ValueError("RetinaNet training needs at least one ground-truth box per image (retinanet_cal.py:110-124)")
                level_hw = [(Hp // 8, Wp // 8), (Hp // 16, Wp // 16), (Hp // 32, Wp // 32)]
                for _ in range(2):                              # P6, P7: 3x3 stride-2 convs with padding 1
                    level_hw.append(((level_hw[-1][0] - 1) // 2 + 1, (level_hw[-1][1] - 1) // 2 + 1))
                level_pix = [h * w for h, w in level_hw]
                cls_sizes, reg_sizes = [N * n * self.cls_ld for n in level_pix], [N * n * 36 for n in level_pix]
                anchors = self.anchors(Hp, Wp, level_hw)
                A_tot = anchors.shape[0]
                matched_dev = torch.empty((N, A_tot), dtype=torch.int32, device=self.dev)
                for i in range(N):
                    ops.match(anchors, gts[i], self.cfg["fg"], self.cfg["bg"], True, out=matched_dev[i])
                matched_all = matched_dev.cpu().numpy()
                n_gt = [int(g.shape[0]) for g in gts]
                gt_off = np.cumsum([0] + n_gt)
                gts_all = torch.cat(gts)
                lvl_start = np.cumsum([0] + [n * 9 for n in level_pix])
                reg_off = np.cumsum([0] + reg_sizes)
                lvl_pix = np.array(level_pix)
                box_idx, anc_idx, gt_idx, wts, nfg = [], [], [], [], []
                for i in range(N):
                    m = matched_all[i]
                    fg = np.flatnonzero(m >= 0)
                    l = np.searchsorted(lvl_start, fg, side="right") - 1
                    rel = fg - lvl_start[l]
                    pix, a = rel // 9, rel % 9
                    box_idx.append(reg_off[l] + (i * lvl_pix[l] + pix) * 36 + 4 * a)
                    anc_idx.append(fg); gt_idx.append(gt_off[i] + m[fg])
                    nfg.append(len(fg)); wts.append(np.full(len(fg), 1.0 / (max(1, len(fg)) * N), np.float32))
                box_idx, anc_idx, gt_idx = [np.concatenate(v).astype(np.int64) for v in (box_idx, anc_idx, gt_idx)]
                packed = torch.from_numpy(np.concatenate([box_idx, anc_idx, gt_idx])).to(self.dev)
                nb = len(box_idx)
                box_idx, anc_sel, gt_sel = packed[:nb], packed[nb:2 * nb], packed[2 * nb:]
                fl = torch.from_numpy(np.concatenate([np.concatenate(wts), np.array([1.0 / (max(1, n) * N) for n in nfg], np.float32)])).to(self.dev)
                box_w, img_w = fl[:nb], fl[nb:]
                reg_tgt = ops.box_encode(gts_all[gt_sel], anchors[anc_sel], (1.0, 1.0, 1.0, 1.0))
                gt_labels_dev = torch.cat(gt_labels).to(self.dev)
                gt_off_dev = torch.from_numpy(gt_off.astype(np.int32)).to(self.dev)
            finally:
                ops._WGRAD_CTX[0] = prev
        if aux is not None:
            self._main.wait_stream(aux[0])
            for t in list(gts) + [gts_all, packed, fl, reg_tgt, matched_dev, gt_labels_dev, gt_off_dev]:
                t.record_stream(self._main)
        mark("targets")
        feats = self._body(u8, rem, Hp, Wp, img_sizes)
        inner = [None] * 3
        inner[2] = self.lat[2].fwd(feats[3])
        for i in (1, 0):
            inner[i] = self.lat[i].fwd(feats[i + 1], up=inner[i + 1])
        P = self._fpn_out_fwd(inner)
        P.append(self.p6.fwd(P[2]))
        p6_relu = ops.relu_bwd_(ops.add(P[3]), P[3])                       # relu(p6): x * (x > 0) on a copy
        P.append(self.p7.fwd(p6_relu))
        assert level_hw == [(p.shape[1], p.shape[2]) for p in P]
        # head outputs of the five levels in ONE buffer each (level block l = [N][pix_l][ld])
        cls_flat = torch.zeros(sum(cls_sizes), dtype=torch.float32, device=self.dev)
        reg_flat = torch.empty(sum(reg_sizes), dtype=torch.float32, device=self.dev)
        # both towers on the five levels: ten problems of one shape per launch (the small levels fill the tail of the large ones)
        acts = {"cls": [[P[l]] for l in range(5)], "reg": [[P[l]] for l in range(5)]}
        for j in range(4):
            xs = [acts["cls"][l][-1] for l in range(5)] + [acts["reg"][l][-1] for l in range(5)]
            ys = ops.conv_group(xs, [self.cls_tower[j]._packed()] * 5 + [self.reg_tower[j]._packed()] * 5, pad=1, relu=True)
            for l in range(5):
                self.cls_tower[j]._count(xs[l], 1); self.reg_tower[j]._count(xs[5 + l], 1)
                acts["cls"][l].append(ys[l]); acts["reg"][l].append(ys[5 + l])
        oc = orr = 0
        cls_views, reg_views = [], []
        for l, (h, w) in enumerate(level_hw):
            cls_views.append(cls_flat[oc:oc + cls_sizes[l]].view(N, h, w, self.cls_ld)); reg_views.append(reg_flat[orr:orr + reg_sizes[l]].view(N, h, w, 36))
            oc += cls_sizes[l]; orr += reg_sizes[l]
            self.cls_out._count(acts["cls"][l][4], 1); self.reg_out._count(acts["reg"][l][4], 1)
        ops.conv_group([acts["cls"][l][4] for l in range(5)], self.cls_out._packed(), pad=1, outs=cls_views, out_ld=self.cls_ld)
        ops.conv_group([acts["reg"][l][4] for l in range(5)], self.reg_out._packed(), pad=1, outs=reg_views, out_ld=36)
        mark("fpn+heads")
        self.last = dict(N=N, P=P, p6_relu=p6_relu, feats=feats, inner=inner, acts=acts, level_hw=level_hw, level_pix=level_pix, cls_flat=cls_flat, reg_flat=reg_flat,
                         cls_sizes=cls_sizes, reg_sizes=reg_sizes, matched=matched_dev, gt_labels=gt_labels_dev, gt_off=gt_off_dev, img_w=img_w, box_idx=box_idx,
                         box_w=box_w, reg_tgt=reg_tgt, matched_host=matched_all)
        losses = {"classification": ops.focal_loss(cls_flat, level_pix, N, 9, K, self.cls_ld, matched_dev, gt_labels_dev, gt_off_dev, img_w),
                  "bbox_regression": ops.smooth_l1(reg_flat, box_idx, reg_tgt, 0.0, 1.0, weights=box_w)}
        mark("losses")
        return losses

    def relu_decisions(self):
        L, out = self.last, self.body_relu_decisions()
        nchw = lambda t: (t > 0).permute(0, 3, 1, 2).cpu()
        for name in ("cls", "reg"):
            for l, xs in enumerate(L["acts"][name]):
                for j in range(4):
                    out["%s.%d.%d" % (name, l, j)] = nchw(xs[j + 1])
        out["p6"] = nchw(L["P"][3])
        return out

    def backward(self, gscale=(1.0, 1.0)):
        L = self.last
        N, P, level_hw = L["N"], L["P"], L["level_hw"]
        gcls = torch.zeros_like(L["cls_flat"]); greg = torch.zeros_like(L["reg_flat"])
        ops.focal_loss(L["cls_flat"], L["level_pix"], N, 9, self.K, self.cls_ld, L["matched"], L["gt_labels"], L["gt_off"], L["img_w"], grad=gcls, gscale=gscale[0])
        ops.smooth_l1(L["reg_flat"], L["box_idx"], L["reg_tgt"], 0.0, 1.0, weights=L["box_w"], grad=greg, gscale=gscale[1])
        # heads: weight gradients level by level (shared weights accumulate; they run on the side stream), data gradients of one layer of
        # BOTH towers on all five levels in one grouped launch
        oc = orr = 0
        g_cls, g_reg = [], []
        for l, (h, w) in enumerate(level_hw):
            g_cls.append(gcls[oc:oc + L["cls_sizes"][l]].view(N, h, w, self.cls_ld)); g_reg.append(greg[orr:orr + L["reg_sizes"][l]].view(N, h, w, 36))
            oc += L["cls_sizes"][l]; orr += L["reg_sizes"][l]
        acts = L["acts"]
        for l in range(5):
            self.cls_out.bwd(g_cls[l], need_dx=False, accumulate=l > 0, x=acts["cls"][l][4])
            self.reg_out.bwd(g_reg[l], need_dx=False, accumulate=l > 0, x=acts["reg"][l][4])
            self.cls_out._count(acts["cls"][l][4], 1); self.reg_out._count(acts["reg"][l][4], 1)
        # every ReLU backward of the towers rides in the epilogue of the data gradient that produces its input gradient
        g_cls = ops.conv_group(g_cls, self.cls_out._packed_grad(), pad=1, masks=[acts["cls"][l][4] for l in range(5)])
        g_reg = ops.conv_group(g_reg, self.reg_out._packed_grad(), pad=1)              # 36 channels: not a tiled-kernel shape
        for l in range(5):
            ops.relu_bwd_(g_reg[l], acts["reg"][l][4])
        for j in (3, 2, 1, 0):
            for l in range(5):
                self.cls_tower[j].bwd(g_cls[l], need_dx=False, accumulate=l > 0, x=acts["cls"][l][j])
                self.reg_tower[j].bwd(g_reg[l], need_dx=False, accumulate=l > 0, x=acts["reg"][l][j])
                self.cls_tower[j]._count(acts["cls"][l][j], 1); self.reg_tower[j]._count(acts["reg"][l][j], 1)
            masks = ([acts["cls"][l][j] for l in range(5)] + [acts["reg"][l][j] for l in range(5)]) if j > 0 else None
            gs = ops.conv_group(g_cls + g_reg, [self.cls_tower[j]._packed_grad()] * 5 + [self.reg_tower[j]._packed_grad()] * 5, pad=1, masks=masks)
            g_cls, g_reg = gs[:5], gs[5:]
        gP = [ops.add(g_cls[l], g_reg[l]) for l in range(5)]
        g6 = self.p7.bwd(gP[4], x=L["p6_relu"])
        ops.relu_bwd_(g6, P[3])
        g6 = ops.add(g6, gP[3])
        gP[2] = ops.add(gP[2], self.p6.bwd(g6, x=P[2]))
        ginner = self._fpn_out_bwd(gP[:3])
        for i in range(1, 3):
            ops.upsample_bwd_(ginner[i - 1], ginner[i])
        gC = [None] * 4
        for i in range(3):
            gC[i + 1] = self.lat[i].bwd(ginner[i], need_dx=self.layers[i + 1][0])
        self._body_backward(gC)
        self._join_side()
        return self.grads


class _LossFn(torch.autograd.Function):
    """Bridges the hand-written backward into torch autograd so that the reference's ``losses.backward()`` works unchanged."""

    @staticmethod
    def forward(ctx, anchor, net, images, targets):
        ctx.net = net
        d = net.forward(images, targets)
        # every forward of the model -- with or without grad -- replaces the layers' ONE set of saved activations, so it bumps the serial:
        # a loss evaluation under no_grad between a forward and its backward invalidates that backward too
        net.forward_serial = ctx.serial = getattr(net, "forward_serial", 0) + 1
        return tuple(d[k].reshape(()) for k in net.LOSS_NAMES)

    @staticmethod
    def backward(ctx, *gs):
        net = ctx.net
        if net.forward_serial != ctx.serial:
            # the layers keep ONE set of saved activations (no per-call graph): differentiating an older forward would silently
            # use the newer one's -- refuse instead
            raise RuntimeError("backward() of a training forward after a later forward of the same model: its saved activations "
                               "were replaced; call backward() before the next model(images, targets)")
        gscale = [0.0 if g is None else float(g) for g in gs]
        # autograd semantics: a parameter whose .grad is set (no zero_grad since the last backward, or zero_grad(set_to_none=False))
        # accumulates.  The .grad tensors ARE views of the flat gradient buffer, so accumulation happens inside the kernels.
        mine = [p.grad is not None and p.grad.data_ptr() == net.grads[k].data_ptr() for k, p in net.params.items()]
        foreign = [k for k, p in net.params.items() if p.grad is not None and p.grad.data_ptr() != net.grads[k].data_ptr()]
        if foreign:
            raise RuntimeError("parameter .grad was replaced by another tensor (%s ...): use zero_grad() between steps" % foreign[0])
        if any(mine) and not all(mine):
            raise RuntimeError("some parameters carry a gradient and some do not: call zero_grad() on all of them")
        net.accumulate_grads = all(mine)
        try:
            net.backward(gscale)                    # net.last IS this forward's record (the serial check above)
        finally:
            net.accumulate_grads = False
        for k in net.names:
            if net.params[k].grad is None:
                net.params[k].grad = net.grads[k]
        return None, None, None, None


class TrainableDetector(object):
    """``task_model`` in train mode: ``model(images, targets) -> {loss name: scalar tensor}`` whose sum can be ``.backward()``-ed."""

    def __init__(self, net):
        self.net = net
        self._anchor = torch.zeros(1, device=net.dev, requires_grad=True)
        self.training = True

    def parameters(self):
        return self.net.parameters()

    def __call__(self, images, targets):
        self.net.grad_wanted = torch.is_grad_enabled()
        out = _LossFn.apply(self._anchor, self.net, images, targets)
        return dict(zip(self.net.LOSS_NAMES, out))


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD(params, lr, momentum, weight_decay) (cald_train.py:397) with the update done by HIP kernels: ONE launch over the
    trainer's flat parameter / gradient buffers when the optimizer holds exactly the trainer's parameters in one group (the reference's
    setup), one launch per tensor otherwise.  ``param_groups[i]['lr']`` is read every step, so torch's lr schedulers (warmup LambdaLR,
    MultiStepLR) drive it unchanged; ``state[p]['momentum_buffer']`` are views of one flat momentum buffer in the fused case."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, net=None):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.net = net
        self._mflat = None

    def _fused_ok(self):
        net = self.net
        if net is None or len(self.param_groups) != 1:
            return False
        ps = self.param_groups[0]["params"]
        if len(ps) != len(net.names):
            return False
        for p, k in zip(ps, net.names):
            if p is not net.params[k] or p.grad is None or p.grad.data_ptr() != net.grads[k].data_ptr():
                return False
        started = ["momentum_buffer" in self.state[p] for p in ps]
        if any(started) and not all(started):
            return False                                   # a partial first step: per-tensor path
        if all(started) and ps:
            self._adopt_momentum(ps)
        return True

    @torch.no_grad()
    def _adopt_momentum(self, ps):
        """Every parameter has a momentum buffer.  After ``load_state_dict()`` (resume, cald_train.py:356-360) those are fresh
        tensors, not views of the flat buffer the fused launch updates: copy them in and re-point ``state`` at the views, so the
        loaded momentum is what the next step uses (and what ``state_dict()`` saves afterwards)."""
        net = self.net
        views_ok = self._mflat is not None
        if views_ok:
            base = self._mflat.data_ptr()
            for p, k in zip(ps, net.names):
                if self.state[p]["momentum_buffer"].data_ptr() != base + 4 * net._off[k]:
                    views_ok = False
                    break
        if views_ok:
            return
        flat = torch.zeros_like(net.flat)
        for p, k in zip(ps, net.names):
            view = flat[net._off[k]:net._off[k] + p.numel()].view(p.shape)
            view.copy_(self.state[p]["momentum_buffer"].to(view.device, torch.float32))
            self.state[p]["momentum_buffer"] = view
        self._mflat = flat

    @torch.no_grad()
    def step(self, closure=None):
        if self._fused_ok():
            grp, net = self.param_groups[0], self.net
            first = self._mflat is None
            if first and grp["momentum"] != 0:
                self._mflat = torch.zeros_like(net.flat)
                for k in net.names:
                    p = net.params[k]
                    self.state[p]["momentum_buffer"] = self._mflat[net._off[k]:net._off[k] + p.numel()].view(p.shape)
            # the pad words between tensors hold 0 in all three buffers and stay 0 under the update
            ops.sgd_(net.flat, net.gflat, self._mflat, grp["lr"], grp["momentum"], grp["weight_decay"], first)
            net.parameters_changed()
            return
        for grp in self.param_groups:
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first and grp["momentum"] != 0:
                    st["momentum_buffer"] = torch.zeros_like(p)
                ops.sgd_(p.data, p.grad, st.get("momentum_buffer"), grp["lr"], grp["momentum"], grp["weight_decay"], first)
        if self.net is not None:
            self.net.parameters_changed()


TrainableFasterRCNN = TrainableDetector
