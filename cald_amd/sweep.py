"""get_uncertainty / cls_kldiv with the reference's signatures (cald_train.py:91-231, :234-271).

``get_uncertainty(task_model, unlabeled_loader, augs, num_cls, bp=1.3)`` returns
``(consistency_all: list[float], cls_all: list[np.ndarray float64 [num_cls-1]])`` in loader order.
``bp`` replaces the module-global ``args.bp`` (cald_train.py:220).  With ``world_size > 1`` each rank
scores the strided shard ``pos % world_size == rank`` and one RCCL all-gather returns the full vectors
on every rank.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi

KNOWN_AUGS = ['flip', 'multi_ga', 'color_adjust', 'color_swap', 'multi_color_adjust', 'multi_sp', 'cut_out',
              'multi_cut_out', 'multi_resize', 'larger_resize', 'smaller_resize', 'rotation', 'ga', 'sp']
SUPPORTED_AUGS = tuple(a for a in KNOWN_AUGS if a != 'multi_color_adjust')


def expand_augs(augs):
    """The augmented views get_uncertainty builds for a list of aug names, in ITS order (cald_train.py:123-183 tests
    `name in augs` in a fixed sequence).  Returns [(kind, param)] for cald_sweep_cfg.augs."""
    F = _ffi
    out = []
    if 'flip' in augs:
        out.append((F.AUG_FLIP, 0.0))
    if 'ga' in augs:
        out.append((F.AUG_GAUSS, 16))
    if 'multi_ga' in augs:
        out += [(F.AUG_GAUSS, i * 8) for i in range(1, 7)]
    if 'color_adjust' in augs:
        out.append((F.AUG_COLOR_ADJUST, 1.5))
    if 'color_swap' in augs:
        out.append((F.AUG_COLOR_SWAP, 0.0))
    if 'multi_color_adjust' in augs:
        # cald_train.py:148 appends an undefined name: the reference raises NameError on the first image with detections
        raise NameError("name 'reference_boxes' is not defined (multi_color_adjust is broken in the reference, cald_train.py:148)")
    if 'sp' in augs:
        out.append((F.AUG_SALT_PEPPER, 0.1))
    if 'multi_sp' in augs:
        out += [(F.AUG_SALT_PEPPER, i * 0.05) for i in range(1, 7)]
    if 'cut_out' in augs:
        out.append((F.AUG_CUTOUT, 2))
    if 'multi_cut_out' in augs:
        out += [(F.AUG_CUTOUT, i) for i in range(1, 5)]
    if 'multi_resize' in augs:
        out += [(F.AUG_RESIZE, i * 0.1) for i in range(7, 10)]
    if 'larger_resize' in augs:
        out.append((F.AUG_RESIZE, 1.2))
    if 'smaller_resize' in augs:
        out.append((F.AUG_RESIZE, 0.8))
    if 'rotation' in augs:
        out.append((F.AUG_ROTATE, 5))
    return out


def make_sweep_cfg(augs, bp=1.3, base_seed=0, batch_images=64):
    specs = expand_augs(augs)
    if len(specs) > _ffi.MAX_AUGS:
        raise ValueError("more than %d augmented views per image" % _ffi.MAX_AUGS)
    cfg = _ffi.SweepCfg()
    cfg.base_seed, cfg.bp, cfg.batch_images, cfg.n_augs = int(base_seed), float(bp), int(batch_images), len(specs)
    for k, (kind, param) in enumerate(specs):
        cfg.augs[k].kind, cfg.augs[k].param = kind, float(param)
    return cfg


def _to_u8_cuda(image, device):
    if isinstance(image, torch.Tensor):
        t = image
        if t.dtype != torch.uint8:
            t = (t * 255.0).round().clamp(0, 255).to(torch.uint8)
        if t.shape[0] == 3 and t.shape[-1] != 3:
            t = t.permute(1, 2, 0)
    else:  # PIL image or ndarray
        t = torch.from_numpy(np.array(image, dtype=np.uint8))
    return t.contiguous().to(device, non_blocking=True)


def sweep_device_images(task_model, images, positions, augs, bp=1.3, base_seed=0, batch_images=64, margins=False):
    """Scores uint8 HWC CUDA tensors already resident in HBM.  Returns (consistency [n] f64, cls_corr [n][C-1] f64); with
    margins=True also the decision-margin records [n][N_MARGINS] float32 of cald_sweep_audit (include/cald_hip.h)."""
    for aug in augs:
        if aug not in KNOWN_AUGS:
            print('{} is not in the pre-set augmentations!'.format(aug))   # cald_train.py:92-95
    L = _ffi.lib()
    n = len(images)
    Cn = task_model.num_classes
    cons = np.zeros(n, np.float64)
    cls = np.zeros((n, Cn - 1), np.float64)
    mg = np.full((n, _ffi.N_MARGINS), np.inf, np.float32) if margins else None
    if n == 0:
        return (cons, cls, mg) if margins else (cons, cls)
    ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in images])
    Hs = np.array([im.shape[0] for im in images], np.int32)
    Ws = np.array([im.shape[1] for im in images], np.int32)
    pos = np.ascontiguousarray(positions, dtype=np.int64)
    cfg = make_sweep_cfg(augs, bp, base_seed, batch_images)
    if margins:
        _ffi.check(L.cald_sweep_audit(task_model.handle(), n, ptrs, _ffi.ptr(Hs, _ffi.c_i), _ffi.ptr(Ws, _ffi.c_i), _ffi.ptr(pos, _ffi.c_i64),
                                      C.byref(cfg), _ffi.ptr(cons, _ffi.c_d), _ffi.ptr(cls, _ffi.c_d), _ffi.ptr(mg, _ffi.c_f)))
        return cons, cls, mg
    _ffi.check(L.cald_sweep(task_model.handle(), n, ptrs, _ffi.ptr(Hs, _ffi.c_i), _ffi.ptr(Ws, _ffi.c_i),
                            _ffi.ptr(pos, _ffi.c_i64), C.byref(cfg), _ffi.ptr(cons, _ffi.c_d), _ffi.ptr(cls, _ffi.c_d)))
    return cons, cls


def shard_positions(pool_size, rank, world_size):
    """Pool positions rank `rank` scores: the strided shard p % world_size == rank."""
    return list(range(rank, pool_size, world_size))


def shard_subset(subset, rank, world_size):
    """The slice of cald_train.py's `subset` (the shuffled unlabeled indices, :427-431) that rank `rank` feeds its own
    loader with: ``DataLoader(dataset_aug, batch_size=1, sampler=SubsetSequentialSampler(shard_subset(subset, r, W)))``
    (:434, ll4al/data/sampler.py:3-16).  Every rank then decodes only its own 1/W of the pool; pass
    ``loader_is_sharded=True`` to get_uncertainty() and the gathered results come back in `subset` order."""
    return list(subset[rank::world_size])


class ShardedSequentialSampler:
    """ll4al/data/sampler.py:3-16 SubsetSequentialSampler restricted to one rank's strided shard."""

    def __init__(self, indices, rank, world_size):
        self.indices = shard_subset(indices, rank, world_size)

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


def allgather_scores(local_pos, cons, cls, pool_size, group=None):
    """One all-gather of the per-image score rows (consistency, cls_corr[C-1]) (RCCL on GPUs, gloo on CPU); replaces
    detection/utils.py:75-115's pickled all_gather pattern.  Every rank holds the strided shard p % world == rank in
    ascending position order, so row j of rank r IS position r + j * world: no index column travels, and shards are
    padded to ceil(pool / world) rows.  Returns the full (consistency [pool], cls_corr [pool][C-1]) on every rank."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    Cm1 = cls.shape[1]
    rows = (pool_size + world - 1) // world
    mine = shard_positions(pool_size, rank, world)
    if list(local_pos) != mine:
        raise ValueError("rank %d must hold exactly the strided shard p %% %d == %d of a pool of %d, in order" % (rank, world, rank, pool_size))
    buf = torch.zeros((rows, 1 + Cm1), dtype=torch.float64)
    k = len(mine)
    if k:
        buf[:k, 0] = torch.from_numpy(np.ascontiguousarray(cons, dtype=np.float64))
        buf[:k, 1:] = torch.from_numpy(np.ascontiguousarray(cls, dtype=np.float64))
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = buf.to(dev)
    out = torch.empty((world * rows, 1 + Cm1), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy().reshape(world, rows, 1 + Cm1)
    full_cons = np.zeros(pool_size, np.float64)
    full_cls = np.zeros((pool_size, Cm1), np.float64)
    for r in range(world):
        pos = shard_positions(pool_size, r, world)
        full_cons[pos] = out[r, :len(pos), 0]
        full_cls[pos] = out[r, :len(pos), 1:]
    return full_cons, full_cls


def get_uncertainty(task_model, unlabeled_loader, augs, num_cls, bp=1.3, base_seed=0, rank=0, world_size=1,
                    batch_images=64, group=None, chunk_images=4096, loader_is_sharded=False):
    """Drop-in for cald_train.py:91 (same positional signature).  Images are uploaded and swept ``chunk_images`` at a
    time, so a loader-fed sweep never holds more than that many decoded images in HBM (a DevicePool holds them all by
    design); results do not depend on the chunking.

    Multi-GPU (one process per GPU): rank r scores pool positions p % world_size == r and one
    all-gather returns the full vectors, in pool (= single-process loader) order, on every rank.
      loader_is_sharded=False  every rank is handed the WHOLE loader and skips foreign items (they are still pulled,
                               i.e. decoded, by the loader -- fine for a DevicePool, wasteful for a DataLoader);
      loader_is_sharded=True   the loader yields ONLY this rank's items, item i being pool position rank + i*world_size
                               (build it from shard_subset(subset, rank, world_size) / ShardedSequentialSampler /
                               DevicePool.loader(subset, rank, world_size)): no rank touches another rank's images."""
    if not hasattr(task_model, "handle"):          # the reference's torch model: mirror it on the HIP side
        from .detector import from_torch_module
        task_model = from_torch_module(task_model)
    task_model.eval()
    # without a GPU the upload below is a no-op and sweep_device_images() raises (no CPU fallback)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    images, positions, all_pos, cons_parts, cls_parts = [], [], [], [], []
    n_items = 0

    def flush():
        if images:
            c, k = sweep_device_images(task_model, images, positions, augs, bp, base_seed, batch_images)
            cons_parts.append(c); cls_parts.append(k); all_pos.extend(positions)
            images.clear(); positions.clear()

    for i, (imgs, _) in enumerate(unlabeled_loader):      # batch size 1, cald_train.py:101-104
        n_items += 1
        if loader_is_sharded:
            pos = rank + i * world_size
        else:
            pos = i
            if pos % world_size != rank:
                continue
        for image in imgs:
            images.append(_to_u8_cuda(image, dev)); positions.append(pos)
        if len(images) >= chunk_images:
            flush()
    flush()
    cons = np.concatenate(cons_parts) if cons_parts else np.zeros(0, np.float64)
    cls = np.concatenate(cls_parts) if cls_parts else np.zeros((0, num_cls - 1), np.float64)
    if world_size > 1:
        pool_size = n_items
        if loader_is_sharded:      # the pool size is the sum of the shard lengths: one tiny all-reduce
            import torch.distributed as dist
            backend = dist.get_backend(group)
            t = torch.tensor([n_items], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, group=group)
            pool_size = int(t.item())
        cons, cls = allgather_scores(all_pos, cons, cls, pool_size, group)
    return [float(c) for c in cons], [cls[i] for i in range(cls.shape[0])]


def _labeled_class_histogram(labeled_loader, width):
    """Mean per-image class count vector of the labeled set (cald_train.py:237-242, :253); labels are 1-based."""
    rows = []
    for _, targets in labeled_loader:
        for target in targets:
            lab = np.asarray([int(l) for l in target['labels']], dtype=np.int64) - 1
            lab[lab < 0] += width                     # Python's negative indexing (label 0 lands in the last slot)
            rows.append(np.bincount(lab, minlength=width)[:width])
    return np.mean(np.array(rows), axis=0)


def cls_kldiv(labeled_loader, cls_corrs, budget, cycle=0, uniform=False):
    """Second-stage class-balance pick of cald_train.py:234-271 (host side, float64; `uniform` replaces the global
    args.uniform).  Candidates whose cls_corr is all zero come first (:246-247, possibly more than `budget` of them).
    The reference then recomputes the same JS vector in every iteration (its `result` is never updated, :270 is
    commented out) and takes the arg-max of the not-yet-picked entries -- i.e. it walks the candidates in descending
    JS order, first index first on ties (ascending for --uniform).  Here the vector is computed ONCE, with the
    reference's own torch float64 operations so the order is index-exact, and walked."""
    corr = np.asarray(cls_corrs, dtype=np.float64)
    picked = [int(i) for i in np.flatnonzero(corr.sum(axis=1) == 0)]
    if len(picked) >= budget:
        return picked
    hist = torch.from_numpy(_labeled_class_histogram(labeled_loader, corr.shape[1])).unsqueeze(0)
    cand = torch.from_numpy(corr)
    if uniform:
        p = torch.softmax(hist + cand, -1)
        q = torch.softmax(torch.ones(hist.shape) / len(hist), -1)
    else:
        p = torch.softmax(hist, -1)
        q = torch.softmax(cand, -1)
    log_mean = ((p + q) / 2).log()
    kl = torch.nn.functional.kl_div
    js = (kl(log_mean, p, reduction='none').sum(dim=1) / 2 + kl(log_mean, q, reduction='none').sum(dim=1) / 2).numpy()
    free = np.ones(len(js), dtype=bool)
    free[picked] = False
    order = np.argsort(js if uniform else -js, kind='stable')
    picked += [int(i) for i in order if free[i]][:budget - len(picked)]
    # budget larger than the candidate list: every entry is masked, so the reference's arg-max/arg-min returns 0 from then on
    picked += [0] * (budget - len(picked))
    return picked


def select(uncertainty, cls_corrs, labeled_loader, budget, mr=1.2, mutual=True, uniform=False):
    """cald_train.py:439-447: argsort ascending, candidate cut, class-balance pick.  Returns positions into the pool."""
    arg = np.argsort(np.asarray(uncertainty))
    if not mutual:
        return arg[:budget]
    cand = arg[:int(mr * budget)]
    tobe = cls_kldiv(labeled_loader, [cls_corrs[i] for i in cand], budget, uniform=uniform)
    return cand[np.asarray(tobe, dtype=np.int64)]
