"""Evaluation-side consumer of the same detector forward.

``voc_detections`` is the forward half of detection/engine.py:86-141 (``voc_evaluate``): it walks the test
loader, runs the HIP detector -- batched, up to 64 images per launch sequence instead of the reference's
batch-1 loop with a ``torch.cuda.synchronize()`` per image -- and returns ``(all_boxes, image_index)`` in the
reference's structure.  ``write_voc_results_file`` emits the wire format of detection/voc_eval.py:188-222
(``/tmp/{path}/det_test_{cls}.txt``: ``image_id score x1+1 y1+1 x2+1 y2+1``), which the reference's own
``_do_python_eval`` consumes; ``cald_amd.voc_eval`` is that consumer (AP over IoU .5:.95, same numbers, annotations
parsed once), and ``voc_evaluate`` below is the whole of detection/engine.py:86-158 with the reference's signature.
"""
import os
import shutil

import torch


def warmup_lr_scheduler(optimizer, warmup_iters, warmup_factor):
    """detection/utils.py:239-247."""
    def f(x):
        if x >= warmup_iters:
            return 1
        alpha = float(x) / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)


def train_one_epoch(task_model, task_optimizer, data_loader, device, cycle, epoch, print_freq):
    """cald_train.py:40-74 with the HIP detector in train mode: warmup in epoch 0, per batch ``loss_dict = model(images, targets)``,
    non-finite loss stops the run (sys.exit(1), :62-65), zero_grad / backward / step / scheduler step.  Returns the list of
    per-iteration summed losses (the reference returns its MetricLogger).  Single process (no reduce_dict)."""
    import math
    import sys
    task_model.train()
    sched = None
    if epoch == 0:
        warmup_iters = min(1000, len(data_loader) - 1)
        if warmup_iters > 0:
            sched = warmup_lr_scheduler(task_optimizer, warmup_iters, 1. / 1000)
    history = []
    for i, (images, targets) in enumerate(data_loader):
        images = list(images)
        targets = [dict(t) for t in targets]
        loss_dict = task_model(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        value = float(losses.detach())
        if not math.isfinite(value):
            print("Loss is {}, stopping training".format(value))
            print({k: float(v.detach()) for k, v in loss_dict.items()})
            sys.exit(1)
        task_optimizer.zero_grad()
        losses.backward()
        task_optimizer.step()
        if sched is not None:
            sched.step()
        history.append(value)
        if print_freq and i % print_freq == 0:
            print("Cycle:[{}] Epoch: [{}]  [{}/{}]  task_loss: {:.4f}  task_lr: {:.6f}".format(
                cycle, epoch, i, len(data_loader), value, task_optimizer.param_groups[0]["lr"]))
    return history


def voc_detections(model, data_loader, num_classes=21, batch_views=64):
    model.eval()
    all_boxes = [[] for _ in range(num_classes)]
    image_index = []
    pending = []

    def flush():
        if not pending:
            return
        outs = model([im for im, _ in pending])
        for (_, name), o in zip(pending, outs):
            image_index.append(name)
            o = {k: v.cpu() for k, v in o.items()}
            per_cls = [[] for _ in range(num_classes)]
            for i in range(o['boxes'].shape[0]):
                per_cls[int(o['labels'][i])].append(torch.cat([o['boxes'][i], o['scores'][i].unsqueeze(0)], dim=0))
            for c in range(num_classes):
                all_boxes[c].append([torch.stack(per_cls[c])] if per_cls[c] else [])
        pending.clear()

    for images, targets in data_loader:
        for img, t in zip(images, targets):
            name = ''.join(chr(int(i)) for i in t['name'].tolist()) if 'name' in t else str(len(image_index) + len(pending))
            pending.append((img, name))
            if len(pending) >= batch_views:
                flush()
    flush()
    return all_boxes, image_index


def write_voc_results_file(all_boxes, image_index, path, classes, root='/tmp'):
    out_dir = os.path.join(root, path)
    if os.path.exists(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    all_boxes = [list(b) for b in all_boxes]
    for cls_ind, cls in enumerate(classes):
        pairs = sorted(zip(image_index, all_boxes[cls_ind]), key=lambda x: x[0])
        if cls == '__background__':
            continue
        with open(os.path.join(out_dir, 'det_test_{:s}.txt'.format(cls)), 'wt') as f:
            prev = ''
            for index, dets in pairs:
                if prev == index:       # repeated inputs (DistributedSampler padding) are discarded
                    continue
                prev = index
                if dets == []:
                    continue
                dets = dets[0]
                for k in range(dets.shape[0]):
                    f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(
                        index, float(dets[k, -1]), float(dets[k, 0]) + 1, float(dets[k, 1]) + 1,
                        float(dets[k, 2]) + 1, float(dets[k, 3]) + 1))
    return out_dir


def voc_evaluate(model, data_loader, year, feature=False, path='results', root='/tmp', batch_views=64):
    """detection/engine.py:86-158: forward over the test loader (batched on the GPU), gather over ranks, results files,
    AP table.  Returns what ``cald_amd.voc_eval.do_python_eval`` returns on the main process, None elsewhere."""
    import torch.distributed as dist
    from .voc_eval import do_python_eval
    if feature:
        raise NotImplementedError("feature=True (models returning (features, outputs)) belongs to the LL4AL baselines")
    classes = data_loader.dataset._transforms.transforms[0].CLASSES
    all_boxes, image_index = voc_detections(model, data_loader, len(classes), batch_views)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:     # utils.all_gather (:143-144)
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (all_boxes, image_index))
        if dist.get_rank() != 0:
            return None
        all_boxes = [sum((p[0][c] for p in parts), []) for c in range(len(classes))]
        image_index = sum((p[1] for p in parts), [])
    write_voc_results_file(all_boxes, image_index, path, classes, root=root)
    return do_python_eval(data_loader, year, path, root=root)


def coco_predictions(model, data_loader, batch_views=64):
    """Forward half of detection/engine.py:178-205 (``coco_evaluate``): ``{image_id: output dict on the CPU}`` for the
    whole loader, which is what ``CocoEvaluator.update`` receives there one image at a time.  Batched on the GPU."""
    model.eval()
    res, pending = {}, []

    def flush():
        if not pending:
            return
        outs = model([im for im, _ in pending])
        for (_, image_id), o in zip(pending, outs):
            res[image_id] = {k: v.cpu() for k, v in o.items() if k != 'features'}
        pending.clear()

    for images, targets in data_loader:
        for img, t in zip(images, targets):
            image_id = t["image_id"].item() if hasattr(t["image_id"], "item") else t["image_id"]
            pending.append((img, image_id))
            if len(pending) >= batch_views:
                flush()
    flush()
    return res


def coco_results(predictions):
    """detection/coco_eval.py:76-98 ``prepare_for_coco_detection`` (+ ``convert_to_xywh`` :162-164): the list of
    ``{"image_id", "category_id", "bbox": [x, y, w, h], "score"}`` records pycocotools' ``loadRes`` consumes."""
    out = []
    for original_id, prediction in predictions.items():
        if len(prediction) == 0:
            continue
        xmin, ymin, xmax, ymax = prediction["boxes"].unbind(1)
        boxes = torch.stack((xmin, ymin, xmax - xmin, ymax - ymin), dim=1).tolist()
        scores = prediction["scores"].tolist()
        labels = prediction["labels"].tolist()
        out.extend({"image_id": original_id, "category_id": labels[k], "bbox": box, "score": scores[k]}
                   for k, box in enumerate(boxes))
    return out


def coco_evaluate(model, data_loader, classwise=True, feature=False, batch_views=64):
    """detection/engine.py:178-256: forward over the test loader (batched on the GPU), CocoEvaluator update / accumulate / summarize
    (cald_amd.coco_eval: bbox AP without pycocotools, parity unpinned), optional per-category AP table.  Returns the evaluator."""
    from .coco_eval import CocoEvaluator, CocoGT
    if feature:
        raise NotImplementedError("feature=True (models returning (features, outputs)) belongs to the LL4AL baselines")
    model.eval()
    coco = CocoGT.from_dataset(data_loader.dataset)
    evaluator = CocoEvaluator(coco, ["bbox"])
    evaluator.update(coco_predictions(model, data_loader, batch_views))
    evaluator.synchronize_between_processes()
    evaluator.accumulate()
    evaluator.summarize()
    if classwise:                                             # per-category AP (engine.py:225-254)
        import numpy as np
        prec = evaluator.coco_eval["bbox"].eval["precision"]
        rows = []
        for idx, cat_id in enumerate(coco.get_cat_ids()):
            p = prec[:, :, idx, 0, -1]
            p = p[p > -1]
            rows.append((str(coco.cats[cat_id].get("name", cat_id)), "%0.3f" % (float(np.mean(p)) if p.size else float("nan"))))
        width = max(len(r[0]) for r in rows) if rows else 8
        print("\n" + "\n".join("| %-*s | %s |" % (width, n, a) for n, a in [("category", "AP")] + rows))
    return evaluator
