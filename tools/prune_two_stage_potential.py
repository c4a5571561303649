"""How much would a two-stage selection save?  Stage 1 computes the exact head only at pixels holding an anchor whose LOWER bound reaches tau
(at least k anchors: they fix a better threshold tau' = k-th largest exact logit among them), stage 2 at the remaining pixels with an upper
bound >= tau' instead of >= tau.  Prints the selected pixel fractions of the current rule and of the two-stage rule on full-size views, from
the capture hooks (look-ahead logits, |patch|_2, dense logits).   python tools/prune_two_stage_potential.py [n_images]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cald_amd import detector, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for tag, depth, shape, ncls, mn, mx in (("R50 VOC", 50, "voc", 21, 600, 1000), ("R101 COCO", 101, "coco", 91, 800, 1333)):
        sd = synth.pseudo_trained_frcnn(ncls, depth, seed=0 if depth == 50 else 1)
        make = detector.fasterrcnn_resnet101_fpn_feature if depth == 101 else detector.fasterrcnn_resnet50_fpn_feature
        m = make(num_classes=ncls, min_size=mn, max_size=mx).to("cuda"); m.load_state_dict(sd); m.eval()
        c1, c0 = m.rpn_prune_bound()
        views = [(torch.from_numpy(im).cuda(), bool(i & 1), None) for i, im in enumerate(synth.make_pool(n, shape, 0))]
        m.set_rpn_prune_capture(True); m.forward_views(views)
        cap = [{k: m.debug_tensor(k, v) for k in ("rpn_look0", "rpn_look1", "rpn_pnorm0", "rpn_pnorm1")} for v in range(n)]
        m.set_rpn_prune_capture(False); m.forward_views(views)
        tot = np.zeros((2, 4))
        for v in range(n):
            for l in range(2):
                dense = m.debug_tensor("rpn%d" % l, v)[:, :, :3].astype(np.float64)
                look = cap[v]["rpn_look%d" % l][:, :, :3].astype(np.float64); pn = cap[v]["rpn_pnorm%d" % l][:, :, 0].astype(np.float64)
                B = c1[None, None, :] * pn[:, :, None] + c0[None, None, :]
                lb, ub = look - B, look + B
                k = min(1000, lb.size)
                tau = np.sort(lb.reshape(-1))[-k]
                cur = (ub >= tau).any(-1)
                s1 = (lb >= tau).any(-1)
                ex = dense[s1].reshape(-1)
                tau2 = np.sort(ex)[-k] if ex.size >= k else -np.inf
                s2 = (ub >= max(tau, tau2)).any(-1) & ~s1
                tot[l] += [cur.mean(), s1.mean(), (s1 | s2).mean(), 1]
        for l in range(2):
            print("%s P%d: current rule %.4f of the pixels; two-stage: stage 1 %.4f, total %.4f" % (tag, l + 2, tot[l, 0] / tot[l, 3], tot[l, 1] / tot[l, 3], tot[l, 2] / tot[l, 3]))
        del m; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
