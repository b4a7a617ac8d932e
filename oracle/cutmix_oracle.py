"""CPU oracle for CutMix.  TEST INFRASTRUCTURE ONLY (see torch_oracle.py).

Restates pixelssl/ssl_algorithm/ssl_cutmix.py: BoxMaskGenerator.produce (:481-547) for the configuration the
algorithm instantiates (:127-128: one box, area proportion, random aspect ratio, within bounds, inverted) and the
training iteration SSLCUTMIX._train (:140-227).  PINNED by oracle/make_golden_cutmix.py against the reference's own
classes (same numpy RNG stream -> same boxes) -> tests/golden/cutmix_65.pt.
"""
from collections import OrderedDict

import numpy as np
import torch

import torch_oracle as TO


def box_masks(mask_num, mask_shape, prop_range=(0.5, 0.5), rng=np.random):
    """BoxMaskGenerator(prop_range, boxes_num=1, random_aspect_ratio=True, area_prop=True, within_bounds=True,
    invert=True).produce (ssl_cutmix.py:494-546): draws, in this order, the box area proportion, the aspect split
    and the box position; the box is 1 inside, 0 outside (invert=True starts from zeros and flips the box)."""
    props = rng.uniform(prop_range[0], prop_range[1], size=(mask_num, 1))
    zero = props == 0.0
    y_props = np.exp(rng.uniform(low=0.0, high=1.0, size=(mask_num, 1)) * np.log(props))
    x_props = props / y_props
    y_props[zero] = 0
    x_props[zero] = 0
    shape = np.array(mask_shape)
    sizes = np.round(np.stack([y_props, x_props], axis=2) * shape[None, None, :])
    pos = np.round((shape - sizes) * rng.uniform(low=0.0, high=1.0, size=sizes.shape))
    masks = np.zeros((mask_num, 1) + tuple(mask_shape), dtype=np.float32)
    for i in range(mask_num):
        y0, x0 = pos[i, 0]
        y1, x1 = pos[i, 0] + sizes[i, 0]
        masks[i, 0, int(y0):int(y1), int(x0):int(x1)] = 1.0
    return masks


class CutMixOracleTrainer(TO.OracleTrainer):
    """SSLCUTMIX._train body (ssl_cutmix.py:140-227), one iteration per call.  hp: cons_scale, cons_rampup_iters,
    cons_threshold, ema_decay, mask_prop_range."""

    def cutmix_step(self, x, gt, lbs, rng=np.random):
        hp = self.hp
        ubs = x.shape[0] - lbs
        half = ubs // 2
        mask = torch.from_numpy(box_masks(half, tuple(x.shape[2:]), hp.get("mask_prop_range", (0.5, 0.5)), rng))
        mix_inp = mask * x[lbs:lbs + half] + (1 - mask) * x[lbs + half:]
        ramp = TO.sigmoid_rampup(self.it, hp["cons_rampup_iters"])
        leaves = TO._param_leaves(self.sd)
        run = TO._with_leaves(self.sd, leaves)
        # labeled samples through the student (BN statistics of this sub-batch only)
        l_logits, _, _, _ = TO.deeplabv2_forward(run, x[:lbs], train=True)
        task_loss = TO.sseg_criterion(l_logits, gt[:lbs], hp["ignore_index"]).mean()
        # unlabeled originals through the teacher (no grad, train-mode BN)
        with torch.no_grad():
            _, t_prob, _, _ = TO.deeplabv2_forward(self.t_sd, x[lbs:], train=True)
        mix_t = mask * t_prob[:half] + (1 - mask) * t_prob[half:]
        confidence = (mix_t.max(dim=1)[0] > hp["cons_threshold"]).float().mean()   # ONE scalar for the half batch
        _, s_prob, _, _ = TO.deeplabv2_forward(run, mix_inp, train=True)
        for k in self.sd:
            if TO.is_buffer(k):
                self.sd[k] = run[k]
        cons = ramp * hp["cons_scale"] * (TO.mse_loss(s_prob, mix_t) * confidence)
        (task_loss + cons).backward()
        grads = OrderedDict((k, v.grad) for k, v in leaves.items())
        with torch.no_grad():
            TO.sgd_step(self.sd, grads, self.mom, self._lrs(), hp["momentum"], hp["weight_decay"])
            TO.ema_update(self.t_sd, self.sd, hp["ema_decay"], self.it)
        self.it += 1
        return dict(task_loss=float(task_loss.detach()), cons_loss=float(cons.detach()),
                    confidence=float(confidence), mask_sum=float(mask.sum()))
