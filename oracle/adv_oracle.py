"""CPU oracle for AdvSSL: FC discriminator, its criterion, the sseg task hooks and the
two-phase SSLADV training iteration.

TEST INFRASTRUCTURE ONLY -- the checker, never the product (see torch_oracle.py for the rules).

Functional fp32 restatement of pixelssl/ssl_algorithm/ssl_adv.py and task/sseg/func.py:137-168; PINNED by
oracle/make_golden_adv.py, which runs the reference's own FCDiscriminator / FCDiscriminatorCriterion / TaskFunc hooks
and `SSLADV._train` on seeded inputs, asserts this file reproduces them and writes tests/golden/adv_65.pt.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

import torch_oracle as TO

FCD_LAYERS = (("conv1", None, 64), ("conv2", 64, 128), ("conv3", 128, 256), ("conv4", 256, 512),
              ("classifier", 512, 1))          # ssl_adv.py:466-476 (ndf = 64), all 4x4 / stride 2 / pad 1 with bias


def init_fcd_state(in_channels=21, seed=0):
    """torch's default nn.Conv2d initialisation in the reference's construction order (ssl_adv.py:469-473) under
    torch.manual_seed(seed): reproducible on any box with the same torch build (checked against the reference)."""
    torch.manual_seed(seed)
    sd = OrderedDict()
    for name, cin, cout in FCD_LAYERS:
        conv = nn.Conv2d(cin if cin is not None else in_channels, cout, kernel_size=4, stride=2, padding=1)
        sd[name + ".weight"] = conv.weight.detach().clone()
        sd[name + ".bias"] = conv.bias.detach().clone()
    return sd


def fcd_forward(sd, task_pred):
    """FCDiscriminator.forward (ssl_adv.py:477-493): 4 x (conv4x4 s2 + LeakyReLU 0.2), classifier conv, bilinear
    up-sampling (align_corners=True) to the input size; the confidence map is NOT activated."""
    x = task_pred
    for name, _, _ in FCD_LAYERS[:-1]:
        x = F.leaky_relu(F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], 2, 1), 0.2)
    x = F.conv2d(x, sd["classifier.weight"], sd["classifier.bias"], 2, 1)
    return F.interpolate(x, size=task_pred.shape[2:], mode="bilinear", align_corners=True)


def fcd_criterion(pred, gt):
    """FCDiscriminatorCriterion.forward (ssl_adv.py:500-503)."""
    return torch.mean(F.binary_cross_entropy_with_logits(pred, gt, reduction="none"), dim=(1, 2, 3))


def preprocess_fcd_criterion(fcd_pred, task_gt, is_real, ignore_index=255):
    """ssladv_preprocess_fcd_criterion (task/sseg/func.py:137-157): target = 1 (real) / 0 (fake); pixels whose task
    label is ignore_index are masked in BOTH the prediction and the target (and still count in the mean)."""
    biclass = 1.0 if is_real else 0.0
    if task_gt is None:
        mask = torch.ones_like(fcd_pred)
    else:
        mask = (task_gt != ignore_index).float()
    return fcd_pred * mask, torch.full_like(fcd_pred, biclass) * mask


def convert_task_gt_to_fcd_input(task_gt, num_classes=21):
    """ssladv_convert_task_gt_to_fcd_input (task/sseg/func.py:159-168): one-hot of the labels (ignored pixels match
    no class -> all zero)."""
    return torch.cat([(task_gt == i).float() for i in range(num_classes)], dim=1)


class AdvOracleTrainer(TO.OracleTrainer):
    """SSLADV._train body (ssl_adv.py:126-283), one iteration per call.  Extra hp: adv_for_labeled,
    labeled_adv_scale, unlabeled_adv_scale, unlabeled_for_discriminator, discriminator_scale, discriminator_lr,
    discriminator_power."""

    def __init__(self, state, d_state, hp):
        super().__init__(state, hp)
        self.hp.update(dict(adv_for_labeled=True, labeled_adv_scale=0.01, unlabeled_adv_scale=0.001,
                            unlabeled_for_discriminator=True, discriminator_scale=1.0, discriminator_lr=1e-4,
                            discriminator_power=0.9))
        self.hp.update(hp)
        self.d_sd = d_state
        self.d_leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in d_state.items())
        # optim.Adam(lr, betas=(0.9, 0.99)) (ssl_adv.py:101-102)
        self.d_opt = torch.optim.Adam(list(self.d_leaves.values()), lr=self.hp["discriminator_lr"], betas=(0.9, 0.99))

    def _d_lr(self):
        # PolynomialLR on the discriminator optimizer (ssl_adv.py:107-108); same initial-step quirk as the task lrer
        return TO.poly_lr(self.hp["discriminator_lr"], self.it + self.hp["lr_iter_offset"], self.hp["max_iters"],
                          self.hp["discriminator_power"])

    def adv_step(self, x, gt, lbs):
        hp = self.hp
        B = x.shape[0]
        # ---- step 1: task model (ssl_adv.py:137-194)
        leaves = TO._param_leaves(self.sd)
        run = TO._with_leaves(self.sd, leaves)
        logits, prob, _, _ = TO.deeplabv2_forward(run, x, train=True)
        for k in self.sd:
            if TO.is_buffer(k):
                self.sd[k] = run[k]
        conf = fcd_forward(self.d_leaves, prob)
        task_loss = TO.sseg_criterion(logits[:lbs], gt[:lbs], hp["ignore_index"]).mean()
        l_adv = torch.zeros(())
        if hp["adv_for_labeled"]:
            p, g = preprocess_fcd_criterion(conf[:lbs], gt[:lbs], True, hp["ignore_index"])
            l_adv = hp["labeled_adv_scale"] * fcd_criterion(p, g).mean()
        u_adv = torch.zeros(())
        if B > lbs:
            p, g = preprocess_fcd_criterion(conf[lbs:], None, True, hp["ignore_index"])
            u_adv = hp["unlabeled_adv_scale"] * fcd_criterion(p, g).mean()
        for v in self.d_leaves.values():
            v.grad = None
        (task_loss + l_adv + u_adv).backward()
        grads = OrderedDict((k, v.grad) for k, v in leaves.items())
        with torch.no_grad():
            TO.sgd_step(self.sd, grads, self.mom, self._lrs(), hp["momentum"], hp["weight_decay"])
        # ---- step 2: discriminator (ssl_adv.py:199-246)
        self.d_opt.zero_grad()
        fake_pred = prob.detach() if hp["unlabeled_for_discriminator"] else prob[:lbs].detach()
        fconf = fcd_forward(self.d_leaves, fake_pred)
        fp, fg = preprocess_fcd_criterion(fconf[:lbs], gt[:lbs], False, hp["ignore_index"])
        if hp["unlabeled_for_discriminator"] and B > lbs:
            up, ug = preprocess_fcd_criterion(fconf[lbs:], None, False, hp["ignore_index"])
            fp, fg = torch.cat((fp, up), 0), torch.cat((fg, ug), 0)
        fake_d = hp["discriminator_scale"] * fcd_criterion(fp, fg).mean()
        real_in = convert_task_gt_to_fcd_input(gt[:lbs])
        rp, rg = preprocess_fcd_criterion(fcd_forward(self.d_leaves, real_in), gt[:lbs], True, hp["ignore_index"])
        real_d = hp["discriminator_scale"] * fcd_criterion(rp, rg).mean()
        ((fake_d + real_d) / 2).backward()
        for gp in self.d_opt.param_groups:
            gp["lr"] = self._d_lr()
        self.d_opt.step()
        self.it += 1
        return dict(task_loss=float(task_loss.detach()), labeled_adv_loss=float(l_adv.detach()),
                    unlabeled_adv_loss=float(u_adv.detach()), fake_d_loss=float(fake_d.detach()),
                    real_d_loss=float(real_d.detach()))

    def d_state(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.d_leaves.items())
