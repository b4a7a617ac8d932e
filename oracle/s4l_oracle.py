"""CPU restatement of PixelSSL's S4L (pixelssl/ssl_algorithm/ssl_s4l.py): rotation pretext task on top of a pixel-wise
task model.  TEST INFRASTRUCTURE ONLY (pinned against the real reference by oracle/make_golden_s4l.py; imported by
tests/ only).

  * RotationClassifer (ssl_s4l.py:371-393): conv 4x4/s2 (C -> C) + BatchNorm2d + LeakyReLU(0.2), conv 4x4/s2 (C -> 2C) +
    BatchNorm2d + LeakyReLU(0.2), global average pool, Linear(2C -> 4); input = the task model's `pred` (logits,
    task/sseg/model.py:63).
  * _batch_prehandle / _rotate_tensor (ssl_s4l.py:296-356): every sample gets one rotated copy (angle index 1..3 drawn
    with np.random.randint(1, 4, size=bs)), appended after the un-rotated batch; ground truths are rotated the same way;
    the last element of the gt tuple is the rotation class (0 for the un-rotated half).
  * _train (ssl_s4l.py:113-209): CE on the un-rotated labeled samples + rotated_sup_scale * CE on their rotated copies +
    rotation_scale * nn.CrossEntropyLoss()(rotation logits, rotation classes) over all 2*bs samples; ONE SGD over
    task_model.param_groups + [{rotation classifier, lr}] (ssl_s4l.py:396-404), polynomial LR on every group.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import torch_oracle as TO

RC_PREFIX = "rotation_classifier."


def init_rc_state(in_channels=21, seed=0):
    """Default torch initialisers of RotationClassifer.__init__ in its construction order (conv1, bn1, conv2, bn2,
    classifier) under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    c = in_channels
    mods = OrderedDict(conv1=nn.Conv2d(c, c, 4, 2, 1), bn1=nn.BatchNorm2d(c), conv2=nn.Conv2d(c, 2 * c, 4, 2, 1),
                       bn2=nn.BatchNorm2d(2 * c), classifier=nn.Linear(2 * c, 4))
    sd = OrderedDict()
    for name, m in mods.items():
        for k, v in m.state_dict().items():
            sd[name + "." + k] = v.detach().clone()
    return sd


def rc_is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked")


def _bn(sd, prefix, x, train, momentum=0.1, eps=1e-5):
    if train:
        sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    rm, rv = sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], train, momentum, eps)
    sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = rm, rv
    return y


def rc_forward(sd, task_pred, train=True):
    """RotationClassifer.forward (ssl_s4l.py:384-391); running statistics of `sd` are replaced when train."""
    x = F.leaky_relu(_bn(sd, "bn1", F.conv2d(task_pred, sd["conv1.weight"], sd["conv1.bias"], 2, 1), train), 0.2)
    x = F.leaky_relu(_bn(sd, "bn2", F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], 2, 1), train), 0.2)
    x = x.mean((2, 3))
    return F.linear(x, sd["classifier.weight"], sd["classifier.bias"])


def rotate(t, angle_idx):
    """_rotate_tensor on a [C, H, W] tensor (ssl_s4l.py:347-355)."""
    if angle_idx == 1:
        return t.transpose(1, 2).flip(2)
    if angle_idx == 2:
        return t.flip(2).flip(1)
    if angle_idx == 3:
        return t.transpose(1, 2).flip(1)
    return t


def batch_prehandle(inp, gt, angles):
    """_batch_prehandle(is_train=True) for one input / one gt tensor; `angles` = the np.random.randint(1, 4, bs) draw."""
    bs = inp.shape[0]
    assert inp.shape[2] == inp.shape[3], "S4L rotates by 90 degrees: square inputs"
    x = torch.zeros((2 * bs,) + tuple(inp.shape[1:]))
    g = torch.zeros((2 * bs,) + tuple(gt.shape[1:]))
    rot = torch.zeros(2 * bs)
    for s in range(bs):
        x[s], g[s] = inp[s], gt[s]
        x[bs + s], g[bs + s] = rotate(inp[s], int(angles[s])), rotate(gt[s], int(angles[s]))
        rot[bs + s] = float(angles[s])
    return x, g, rot.long()


class S4LOracleTrainer(TO.OracleTrainer):
    """SSLS4L._train, one iteration per call.  hp adds rotated_sup_scale, rotation_scale."""

    def __init__(self, state, rc_state, hp):
        super().__init__(state, hp)
        self.hp.setdefault("rotated_sup_scale", 0.5)
        self.hp.setdefault("rotation_scale", 0.1)
        self.rc = rc_state
        self.rc_mom = {}

    def s4l_step(self, inp, gt, lbs, angles):
        """inp [bs,3,H,H], gt [bs,1,H,H] (labeled first), lbs = ORIGINAL labeled batch size, angles [bs] in 1..3."""
        hp = self.hp
        bs = inp.shape[0]
        x, g, rot_gt = batch_prehandle(inp, gt, angles)
        leaves = TO._param_leaves(self.sd)
        run = TO._with_leaves(self.sd, leaves)
        rc_leaves = OrderedDict((k, v.detach().requires_grad_(True)) for k, v in self.rc.items() if not rc_is_buffer(k))
        rc_run = OrderedDict(self.rc)
        rc_run.update(rc_leaves)
        logits, prob, _, _ = self.forward(run, x, train=True)
        for k in self.sd:
            if TO.is_buffer(k):
                self.sd[k] = run[k]
        pred_rot = rc_forward(rc_run, logits, train=True)
        for k in self.rc:
            if rc_is_buffer(k):
                self.rc[k] = rc_run[k]
        unrot = TO.sseg_criterion(logits[:lbs], g[:lbs], hp["ignore_index"]).mean()
        rotated = hp["rotated_sup_scale"] * TO.sseg_criterion(logits[bs:bs + lbs], g[bs:bs + lbs], hp["ignore_index"]).mean()
        rot_loss = hp["rotation_scale"] * F.cross_entropy(pred_rot, rot_gt)
        (unrot + rotated + rot_loss).backward()
        acc = (pred_rot.detach().argmax(1) == rot_gt).float().sum().item() * 100.0 / (2 * bs)
        out = dict(unrotated_task_loss=float(unrot.detach()), rotated_task_loss=float(rotated.detach()),
                   rotation_loss=float(rot_loss.detach()), rotation_acc=acc, pred_rotation=pred_rot.detach().clone(),
                   logits=logits.detach())
        lrs = self._lrs()
        with torch.no_grad():
            TO.sgd_step(self.sd, OrderedDict((k, v.grad) for k, v in leaves.items()), self.mom, lrs, hp["momentum"],
                        hp["weight_decay"])
            # the rotation classifier is its own param group at the base learning rate (ssl_s4l.py:403-404)
            for k, v in rc_leaves.items():
                p = self.rc[k]
                d = v.grad.add(p, alpha=hp["weight_decay"])
                if k not in self.rc_mom:
                    self.rc_mom[k] = d.clone()
                else:
                    self.rc_mom[k].mul_(hp["momentum"]).add_(d)
                p.add_(self.rc_mom[k], alpha=-lrs[0])
        self.it += 1
        return out


def draw_angles(bs):
    """The draw of _batch_prehandle (ssl_s4l.py:298) from numpy's global stream."""
    return np.random.randint(low=1, high=4, size=bs)
