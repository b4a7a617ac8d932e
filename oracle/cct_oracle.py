"""CPU oracle for SSLCCT: the auxiliary decoders, the wrapped main-model + decoders forward and
the SSLCCT training iteration on PSPNet.

TEST INFRASTRUCTURE ONLY -- the checker, never the product (see torch_oracle.py for the rules).

Functional fp32 restatement of pixelssl/ssl_algorithm/ssl_cct.py (file:line cited per function); PINNED by
oracle/make_golden_cct.py, which runs the reference's own `SSLCCT._train` (WrappedCCTModel + VAT / DropOut / Con-Msk /
Obj-Msk / F-Drop / F-Noise decoders) on seeded inputs, asserts this file reproduces it and writes
tests/golden/cct_65.pt.

Parity note -- G-Cutout: `CutOutDecoder.guided_cutout` (ssl_cct.py:615-656) calls cv2.findContours (OpenCV, a
third-party dependency that is NOT installed in this image and not vendored by the reference), so that one decoder is
"parity unpinned": `external_contour_boxes` below restates the published algorithm (Suzuki-Abe border following,
RETR_EXTERNAL + CHAIN_APPROX_SIMPLE: outer borders of the 8-connected foreground components that are not enclosed by
another component, vertices = direction changes of the traced border) and the product's host routine is checked
against it, but neither could be compared with OpenCV here.  What is restated, and from where (OpenCV 3.4 / 4.x,
modules/imgproc/src/contours.cpp, recalled -- there is no network and no OpenCV in this image):
  * start pixels: the raster scan opens an outer border at a 0 -> 1 transition (`cvFindNextContour`); in RETR_EXTERNAL
    mode it skips hole borders and any outer border met while the last marked border pixel on the row is a positively
    marked one, i.e. while the scan is inside an already traced outer border -- the components kept are those that touch
    the background connected to the frame;
  * vertex rule (`icvFetchContour`, CHAIN_APPROX_SIMPLE): `prev_s` starts at -1 and the current point is written
    whenever the chain code of the step leaving it differs from the previous step's, so the start pixel is always a
    vertex and every later vertex is a change of direction; an isolated pixel is one point.  The raster-first pixel of a
    component is left going SW/S/SE/E and re-entered going W/NW/N/NE, never the same code, so "number of circular
    direction changes" (below) equals OpenCV's count; the count does not depend on the tracing sense;
  * list ORDER (`icvEndProcessContour`): in RETR_EXTERNAL / RETR_LIST mode each finished contour is linked in FRONT of
    its siblings (`contour->h_next = parent->first_child; parent->first_child = contour`), and `cv::findContours` walks
    `h_next` from the first one, so the list is newest-found first = REVERSE raster order of the start pixels (the
    familiar "contours come back bottom-to-top").  The order matters here: ssl_cct.py:637-638 draws two
    `random.randint` per kept contour in list order.  Round 4 switched this file and csrc/contour.cpp from raster to
    reverse-raster order and regenerated the G-Cutout fixtures.
tests/test_cct.py holds hand-derived known answers for these three rules (nested blobs, a 3n+1-vertex staircase either
side of the `> 50` filter, two blobs whose raster and list orders differ); a vector from a real OpenCV is still missing.
Round 5: the first rule (which outer borders are external) and the bounding boxes are additionally checked against
scipy.ndimage's connected-component labelling -- an implementation that shares nothing with this file or csrc/contour.cpp
(tests/test_cct.py::test_contour_boxes_against_an_independent_component_analysis); the vertex rule and the list order
remain restatements.

Randomness: the reference draws from four host RNG streams (torch CPU generator: I-VAT's `torch.rand`, Dropout2d,
F-Noise's Uniform.sample; numpy: F-Drop's threshold; python `random`: G-Cutout).  Every decoder function below takes
the draw as an optional argument and otherwise makes the SAME library call as the reference, so that with the same
seeds the streams line up bit-exactly; the draws are returned so that fixtures can carry them to the GPU tests.
"""
import math
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions.uniform import Uniform

import torch_oracle as TO

DECODER_KINDS = ("vat", "drop", "cut", "context", "object", "fd", "fn")      # ModuleList order, ssl_cct.py:189-199


def decoder_param_shapes(in_channels=512, num_classes=21, upscale=8):
    """`upsample(in_channels, num_classes, upscale)` (ssl_cct.py:524-532) inside a decoder module named `upsample`."""
    sd = OrderedDict()
    sd["upsample.0.weight"] = (num_classes, in_channels, 1, 1)
    for i in range(1, int(math.log(upscale, 2)) + 1):
        sd["upsample.%d.conv.weight" % i] = (num_classes * 4, num_classes, 1, 1)
        sd["upsample.%d.conv.bias" % i] = (num_classes * 4,)
    return sd


def init_decoder_state(seed, in_channels=512, num_classes=21, upscale=8):
    """Reference distributions (not its RNG stream): kaiming_normal(relu) for the 1x1 conv, ICNR for the PixelShuffle
    convs, torch's default bias (ssl_cct.py:497-521, 527-528)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in decoder_param_shapes(in_channels, num_classes, upscale).items():
        if name == "upsample.0.weight":
            sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[1])
        elif name.endswith("conv.weight"):
            base = torch.randn(shape[0] // 4, shape[1], 1, 1, generator=g) * math.sqrt(2.0 / shape[1])
            sd[name] = base.repeat_interleave(4, dim=0).contiguous()
        else:
            bound = 1.0 / math.sqrt(num_classes)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


def decoder_forward(sd, x):
    """The `upsample` Sequential: 1x1 conv (no bias) + 3 x [1x1 conv + bias, ReLU, PixelShuffle(2)]."""
    return TO.subpixel_decoder_forward(sd, x, prefix="upsample")


# ------------------------------------------------------------------------------------------------------------------
# perturbations (one per auxiliary decoder kind); each returns (perturbed latent, draw)
# ------------------------------------------------------------------------------------------------------------------

def _l2_normalize(d):
    """VATDecoder._l2_normalize (ssl_cct.py:577-581): per-sample L2 norm + 1e-8."""
    n = torch.norm(d.reshape(d.shape[0], -1), dim=1).reshape(-1, 1, 1, 1)
    return d / (n + 1e-8)


def vat_perturb(sd, x, xi, eps, d0=None):
    """VATDecoder.get_r_adv (ssl_cct.py:548-575), one power iteration: r_adv = eps * normalize(grad_d KL(p || p_hat))."""
    x_det = x.detach()
    with torch.no_grad():
        pred = F.softmax(decoder_forward(sd, x_det), dim=1)
    if d0 is None:
        d0 = torch.rand(x.shape).sub(0.5)
    d = _l2_normalize(d0.clone()).requires_grad_(True)
    frozen = OrderedDict((k, v.detach()) for k, v in sd.items())
    pred_hat = decoder_forward(frozen, x_det + xi * d)
    adv = F.kl_div(F.log_softmax(pred_hat, dim=1), pred, reduction="batchmean")
    (g,) = torch.autograd.grad(adv, d)
    return x + _l2_normalize(g) * eps, d0


def drop_perturb(x, rate, spatial=True, scale=None):
    """DropOutDecoder (ssl_cct.py:584-592): nn.Dropout2d(p) (spatial) or nn.Dropout(p) in training mode; draw = the keep-scale
    (0 or 1/(1-p)): [B, C] for the spatial kind, x's shape for the element-wise kind."""
    if scale is None:
        if spatial:
            scale = F.dropout2d(torch.ones(x.shape[0], x.shape[1], 1, 1), rate, True).reshape(x.shape[0], x.shape[1])
        else:
            scale = F.dropout(torch.ones(x.shape), rate, True)
    return x * (scale[:, :, None, None] if scale.dim() == 2 else scale), scale


def fg_mask(main_pred, size):
    """guided_masking's mask (ssl_cct.py:664-669): (argmax > 0), nearest-resized to the latent size."""
    m = (main_pred.argmax(1) > 0).float().unsqueeze(1)
    return F.interpolate(m, size=size, mode="nearest")


def context_perturb(x, main_pred):
    """ContextMaskingDecoder (ssl_cct.py:659-683): keep the features under the predicted objects."""
    return x * fg_mask(main_pred, x.shape[2:]), None


def object_perturb(x, main_pred):
    """ObjectMaskingDecoder (ssl_cct.py:686-711): keep the features under the predicted background."""
    return x * (1 - fg_mask(main_pred, x.shape[2:])), None


def fd_perturb(x, u=None):
    """FeatureDropDecoder.feature_dropout (ssl_cct.py:718-724)."""
    if u is None:
        u = float(np.random.uniform(0.7, 0.9))
    att = torch.mean(x, dim=1, keepdim=True)
    mx, _ = torch.max(att.reshape(x.shape[0], -1), dim=1, keepdim=True)
    thr = (mx * u).reshape(x.shape[0], 1, 1, 1).expand_as(att)
    return x.mul((att < thr).float()), u


def fn_perturb(x, uniform_range, noise=None):
    """FeatureNoiseDecoder.feature_based_noise (ssl_cct.py:738-741): x * U(-r, r)[C,h,w] + x."""
    if noise is None:
        noise = Uniform(-uniform_range, uniform_range).sample(x.shape[1:])
    return x.mul(noise.unsqueeze(0)) + x, noise


# ---- G-Cutout (parity unpinned: OpenCV absent, see the header) -------------------------------------------------------

_NB8 = ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1))     # clockwise from east (dy, dx)


def external_contour_boxes(mask, min_vertices=50):
    """Bounding boxes (min_x, max_x, min_y, max_y) of the external contours of a binary image whose CHAIN_APPROX_SIMPLE
    polygon has more than `min_vertices` vertices (ssl_cct.py:630-636), in OpenCV's list order: newest-found first, i.e.
    REVERSE raster order of the contours' first pixels (see the header)."""
    return [box for nvert, box in _external_contours(mask) if nvert > min_vertices]


def _external_contours(mask):
    """(vertex count, bounding box) of EVERY external contour.  Pure-python restatement of the published
    border-following algorithm (see the header)."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    pad = np.zeros((H + 2, W + 2), dtype=bool)
    pad[1:-1, 1:-1] = m
    # background connected to the frame (4-connectivity): components touching it are the external ones
    outer = np.zeros_like(pad)
    stack = [(0, 0)]
    outer[0, 0] = True
    while stack:
        y, x = stack.pop()
        for dy, dx in ((0, 1), (1, 0), (0, -1), (-1, 0)):
            yy, xx = y + dy, x + dx
            if 0 <= yy < H + 2 and 0 <= xx < W + 2 and not pad[yy, xx] and not outer[yy, xx]:
                outer[yy, xx] = True
                stack.append((yy, xx))
    seen = np.zeros_like(pad)
    boxes = []
    for y in range(1, H + 1):
        for x in range(1, W + 1):
            if not pad[y, x] or seen[y, x]:
                continue
            # flood the 8-connected component
            comp = [(y, x)]
            seen[y, x] = True
            stack = [(y, x)]
            while stack:
                cy, cx = stack.pop()
                for dy, dx in _NB8:
                    yy, xx = cy + dy, cx + dx
                    if pad[yy, xx] and not seen[yy, xx]:
                        seen[yy, xx] = True
                        comp.append((yy, xx))
                        stack.append((yy, xx))
            if not outer[y, x - 1]:
                continue                   # first pixel borders an enclosed hole: the component is not external
            nvert = _traced_vertices(pad, y, x)
            ys = [p[0] for p in comp]
            xs = [p[1] for p in comp]
            boxes.append((nvert, (min(xs) - 1, max(xs) - 1, min(ys) - 1, max(ys) - 1)))
    return boxes[::-1]                     # each finished contour is linked in front of the earlier ones (header)


def _traced_vertices(pad, y0, x0):
    """Moore border tracing of the outer border starting at the raster-first pixel (its west neighbour is background);
    returns the number of vertices left by CHAIN_APPROX_SIMPLE (points where the chain direction changes)."""
    # find the first neighbour clockwise, starting from the west (the pixel we 'came from')
    def next_from(y, x, start):
        for k in range(8):
            d = (start + k) % 8
            dy, dx = _NB8[d]
            if pad[y + dy, x + dx]:
                return d
        return -1
    d = next_from(y0, x0, 5)           # start the search just after the west neighbour (index 4), clockwise
    if d < 0:
        return 1                       # isolated pixel
    dirs = []
    y, x, first = y0, x0, d
    while True:
        dirs.append(d)
        y, x = y + _NB8[d][0], x + _NB8[d][1]
        nd = next_from(y, x, (d + 5) % 8)     # back-track neighbour + 1, clockwise
        if y == y0 and x == x0 and nd == first:
            break
        d = nd
        if len(dirs) > 8 * pad.size:
            raise RuntimeError("border tracing did not terminate")
    return sum(1 for i in range(len(dirs)) if dirs[i] != dirs[i - 1])


def find_contours_stand_in(mask_np, mode=None, method=None):
    """Stand-in for `cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)` (OpenCV is not installed): one int32
    array [n_vertices, 1, 2] of (x, y) points per external contour found by `external_contour_boxes`' border following,
    with the same vertex COUNT and the same bounding box as the traced polygon -- the two properties
    CutOutDecoder.guided_cutout reads (ssl_cct.py:632-636: `c.shape[0] > 50`, min / max of x and y).  Used by
    make_golden_cct.py to let the reference's own G-Cutout code run; what stays unpinned is exactly this function."""
    out = []
    for nvert, (min_w, max_w, min_h, max_h) in _external_contours(mask_np):
        corners = np.array([[min_w, min_h], [max_w, min_h], [max_w, max_h], [min_w, max_h]], dtype=np.int32)
        out.append(corners[np.arange(max(nvert, 1)) % 4].reshape(-1, 1, 2))
    return out, None


def cutout_mask(main_pred, erase, size, rnd=None, min_vertices=50):
    """CutOutDecoder.guided_cutout (ssl_cct.py:615-656): per sample, erase a random `erase`-sized window inside the
    bounding box of every (large enough) predicted object; nearest-resize to the latent size.  `rnd` = iterator of the
    uniform [0,1) draws replacing python's random.randint(0, n) as floor(u * (n + 1)) (two per box: w then h)."""
    masks = (main_pred.argmax(1) > 0)
    out = []
    draws = []
    for msk in masks:
        ones = np.ones(msk.shape, dtype=np.float32)
        for (min_w, max_w, min_h, max_h) in external_contour_boxes(msk.numpy(), min_vertices):
            bb_w, bb_h = max_w - min_w, max_h - min_h
            nw, nh = int(bb_w * (1 - erase)), int(bb_h * (1 - erase))
            # no injected draw: the reference's own stream, random.randint(0, n) (ssl_cct.py:637-638), recorded as the
            # uniform u with floor(u * (n + 1)) = k so that a replay needs no knowledge of n
            uw = (random.randint(0, nw) + 0.5) / (nw + 1) if rnd is None else next(rnd)
            uh = (random.randint(0, nh) + 0.5) / (nh + 1) if rnd is None else next(rnd)
            draws += [uw, uh]
            sw = int(uw * (nw + 1))
            sh = int(uh * (nh + 1))
            ones[min_h + sh:min_h + sh + int(bb_h * erase), min_w + sw:min_w + sw + int(bb_w * erase)] = 0
        out.append(ones)
    m = torch.from_numpy(np.stack(out)).unsqueeze(1)
    return F.interpolate(m, size=size, mode="nearest"), draws


def cut_perturb(x, main_pred, erase, rnd=None, min_vertices=50):
    m, draws = cutout_mask(main_pred, erase, x.shape[2:], rnd, min_vertices)
    return x * m, draws


# ------------------------------------------------------------------------------------------------------------------
# wrapped model + training iteration
# ------------------------------------------------------------------------------------------------------------------

def aux_forward(kind, cfg, sd, x, main_pred, draw=None):
    """One auxiliary decoder: perturb the latent, decode.  -> (prediction at the decoder's resolution, draw)"""
    if kind == "vat":
        xp, draw = vat_perturb(sd, x, cfg.get("xi", 1e-6), cfg.get("eps", 2.0), draw)
    elif kind == "drop":
        xp, draw = drop_perturb(x, cfg.get("rate", 0.5), cfg.get("spatial", True), draw)
    elif kind == "cut":
        xp, draw = cut_perturb(x, main_pred, cfg.get("erase", 0.4), iter(draw) if draw is not None else None,
                               cfg.get("min_vertices", 50))
    elif kind == "context":
        xp, draw = context_perturb(x, main_pred)
    elif kind == "object":
        xp, draw = object_perturb(x, main_pred)
    elif kind == "fd":
        xp, draw = fd_perturb(x, draw)
    elif kind == "fn":
        xp, draw = fn_perturb(x, cfg.get("uniform", 0.3), draw)
    else:
        raise ValueError(kind)
    return decoder_forward(sd, xp), draw


def cons_loss_of(ad_preds, main_prob):
    """WrappedCCTModel.forward (ssl_cct.py:481-487): bilinear (align_corners=False) to the prediction size, softmax,
    sum of MSE against the detached main soft-max, / number of decoders."""
    tgt = main_prob.detach()
    up = [F.interpolate(p, size=tgt.shape[2:], mode="bilinear") for p in ad_preds]
    return sum(F.mse_loss(F.softmax(p, dim=1), tgt) for p in up) / len(up)


class CCTOracleTrainer(TO.OracleTrainer):
    """SSLCCT._train body (ssl_cct.py:226-300), one iteration per call, PSPNet main model.
    decoders: list of (kind, cfg dict, state dict).  Extra hp: cons_scale, cons_rampup_iters, ad_lr_scale."""

    def __init__(self, state, decoders, hp, forward=None):
        super().__init__(state, hp, forward=forward or TO.pspnet_forward)      # or deeplabv2_forward (2048-ch latent)
        self.hp.update(dict(cons_scale=30.0, cons_rampup_iters=0, ad_lr_scale=10.0))
        self.hp.update(hp)
        self.decoders = decoders
        self.ad_mom = [dict() for _ in decoders]

    def cct_step(self, x, gt, lbs, draws=None):
        hp = self.hp
        ramp = TO.sigmoid_rampup(self.it, hp["cons_rampup_iters"])
        leaves = TO._param_leaves(self.sd)
        run = TO._with_leaves(self.sd, leaves)
        ad_leaves = [OrderedDict((k, v.detach().requires_grad_(True)) for k, v in sd.items()) for _, _, sd in self.decoders]
        # labeled pass, then a SEPARATE unlabeled pass through the same model (two BN batches, ssl_cct.py:248-266)
        l_logits, _, _, _ = self.forward(run, x[:lbs], train=True)
        task = TO.sseg_criterion(l_logits, gt[:lbs], hp["ignore_index"]).mean()
        out_draws = []
        if x.shape[0] > lbs:
            u_logits, u_prob, u_lat, _ = self.forward(run, x[lbs:], train=True)
            preds = []
            for i, (kind, cfg, _) in enumerate(self.decoders):
                p, d = aux_forward(kind, cfg, ad_leaves[i], u_lat, u_logits.detach(), None if draws is None else draws[i])
                preds.append(p)
                out_draws.append(d)
            cons = ramp * hp["cons_scale"] * cons_loss_of(preds, u_prob)
        else:
            cons = torch.zeros(())
        for k in self.sd:
            if TO.is_buffer(k):
                self.sd[k] = run[k]
        (task + cons).backward()
        grads = OrderedDict((k, v.grad) for k, v in leaves.items())
        lr, lr10 = self._lrs()
        with torch.no_grad():
            TO.sgd_step(self.sd, grads, self.mom, (lr, lr10), hp["momentum"], hp["weight_decay"])
            for i, (_, _, sd) in enumerate(self.decoders):
                for k, v in ad_leaves[i].items():
                    if v.grad is None:
                        continue
                    d = v.grad.add(sd[k], alpha=hp["weight_decay"])
                    if k not in self.ad_mom[i]:
                        self.ad_mom[i][k] = d.clone()
                    else:
                        self.ad_mom[i][k].mul_(hp["momentum"]).add_(d)
                    sd[k].add_(self.ad_mom[i][k], alpha=-lr * hp["ad_lr_scale"])
        self.it += 1
        return dict(task_loss=float(task.detach()), cons_loss=float(cons.detach()), draws=out_draws,
                    grads=OrderedDict((k, g.clone()) for k, g in grads.items()),
                    ad_grads=[OrderedDict((k, v.grad.clone()) for k, v in al.items() if v.grad is not None)
                              for al in ad_leaves])
