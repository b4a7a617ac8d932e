"""CutMix: box-mask generator vs the reference's boxes, oracle vs fixture (not gpu); device
mask-and-mix + confidence and the mirrored training step vs the reference's logged losses (gpu)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
FX = os.path.join(ROOT, "tests", "golden", "cutmix_65.pt")
DEV = "cuda"


def test_box_masks_reproduce_the_reference_stream():
    import cutmix_oracle as CO
    from pixelssl_amd.ssl_algorithm.ssl_cutmix import BoxMaskGenerator
    fx = torch.load(FX, weights_only=False)
    want = fx["masks_seed3"].numpy().astype(np.float32)
    size = fx["size"]
    assert np.array_equal(CO.box_masks(5, (size, 97), (0.5, 0.5), np.random.RandomState(3)), want)
    gen = BoxMaskGenerator((0.5, 0.5), 1, True, True, True, True, rng=np.random.RandomState(3))
    got = gen.produce(5, (size, 97))
    assert np.array_equal(got, want)
    # area proportion 0.5 +- rounding, box inside the image
    frac = got.reshape(5, -1).mean(1)
    assert np.all(np.abs(frac - 0.5) < 0.03)
    with pytest.raises(NotImplementedError):
        BoxMaskGenerator((0.5, 0.5), boxes_num=2, invert=True)


@pytest.mark.gpu
def test_cutmix_mix_and_confidence():
    from pixelssl_amd.ssl_algorithm.ssl_cutmix import cutmix_mix, BoxMaskGenerator
    g = torch.Generator().manual_seed(1)
    B, C, H, W = 3, 21, 65, 97
    a = torch.softmax(torch.randn(B, C, H, W, generator=g) * 3, 1)
    b = torch.softmax(torch.randn(B, C, H, W, generator=g) * 3, 1)
    mask = torch.from_numpy(BoxMaskGenerator((0.3, 0.7), 1, True, True, True, True, rng=np.random.RandomState(5)).produce(B, (H, W)))
    want = mask * a + (1 - mask) * b
    got, conf = cutmix_mix(mask.to(DEV), a.to(DEV), b.to(DEV), 0.6)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    assert abs(conf.item() - (want.max(dim=1)[0] > 0.6).float().mean().item()) < 1e-6
    imgs = torch.randn(B, 3, H, W, generator=g)
    got2 = cutmix_mix(mask.to(DEV), imgs.to(DEV), imgs.flip(0).contiguous().to(DEV))
    assert torch.equal(got2.cpu(), mask * imgs + (1 - mask) * imgs.flip(0))


@pytest.mark.gpu
def test_sslcutmix_train_steps_vs_reference_meters():
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = torch.load(FX, weights_only=False)
    lbs, ubs = fx["lbs"], fx["ubs"]
    args = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4,
                              momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1,
                              epochs=1, iters_per_epoch=4, ignore_index=255, labeled_batch_size=lbs,
                              unlabeled_batch_size=ubs, batch_size=lbs + ubs, ignore_unlabeled=False, is_epoch_lrer=False,
                              log_freq=1000, task="sseg", engine_dtype="fp32", cons_type="mse", cons_scale=fx["cons_scale"],
                              cons_rampup_epochs=0, cons_threshold=fx["cons_threshold"], ema_decay=0.99,
                              mask_prop_range=(0.5, 0.5))
    algo = P.ssl_algorithm.ssl_cutmix.ssl_cutmix(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                                {"model": plr.polynomiallr(args)},
                                                {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.module.model.load_state_dict(TO.init_deeplabv2_state(seed=fx["weight_seed"]))
    algo.t_model.module.model.load_state_dict(TO.init_deeplabv2_state(seed=fx["weight_seed"] + 1))
    algo.mask_generator.rng = np.random.RandomState(fx["np_seed"])       # the reference's global-RNG stream
    algo.s_model.train()
    algo.t_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(lbs + ubs, fx["size"], lbs, seed=s, block=fx["block"])
        out = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, 0)
        got = {k: v.item() for k, v in out.items()}
        ref = fx["per_iter"][i]
        print("cutmix iter", i, got, ref)
        # iteration 0 is parity; iteration 1 follows one SGD step of the ill-conditioned random-init net
        # (test_gpu_net.py), and its confidence counts softmax maxima around the threshold -> sanity band
        # (repeated runs of this same binary give iteration-1 consistency losses between 0.5x and 1.5x of the reference:
        # it is a mean over the pixels whose teacher confidence passes the threshold, a count that flips with the
        # noise; it is required to stay finite, positive and within 4x)
        for k in ("task_loss", "cons_loss"):
            if i > 0 and k == "cons_loss":
                assert got[k] == got[k] and 0.25 * ref[k] - 1e-7 <= got[k] <= 4 * ref[k] + 1e-7, (i, k, got[k], ref[k])
                continue
            tol = 1e-3 if i == 0 else 0.25
            assert abs(got[k] - ref[k]) < tol * abs(ref[k]) + 1e-7, (i, k, got[k], ref[k])
