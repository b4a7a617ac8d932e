"""Which switch removes the intermittent discrete error levels (3.5e-6 / 5.6e-4 / 2.7e-3) of the two-rank fp32 gradient against the
single-rank full-batch gradient (tests/test_gpu_dist.py)?  Spawns the test's two workers under several environments, several trials each."""
import os
import sys
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import test_gpu_dist as T
    import torch_oracle as TO
    from pixelssl_amd import functional as PF
    from pixelssl_amd.engine import DeepLabV2Core
    state = TO.init_deeplabv2_state(seed=3, layers=T.LAYERS)
    x, gt = TO.synthetic_batch(4, 65, 4, seed=4, block=16)
    os.environ["PXL_FORCE_CLAMP_VAR"] = "1"
    refs = []
    for k in range(3):
        core = DeepLabV2Core(backbone=T.LAYERS, device="cuda:0", engine_dtype=torch.float32)
        core.autotune = False
        core.load_state_dict(state)
        core.train()
        logits, _, _ = core(x.cuda())
        PF.cross_entropy_per_sample(logits, gt.cuda(), 255).mean().backward()
        torch.cuda.synchronize()
        refs.append(core.flat.grads.detach().cpu().clone())
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    print("single-rank run-to-run: %.2e %.2e" % (rel(refs[1], refs[0]), rel(refs[2], refs[0])), flush=True)
    del os.environ["PXL_FORCE_CLAMP_VAR"]
    configs = [("default", {}), ("one bucket", {"PXL_TEST_BUCKET_MB": "4096"}), ("no side stream", {"PXL_SIDE_STREAM": "0"}),
               ("no fused join", {"PXL_FUSE_JOIN": "0"}), ("no fused bn reduce", {"PXL_FUSE_BN_REDUCE": "0"}),
               ("no onload", {"PXL_BN_ONLOAD": "0"}), ("f32 generic", {"PXL_F32_DMA": "0"})]
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for name, env in configs:
        errs = []
        for t in range(trials):
            for k, v in env.items():
                os.environ[k] = v
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            port = T._free_port()
            procs = [ctx.Process(target=T._worker, args=(r, 2, port, q, "1")) for r in range(2)]
            for p in procs:
                p.start()
            res = dict(q.get(timeout=600) for _ in procs)
            for p in procs:
                p.join(timeout=120)
            for k in env:
                del os.environ[k]
            g = torch.from_numpy(res[0][str(torch.float32)]["grads"])
            g2 = torch.from_numpy(res[0][str(torch.float32)]["grads2"])
            errs.append("%.1e/%.1e" % (rel(g, refs[0]), rel(g2, 2 * g)))
        print("%-20s grads vs full batch / second pass vs 2x first: %s" % (name, "  ".join(errs)), flush=True)


if __name__ == "__main__":
    main()
