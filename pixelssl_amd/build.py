"""Build libpixelhip.so (HIP kernels + C-ABI + network executor) for gfx950, in-tree.

    python -m pixelssl_amd.build            # incremental
    python -m pixelssl_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
snapshot to the GPU box (it is NOT listed in .gpurunignore).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpixelhip.so")
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_dma_kernel.h"), os.path.join(CSRC, "conv_halo_kernel.h"),
           os.path.join(os.path.dirname(HERE), "include", "pixelhip.h")]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wno-unused-result", "-x", "hip"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    if force or _stale(obj, [src] + HEADERS):
        cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
