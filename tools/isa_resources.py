#!/usr/bin/env python
"""Per-kernel register / LDS / scratch budget of the built library, read from the gfx950 code objects -- no GPU needed.

    python tools/isa_resources.py [--all] > profiles/r05_isa_resources.txt

For every object under pixelssl_amd/csrc/_obj the .hip_fatbin section is unbundled (llvm-objcopy + clang-offload-bundler), the
AMDGPU metadata note is read (llvm-readelf --notes) and one line per kernel is printed: VGPRs (arch + accumulation), SGPRs, static
LDS, scratch (spills), work-group size, and what that allows per SIMD / per CU on CDNA4 (512 unified VGPRs per lane per SIMD,
allocation granule 8; 160 KB LDS per CU; MI355X_MICROARCH.md).  The default listing keeps the kernels of the training step's
profile (profiles/r05_*_kernel_stats.csv names) -- --all prints every kernel (hundreds of template instances)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
OBJ = os.path.join(ROOT, "pixelssl_amd", "csrc", "_obj")

HOT = ("conv_dma_kernel", "wgrad_dma", "bn_apply_fwd", "residual_fwd", "bn_bwd", "residual_bwd", "sgd_ema_pack", "head_loss",
       "maxpool", "splitk_finish", "bn_reduce", "peer_allreduce", "pack_", "colsum", "igemm", "upsample")


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(tool):
        return names
    p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")[:len(names)] if p.returncode == 0 else names


def short(name, width=118):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0                                            # drop the argument list: the last balanced (...) group
    if name.endswith(")"):
        for i in range(len(name) - 1, -1, -1):
            depth += (name[i] == ")") - (name[i] == "(")
            if depth == 0:
                name = name[:i]
                break
    return name if len(name) <= width else name[:width - 3] + "..."


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "x.fatbin")
    co = os.path.join(tmp, "x.co")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return []
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
    if r.returncode != 0:
        return []
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    os.remove(fat)
    os.remove(co)
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key, d=0: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, d])[1]
        out.append(dict(name=get("name", "?"), agpr=int(get("agpr_count")), vgpr=int(get("vgpr_count")), sgpr=int(get("sgpr_count")),
                        lds=int(get("group_segment_fixed_size")), scratch=int(get("private_segment_fixed_size")),
                        wg=int(get("max_flat_workgroup_size")), vspill=int(get("vgpr_spill_count")),
                        sspill=int(get("sgpr_spill_count")), dyn_stack=get("uses_dynamic_stack", "false")))
    return out


def occupancy(k):
    """waves per SIMD the register file allows, work-groups per CU the LDS allows (static LDS only)"""
    regs = max(8, (k["vgpr"] + 7) // 8 * 8)             # .vgpr_count is the unified count (arch + acc) on gfx90a+
    waves = min(8, 512 // regs)
    by_lds = (160 * 1024) // k["lds"] if k["lds"] else 99
    return waves, by_lds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(OBJ):
        sys.exit("no objects under %s: run `python -c 'import __graft_entry__ as g; g.build()'` first" % OBJ)
    print("# libpixelhip.so, gfx950 code objects: per-kernel resources (tools/isa_resources.py%s)" % (" --all" if a.all else ""))
    print("# vgpr = unified count (arch + acc), waves/SIMD = min(8, 512 // ceil8(vgpr)); wg/CU(lds) = 160 KB // static LDS")
    total = spills = 0
    with tempfile.TemporaryDirectory() as tmp:
        for fn in sorted(os.listdir(OBJ)):
            if not fn.endswith(".o"):
                continue
            ks = kernels_of(os.path.join(OBJ, fn), tmp)
            if not ks:
                continue
            names = demangle([k["name"] for k in ks])
            rows = []
            for k, n in zip(ks, names):
                total += 1
                spilled = k["vspill"] or k["sspill"] or k["scratch"]
                spills += 1 if spilled else 0
                if a.all or spilled or any(h in n for h in HOT):
                    rows.append((k, n))
            print("\n## %s: %d kernels%s" % (fn[:-2], len(ks), "" if a.all else ", %d listed" % len(rows)))
            print("%5s %5s %5s %7s %7s %4s %6s %6s  %s" % ("vgpr", "agpr", "sgpr", "lds B", "scratch", "wg", "w/SIMD", "wg/CU", "kernel"))
            seen = set()
            for k, n in sorted(rows, key=lambda r: short(r[1])):
                key = (short(n), k["vgpr"], k["lds"])
                if key in seen:
                    continue
                seen.add(key)
                w, l = occupancy(k)
                print("%5d %5d %5d %7d %7d %4d %6d %6s  %s%s" % (k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"], k["wg"], w,
                                                               l if l < 99 else "-", short(n),
                                                               "   <-- SPILLS (v %d, s %d)" % (k["vspill"], k["sspill"]) if k["vspill"] or k["sspill"] else ""))
    print("\n# %d kernels in the library, %d with scratch or register spills" % (total, spills))


if __name__ == "__main__":
    main()
