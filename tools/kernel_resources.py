#!/usr/bin/env python3
"""Register / LDS budget of every gfx950 kernel in ttts_amd/csrc: compiles each source to device assembly with the product's flags
(no GPU needed) and reads the code-object metadata -- VGPRs (+ AGPRs), SGPRs, spilled registers, scratch bytes, static LDS,
workgroup size -- and derives the waves per SIMD the register file allows (512 VGPRs per SIMD lane, granule 8).
  python tools/kernel_resources.py [file.hip ...] [--md]        (default: every source of ttts_amd.lib.SOURCES)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ttts_amd import lib  # noqa: E402

CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def demangle(names):
    for tool in (CXXFILT, "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return dict(zip(names, out))
        except (OSError, subprocess.CalledProcessError):
            continue
    return {n: n for n in names}


def resources(src):
    """[(kernel, {vgpr, agpr, sgpr, spill, scratch, lds, wg})] of one source file."""
    path = src if os.path.isabs(src) else os.path.join(lib.CSRC, src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = [f for f in lib.FLAGS if f not in ("-fPIC",)] + lib.EXTRA_FLAGS.get(os.path.basename(path), [])
        cmd = [lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out, path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-2000:]))
        asm = open(out).read()
    res = []
    meta = asm[asm.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in asm else ""
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d=0: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, blk) else d  # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        res.append((name, {"vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"), "spill": g("vgpr_spill_count"),
                           "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                           "wg": g("max_flat_workgroup_size")}))
    return res


def waves_per_simd(vgpr):
    """Waves one SIMD can hold at `vgpr` unified registers per lane (512 per SIMD, allocation granule 8, at most 8 waves)."""
    return max(1, min(8, 512 // max(8, (vgpr + 7) // 8 * 8)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = "--md" in sys.argv
    srcs = args or list(lib.SOURCES)
    with ThreadPoolExecutor(max_workers=8) as ex:
        allres = list(ex.map(resources, srcs))
    names = [n for res in allres for n, _ in res]
    dm = demangle(names)
    if md:
        print("| file | kernel | VGPR (+AGPR) | SGPR | spilled | scratch B | static LDS B | threads | waves / SIMD by registers |")
        print("|---|---|---|---|---|---|---|---|---|")
    for src, res in zip(srcs, allres):
        for n, r in sorted(res, key=lambda x: dm[x[0]]):
            k = re.sub(r"\(.*", "", dm[n]).replace("void ", "").replace("ttts::", "")
            tot = r["vgpr"]      # gfx950 reports the unified count (AGPRs included)
            row = (os.path.basename(src), k, "%d%s" % (r["vgpr"], " (%d)" % r["agpr"] if r["agpr"] else ""), r["sgpr"], r["spill"],
                   r["scratch"], r["lds"], r["wg"], waves_per_simd(tot))
            if md:
                print("| " + " | ".join(str(c) for c in row) + " |")
            else:
                print("%-18s %-62s vgpr %-9s sgpr %3d spill %3d scratch %4d lds %6d wg %4d waves/simd %d" % row)


if __name__ == "__main__":
    main()
