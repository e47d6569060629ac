"""Static check for prefetches the compiler defeated: compiles one kernel file to gfx950 assembly and reports, per loop,
`s_waitcnt vmcnt(N)` instructions that follow global loads of the SAME iteration before half of the loop's MFMA (or VALU)
work has been issued -- i.e. loads that were meant to fly under the compute but are waited for in front of it.

Found this way in round 2 (none of them visible in a profile as anything but "waiting"):
  * TN GEMM: ds_read_b64_tr_b16 after an LDS-DMA -> s_waitcnt vmcnt(0) in front of every k-tile (fixed: inline-asm DMA)
  * attention dK/dV, fused conv weight gradient, nearest-code search: `cond ? loaded : 0` / `loaded * c` right at the load
    pins the wait to the load (candidates TTTS_DKDV_LATE / TTTS_WGRAD_LATE / TTTS_VQ_UNGUARDED)
A countdown (vmcnt(15), vmcnt(14), ...) directly after a batch of loads is reported too: it is fine when the batch is
the PREVIOUS iteration's prefetch being consumed (the newest loads stay in flight), and a synchronous stage when it is not --
read the loop (`--trace KERNEL` prints its instruction classes in order).

usage: python tools/isa_scan.py ttts_amd/csrc/attn.hip [-DNAME=1 ...] [--trace attn_bwd_dkdv_kernelILi64ELb1]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def cls(line):
    t = line.strip()
    if t.startswith("v_mfma"): return "M"
    if t.startswith("ds_read"): return "r"
    if t.startswith("ds_write"): return "w"
    if t.startswith(("global_load", "buffer_load")): return "G"
    if t.startswith(("global_store", "buffer_store", "global_atomic")): return "S"
    if t.startswith("s_waitcnt"): return "[" + t.replace("s_waitcnt ", "") + "]"
    if t.startswith("s_barrier"): return "|BAR|"
    if t.startswith(("s_cbranch", "s_branch")): return "<" + t.split()[0][2:] + ">"
    if t.startswith("v_"): return "v"
    if t.startswith("s_"): return "s"
    return ""


def kernels(asm):
    for m in re.finditer(r"\n(_ZN4ttts[^\n:]*): ", asm):
        st = m.end()
        yield m.group(1), asm[st:asm.index(".end_amdhsa_kernel", st)].split("\n")


def loops(lines):
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)
        if m:
            key = "Header=" + m.group(1).replace(".L", "")
            idx = [j for j, x in enumerate(lines) if key in x]
            yield m.group(1), lines[i:(max(idx) if idx else i) + 40]


def compress(seq, limit):
    out, prev, cnt = "", "", 0
    for c in seq:
        if c == prev and len(c) == 1:
            cnt += 1
            continue
        if prev:
            out += prev + (str(cnt) if cnt > 1 else "") + " "
        prev, cnt = c, 1
    return (out + prev + (str(cnt) if cnt > 1 else ""))[:limit]


def main():
    args = sys.argv[1:]
    src = args[0]
    defs = [a for a in args[1:] if a.startswith("-D")]
    trace = args[args.index("--trace") + 1] if "--trace" in args else None
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "ttts_amd", "csrc"), "--cuda-device-only", "-S", "-o", out, src] + defs,
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    for name, lines in kernels(asm):
        for label, body in loops(lines):
            seq = [c for c in map(cls, body) if c]
            if trace:
                if trace in name:
                    print(name, label)
                    print(compress(seq, 4000), "\n")
                continue
            n_m, n_g = seq.count("M"), seq.count("G")
            if n_g == 0 or (n_m < 4 and seq.count("v") < 100):
                continue
            total = n_m if n_m else seq.count("v")
            work, pending, early = 0, 0, []
            for c in seq:
                if c == "G":
                    pending += 1
                elif c == "M" or (n_m == 0 and c == "v"):
                    work += 1
                elif c.startswith("[") and "vmcnt" in c and pending:
                    if work / max(1, total) < 0.5:
                        early.append((c, pending, round(work / max(1, total), 2)))
                    if "vmcnt(0)" in c:
                        pending = 0
            if early:
                print("%-90s loop %-10s MFMA %3d loads %3d | early waits (wait, loads issued before it, work fraction done): %s"
                      % (name[8:98], label, n_m, n_g, early[:5]))


if __name__ == "__main__":
    main()
