#!/usr/bin/env python3
"""Digest of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gpt_step_once.py (tools/gpt_pmc.sh output) into
profiles/pmc_traffic.json["traffic_bytes_per_launch"]: HBM-side bytes per launch of every GPT kernel family bench.py reports
(FETCH_SIZE x 2 on gfx950 -- 128-byte read requests are tallied at 64 bytes, MI355X_MICROARCH.md -- plus WRITE_SIZE; rocprofv3
reports both in KB).   python tools/pmc_digest.py <fetch.txt> <write.txt> <tag> [<sq.txt>]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path, counter):
    out, cur = {}, None
    for line in open(path):
        if line.startswith(("void ", "ttts::")) or (line and not line.startswith(" ") and "(" in line):
            cur = line.strip()
        m = re.match(r"\s+%s\s+per launch\s+([0-9.]+)\s+\((\d+) launches\)" % counter, line)
        if m and cur:
            out[cur] = (float(m.group(1)), int(m.group(2)))
    return out


def family(name):
    epi = {"0": "store_bf16", "1": "gelu", "2": "resid_add", "3": "dgelu", "4": "store_f32"}
    m = re.search(r"gemm_nt_(?:glds|tall|wreg)_kernel<(\d)", name)
    if m:
        return "gemm_nt_kernel<%s>" % epi[m.group(1)]
    if "gemm_nt_kernel<" in name:
        return "gemm_nt_kernel<%s>" % epi[re.search(r"gemm_nt_kernel<(\d)", name).group(1)]
    if "attn_fwd" in name:
        return "attn_fwd_kernel"
    if "attn_bwd" in name or "attn_delta" in name:
        return "attn_bwd(delta+dkdv+dq)"
    if "gemm_tn_grouped" in name:
        return "gemm_tn_grouped_kernel"
    if "gemm_tn" in name:
        return "gemm_tn_kernel"
    return None


def main():
    fetch, write, tag = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    per_kernel, fam_bytes, fam_launches = {}, {}, {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
        b = (2.0 * f + w) * 1024.0
        per_kernel[k[:100]] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "bytes_per_launch": int(b), "launches": max(nf, nw)}
        fam = family(k)
        if fam:
            n = max(nf, nw)
            fam_bytes[fam] = fam_bytes.get(fam, 0.0) + b * n
            fam_launches.setdefault(fam, {})[k] = n
    # bytes per launch of a family = total bytes / launches of its MAIN kernel (the attention backward family counts one launch
    # per dK/dV kernel launch: delta + dkdv + dq together are one attn_bwd call)
    traffic = {}
    for fam, tot in fam_bytes.items():
        ks = fam_launches[fam]
        if fam.startswith("attn_bwd"):
            n = max(v for k, v in ks.items() if "dkdv" in k)
        elif fam == "gemm_tn_kernel":
            n = max(ks.values())
        else:
            n = sum(ks.values())
        traffic[fam] = int(tot / max(1, n))
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["traffic_bytes_per_launch"] = traffic
    doc["traffic_source"] = "%s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over tools/gpt_step_once.py; digest tools/pmc_digest.py" % tag
    doc["per_kernel_" + tag] = per_kernel
    doc.pop("raw_kb", None)
    doc["_note"] = ("HBM-side traffic per launch at the BASELINE shape (B 8, H 8, S 1156, d_h 64, dropout on). Units: bytes = (FETCH_SIZE x 2 + "
                    "WRITE_SIZE) x 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B). "
                    "Fabric-side counters include Infinity-Cache hits.")
    if len(sys.argv) > 4:     # the SQ pass of the same script: matrix-core busy fraction of the attention kernels (bench.py: roofline.attention_mfma_busy)
        busy, gui = read(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES"), read(sys.argv[4], "GRBM_GUI_ACTIVE")
        ab = {}
        for k in busy:
            name = "fwd" if "attn_fwd" in k else "dq" if "attn_bwd_dq" in k else "dkdv" if "attn_bwd_dkdv" in k else None
            if name and k in gui and "dh64" in k:
                # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
                ab[name] = round(busy[k][0] / (gui[k][0] / 8.0 * 1024.0), 4)
        ab["source"] = "%s: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), rocprofv3 --pmc over tools/gpt_step_once.py (BASELINE shape, dropout 0.1)" % tag
        doc["attention_mfma_busy"] = ab
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
