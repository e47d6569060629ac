#!/usr/bin/env python3
"""tools/conv_census.py JSON (gpurun_out/conv_census.json or a copy under profiles/) -> the case list tools/ubench/conv_variants.cpp
reads: one convolution call signature per line, heaviest (count x time) first.
  python tools/ubench/conv_cases.py profiles/r02_conv_census_193ms.json [top] > tools/ubench/conv_cases.txt
line formats (slopes: 1 = none; gate: 0 / 1; out_act: 0 none, 1 tanh, 2 lrelu):
  fwd   count B Cin Lin Cout K stride pad dil groups in_slope gate out_act
  dgrad count B Cout Lout Cin Lin K stride pad dil groups in_slope gate
  wgrad count B Cout Lout Cin Lin K stride pad dil groups x_slope dy_slope"""
import ast
import json
import sys

SLOPE = 0.1          # modules.LRELU_SLOPE: the only leaky-relu slope fused into these calls
ACT = {None: 0, "none": 0, "tanh": 1, "lrelu": 2}


def main():
    rows = json.load(open(sys.argv[1]))["rows"]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
    rows = sorted(rows, key=lambda r: -r["total_ms"])
    print("# from %s: %d of %d signatures, %.1f of %.1f ms of one step" % (
        sys.argv[1], min(top, len(rows)), len(rows), sum(r["total_ms"] for r in rows[:top]), sum(r["total_ms"] for r in rows)))
    for r in rows[:top]:
        k = ast.literal_eval(r["sig"])
        if k[0] == "fwd":
            (b, cin, lin), (cout, _, kk), stride, pad, dil, groups, _, _, gate, _, slope, act = k[1:]
            print("fwd", r["count"], b, cin, lin, cout, kk, stride, pad, dil, groups, SLOPE if slope else 1, int(gate), ACT[act])
        elif k[0] == "dgrad":
            (b, cout, lout), (_, cpg, kk), lin, stride, pad, dil, groups, gate, _, _, slope = k[1:]
            print("dgrad", r["count"], b, cout, lout, cpg * groups, lin, kk, stride, pad, dil, groups, SLOPE if slope else 1, int(gate))
        elif k[0] == "wgrad":
            (b, cout, lout), (_, cin, lin), kk, stride, pad, dil, groups, xs, ds = k[1:]
            print("wgrad", r["count"], b, cout, lout, cin, lin, kk, stride, pad, dil, groups, SLOPE if xs else 1, SLOPE if ds else 1)


main()
