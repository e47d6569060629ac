#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing, plus the issue-order 'tape' of the large blocks
(M mfma, v valu, t transcendental, d LDS, g global/DMA, s salu, w s_waitcnt, B barrier) -- the static check used while
software-pipelining the attention kernels (no GPU needed).
   python tools/isa_blocks.py file.s <kernel-name-substring> [min_block_size] [--tape]"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'): return 'M'
    if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', op): return 't'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_'): return 'd'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'g'
    if op.startswith('s_waitcnt'): return 'w'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_'): return 's'
    return '?'


def main():
    txt = open(sys.argv[1]).read()
    name = sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 40
    tape = '--tape' in sys.argv
    m = re.search(r'^(\S*%s\S*):' % re.escape(name), txt, re.M)
    if not m:
        sys.exit('kernel not found')
    i = m.end()
    j = txt.index('s_endpgm', i)
    blocks = []
    cur = ['entry', [], collections.Counter()]
    blocks.append(cur)
    for line in txt[i:j].splitlines():
        s = line.strip()
        lab = re.match(r'^(\.LBB\d+_\d+):', s) or re.match(r'^; (%bb\.\d+):', s)
        if lab:
            cur = [lab.group(1) + (' (loop)' if 'Loop' in s else ''), [], collections.Counter()]
            blocks.append(cur)
            continue
        if not s or s.startswith((';', '.')):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[1].append((c, op, s))
        cur[2][op] += 1
    print(m.group(1))
    for lab, ins, cnt in blocks:
        if len(ins) < minsz:
            continue
        mix = collections.Counter(c for c, _, _ in ins)
        print('%-22s n=%4d  %s' % (lab, len(ins), ' '.join('%s=%d' % kv for kv in sorted(mix.items()))))
        print('     top valu:', ', '.join('%s %d' % (k, v) for k, v in cnt.most_common(40) if k.startswith('v_') and not k.startswith('v_mfma'))[:400])
        if tape:
            t = ''.join(c for c, _, _ in ins)
            for k in range(0, len(t), 120):
                print('     ' + t[k:k + 120])


if __name__ == '__main__':
    main()
