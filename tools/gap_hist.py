#!/usr/bin/env python3
"""Filler instructions per MFMA gap in the steady-state loop of an f16 kernel (from hipcc -S output).
usage: tools/gap_hist.py file.s [ILb0|ILb1]   (NN | TN)
One wave per SIMD hides ~3 single-issue instructions behind a 16x16x32 MFMA; gaps with more pay ~4.8 cycles each (r02_evidence 3d)."""
import sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n')
which = sys.argv[2] if len(sys.argv) > 2 else 'ILb0'
start = [i for i, l in enumerate(L) if l.startswith('_ZN') and 'gemm_f16_m16_kernel' + which in l][0]
end = [i for i, l in enumerate(L) if i > start and 's_endpgm' in l][0]
K = L[start:end]
print('spills:', sum('Folded' in l for l in K), ' scratch:', sum('scratch_' in l for l in K))
hs = [i for i, l in enumerate(K) if 'Loop Header' in l][0]
lbl = K[hs].split(':')[0]
be = [i for i, l in enumerate(K) if i > hs and 's_cbranch' in l and lbl in l][0]
body = [l.strip() for l in K[hs + 1:be + 1] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
print(len(body), 'instructions in the steady loop;', sum(1 for l in body if l.startswith('v_mfma')), 'MFMAs')
gaps, cur = [], []
for l in body:
    if l.startswith('v_mfma'):
        gaps.append(cur); cur = []
    else:
        cur.append(l.split()[0])
gaps.append(cur)
gaps[0] = gaps[-1] + gaps[0]  # the gap across the back-edge
gaps.pop()
print('fillers per gap histogram:', sorted(Counter(len(g) for g in gaps).items()), ' excess over 3:', sum(max(0, len(g) - 3) for g in gaps))
for i, g in enumerate(gaps):
    print(i, len(g), ' '.join(g))
