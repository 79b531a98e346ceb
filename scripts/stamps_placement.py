"""Experiment helper (GPU box output): where and when did the workgroups of a pipelined-kernel launch run?
Reads a FTK_STAMPS_DUMP file of a -DFTK_STAMPS build (slot 6: s_memrealtime at start, slot 4: at end (100 MHz), slot 2: XCC_ID << 32 |
HW_ID, slot 7: total shader ticks) and prints the per-CU workgroup counts and the finishing times by CU load."""
import sys
from collections import Counter, defaultdict

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t0, t1, hw, ticks = a[:, 6].astype(np.int64), a[:, 4].astype(np.int64), a[:, 2], a[:, 7].astype(np.float64)
ok = t0 > 0
t0, t1, hw, ticks = t0[ok], t1[ok], hw[ok], ticks[ok]
base = t0.min()
start, end = (t0 - base) * 0.01, (t1 - base) * 0.01  # us
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = hw.astype(np.int64) & 0xFFFFFFFF
wave, simd, cu, sh, se = hwid & 0xF, (hwid >> 4) & 0x3, (hwid >> 8) & 0xF, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7
key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
per_cu = Counter(key)
print("workgroups", len(t0), "CUs used", len(per_cu), "span %.2f us" % end.max(), "mean life %.2f us" % (end - start).mean())
print("workgroups per CU histogram:", sorted(Counter(per_cu.values()).items()))
by_load = defaultdict(list)
for k, e, s in zip(key, end, start):
    by_load[per_cu[k]].append((e, e - s))
for load in sorted(by_load):
    v = np.array(by_load[load])
    print("  CUs with %2d workgroups: %5d workgroups, end mean %.2f max %.2f us, life mean %.2f us" % (load, len(v), v[:, 0].mean(), v[:, 0].max(), v[:, 1].mean()))
per_simd = Counter(zip(key, simd.tolist()))
print("consumer waves per SIMD histogram:", sorted(Counter(per_simd.values()).items()))
by_simd = defaultdict(list)
for k, sd, e, s in zip(key, simd.tolist(), end, start):
    by_simd[per_simd[(k, sd)]].append((e, e - s))
for load in sorted(by_simd):
    v = np.array(by_simd[load])
    print("  SIMDs with %2d consumer waves: %5d workgroups, end mean %.2f max %.2f us, life mean %.2f" % (load, len(v), v[:, 0].mean(), v[:, 0].max(), v[:, 1].mean()))
print("per XCC workgroups:", sorted(Counter(xcc.tolist()).items()))
late = np.argsort(end)[-10:]
print("10 latest:", [(round(float(end[i]), 1), round(float(start[i]), 1), key[i], int(simd[i])) for i in late])
