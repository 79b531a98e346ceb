#!/usr/bin/env python3
"""Diagnostic (GPU box): the launch permutation the order kernel produces for a workload, under FTK_KLT_SCHED_MODE.
    FTK_KLT_SCHED_MODE=3 python scripts/sched_dump.py config4"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = sys.argv[1] if len(sys.argv) > 1 else "config4"
dump = os.path.join(tempfile.gettempdir(), "sched_dump.bin")
env = dict(os.environ, FTK_KLT_SCHED_DUMP=dump)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], env=env, check=True,
               stdout=subprocess.DEVNULL)
d = np.fromfile(dump, dtype=np.int32)
n = d.size // 2
order, iters = d[:n], d[n:]
assert np.array_equal(np.sort(order), np.arange(n)), "not a permutation"
print("n", n, "first 48 slots (feature:iters):", " ".join(f"{o}:{iters[o]}" for o in order[:48]))
# structure inside the biggest bin
vals, counts = np.unique(iters, return_counts=True)
big = vals[np.argmax(counts)]
sel = order[iters[order] == big]
print("largest bin: count", big, "size", sel.size, "first 40 features:", sel[:40].tolist())
print("   mean |step| between consecutive features in that bin:", float(np.abs(np.diff(sel)).mean()), " fraction of ascending steps:", float((np.diff(sel) > 0).mean()))
