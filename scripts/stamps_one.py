"""Experiment helper: the per-phase ticks of the slowest feature in a FTK_STAMPS_DUMP file (-DFTK_STAMPS build)."""
import sys

import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
names = ["ref_stage(level windows)", "setup", "cur_stage (affine levels: lanes of wave 0 that sampled global memory)", "phaseA(produce)", "count(first chain)", "chain->LDLT", "solve(barrier)", "total"]
i = int(np.argmax(a[:, 7]))
print("feature", i, {n: int(v) for n, v in zip(names, a[i])}, "unaccounted", int(a[i, 7] - a[i, :7].sum()))
