"""Experiment helper: distribution of the per-feature tick totals of a FTK_STAMPS_DUMP file (-DFTK_STAMPS build)."""
import sys

import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
tot = a[:, 7]
print("features", len(tot), "mean total ticks %.0f" % tot.mean(), "max %.0f" % tot.max(), "p99 %.0f" % np.percentile(tot, 99), "p999 %.0f" % np.percentile(tot, 99.9))
print("top 10:", np.sort(tot)[-10:].astype(int))
print("sum / (1024 SIMD * waves...) : total wave-ticks %.3g" % tot.sum())
