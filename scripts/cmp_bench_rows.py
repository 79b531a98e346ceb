"""Experiment helper: per-row time difference of two bench_configs.py outputs."""
import json, sys
def load(p):
    d = {}
    for l in open(p):
        try: r = json.loads(l)
        except Exception: continue
        t = r.get('gpu_kernel_ms') or r.get('gpu_call_ms') or r.get('gpu_launch_ms') or r.get('gpu_host_call_ms')
        d[r['case']] = t * 1e3
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
for k in a:
    if k in b: print('%-50s %9.1f %9.1f  %+5.1f%%' % (k, a[k], b[k], 100 * (b[k] / a[k] - 1)))
