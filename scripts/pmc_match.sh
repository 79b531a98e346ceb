#!/bin/bash
# GPU box: PMC counters of the Hamming matcher kernels (one rocprofv3 --pmc pass per group, no tracing); env assignments as arguments
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum")
i=0
for grp in "${GROUPS_[@]}"; do
  rm -rf /tmp/pm$i; rocprofv3 --pmc $grp --output-format csv -d /tmp/pm$i -- python3 $ROOT/scripts/bench_configs.py --only match --quick > /tmp/pm$i.log 2>&1
  f=$(find /tmp/pm$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'hamming_match' in r['Kernel_Name']:
        acc[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    v.sort(); print('%-42s %-28s n %3d median %14.0f' % (k, c, len(v), v[len(v)//2]))
PY
  i=$((i+1))
done
