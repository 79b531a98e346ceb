#!/bin/bash
# PMC pass over the float-descriptor matcher (GPU box): scripts/pmc_cosine.sh <tag>
set -u
TAG=${1:-c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 $ROOT/scripts/bench_configs.py --only cosine --quick > "$OUT/${name}.log" 2>&1 || echo "failed: $grp" >> "$OUT/errors.log"
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,statistics,collections,re
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'((cosine|hamming)_\w+(<[^>]*>)?)', r['Kernel_Name'])
        if m: vals[m.group(1)+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(vals):
    print(k)
    for c,v in sorted(vals[k].items()): print(f"    {c:28s} median {statistics.median(v):14.0f} min {min(v):14.0f} max {max(v):14.0f} n={len(v)}")
PY
