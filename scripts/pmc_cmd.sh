#!/bin/bash
# Instruction mix / wait counters of the kernels of ANY command (GPU box):
#   scripts/pmc_cmd.sh <tag> <kernel name substring> python3 scripts/bench_configs.py --quick --only direct
# One rocprofv3 --pmc pass per counter group (never combined with a trace); prints the median per counter over the launches of every
# matching kernel and the derived per-wave figures.
set -u
TAG=$1; MATCH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  ( cd "$ROOT" && rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/${name}.log" 2>&1 )
done
python3 - "$OUT" "$MATCH" <<'PY'
import csv,glob,sys,statistics,collections
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if sys.argv[2] in k: vals[k[:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in vals.items():
    print(k)
    m={n:statistics.median(v) for n,v in c.items()}
    for n,v in sorted(m.items()): print(f"  {n:28s} {v:14.0f}")
    w=m.get('SQ_WAVES',0)
    if w:
        insts=sum(m.get(x,0) for x in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_SMEM','SQ_INSTS_BRANCH'))
        print(f"  per wave: {insts/w:.0f} instructions (VALU {m.get('SQ_INSTS_VALU',0)/w:.0f}, SALU {m.get('SQ_INSTS_SALU',0)/w:.0f}, LDS {m.get('SQ_INSTS_LDS',0)/w:.0f}, VMEM {m.get('SQ_INSTS_VMEM_RD',0)/w:.0f}), "
              f"{m.get('SQ_WAVE_CYCLES',0)/w*4:.0f} cycles, {m.get('SQ_WAVE_CYCLES',0)*4/max(insts,1):.2f} cycles per instruction, wait_any {m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
