#!/bin/bash
# Experiment (GPU box): the wave policy at SMALL feature counts (latency-bound calls): every variant x 200 ... 1 000 features x 1 - 4 waves.
SPECS=""
for n in 200 400 700 1000; do for mm in lssd:fast lssd:direct lssd:inverse affine:inverse affine:direct affine:fast basic:direct basic:inverse basic:fast; do SPECS="$SPECS $mm:$n:6"; done; done
for w in default 1 2 3 4; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 400 python scripts/time_variant.py $SPECS --steps 30 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
