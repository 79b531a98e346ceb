#!/bin/bash
# Experiment (GPU box): where three waves per feature stop paying (13 x 13: 1 100 ... 1 550 features, two against three waves).
SPECS=""
for n in 1100 1250 1350 1450 1550; do for mm in lssd:fast lssd:direct affine:direct basic:direct affine:inverse; do SPECS="$SPECS $mm:$n:6"; done; done
for w in 2 3; do export FTK_KLT_WAVES=$w
  timeout -k 10 400 python scripts/time_variant.py $SPECS --steps 30 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
