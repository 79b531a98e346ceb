#!/bin/bash
# Experiment (GPU box): the wave policy over every variant x 1 200 ... 3 000 features x one / two / three waves per feature (FTK_KLT_WAVES) beside the default.
SPECS=""
for n in 1200 1600 2000 2400 3000; do for mm in lssd:fast lssd:direct lssd:inverse affine:inverse affine:direct affine:fast basic:direct basic:inverse basic:fast; do SPECS="$SPECS $mm:$n:6"; done; SPECS="$SPECS lssd:fast:$n:6:lum"; done
for w in default 1 2 3; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 400 python scripts/time_variant.py $SPECS --steps 30 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
