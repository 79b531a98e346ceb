#!/bin/bash
# Experiment (GPU box): the wave policy at 21 x 21 (P = 441 > 256 pixels): every variant x 1 200 ... 5 000 features x 1 - 4 waves per feature.
SPECS=""
for n in 1200 2000 3000 5000; do for mm in lssd:fast lssd:direct lssd:inverse affine:inverse affine:direct affine:fast basic:direct basic:fast; do SPECS="$SPECS $mm:$n:10"; done; SPECS="$SPECS lssd:fast:$n:10:lum"; done
for w in default 1 2 3 4; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 600 python scripts/time_variant.py $SPECS --steps 20 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
