#!/bin/bash
# Experiment (GPU box): the wave policy between the swept sizes: 15 x 15 (225 pixels) and 17 x 17 (289), every variant x 1 200 / 2 000 / 3 000 features x 1 - 4 waves.
SPECS=""
for h in 7 8; do for n in 1200 2000 3000; do for mm in lssd:fast lssd:direct lssd:inverse affine:inverse affine:direct affine:fast basic:direct basic:fast basic:inverse; do SPECS="$SPECS $mm:$n:$h"; done; SPECS="$SPECS lssd:fast:$n:$h:lum"; done; done
for w in default 1 2 3 4; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 600 python scripts/time_variant.py $SPECS --steps 20 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
