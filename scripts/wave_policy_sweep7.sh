#!/bin/bash
# Experiment (GPU box): the wave policy at small patches: 9 x 9 (81 pixels) and 11 x 11 (121), every variant x 600 ... 6 000 features x 1 - 3 waves.
SPECS=""
for h in 4 5; do for n in 600 1500 3000 6000; do for mm in lssd:fast lssd:direct lssd:inverse affine:inverse affine:direct affine:fast basic:direct basic:fast basic:inverse; do SPECS="$SPECS $mm:$n:$h"; done; done; done
for w in default 1 2 3; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 600 python scripts/time_variant.py $SPECS --steps 20 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
