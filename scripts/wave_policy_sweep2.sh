#!/bin/bash
# Experiment (GPU box): LSSD fast + luminance at small feature counts and LSSD direct at large ones, by waves per feature.
SPECS=""
for n in 200 400 600 800 1000; do SPECS="$SPECS lssd:fast:$n:6:lum"; done
for n in 2400 3000 4000 5000 6000 8000 10000; do SPECS="$SPECS lssd:direct:$n:6"; done
for w in default 1 2 3; do if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
  timeout -k 10 400 python scripts/time_variant.py $SPECS --steps 30 --no-oracle 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['spec'], d['us_per_step'])"
done
