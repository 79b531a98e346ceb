#!/bin/bash
# Real-image calls (a few never-converging features set the launch): which wave count per feature is fastest?
NS=${NS:-"300 1200 2000"}; WS=${WS:-"default 1 2 3 4"}
for n in $NS; do
  for w in $WS; do
    if [ $w = default ]; then unset FTK_KLT_WAVES; else export FTK_KLT_WAVES=$w; fi
    echo "--- n=$n waves=$w"
    python scripts/time_variant.py basic:inverse:$n:6 basic:direct:$n:6 affine:inverse:$n:6 affine:direct:$n:6 lssd:inverse:$n:6 lssd:direct:$n:6 lssd:fast:$n:6 --real --steps 60 --no-oracle || exit 1
  done
done
