#!/bin/bash
# Does the longest-first launch order pay below 4 096 features on real images (multi-wave features do not all fit the chip at once)?
for n in 1200 2000 3000; do
  for m in 4096 1024; do
    export FTK_KLT_SCHED_MIN=$m
    echo "--- n=$n sched_min=$m"
    python scripts/time_variant.py basic:inverse:$n:6 basic:direct:$n:6 basic:fast:$n:6 affine:inverse:$n:6 affine:direct:$n:6 affine:fast:$n:6 lssd:inverse:$n:6 lssd:direct:$n:6 lssd:fast:$n:6 --real --steps 60 --no-oracle || exit 1
  done
done
