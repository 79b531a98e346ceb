#!/bin/bash
# The same switch on the synthetic scene (every feature takes the same few iterations) and below 1 024 features
for m in 4096 1024 512; do
  export FTK_KLT_SCHED_MIN=$m
  echo "--- synthetic sched_min=$m"
  python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:2000:6 basic:direct:2000:6 basic:fast:2000:6 affine:inverse:2000:6 affine:direct:2000:6 affine:fast:2000:6 lssd:inverse:2000:6 lssd:direct:2000:6 lssd:fast:2000:6 lssd:fast:2000:6:lum basic:inverse:1200:6 lssd:fast:1200:6 affine:fast:1200:6 --steps 100 --no-oracle || exit 1
  echo "--- real600 sched_min=$m"
  python scripts/time_variant.py basic:inverse:600:6 basic:direct:600:6 basic:fast:600:6 affine:inverse:600:6 affine:direct:600:6 affine:fast:600:6 lssd:inverse:600:6 lssd:direct:600:6 lssd:fast:600:6 --real --steps 60 --no-oracle || exit 1
done
