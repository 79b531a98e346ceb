#!/bin/bash
# Experiment, same box, same library: time-paced issue priority in the pipelined Basic kernel (FTK_KLT_PACE = ns per position unit; unset = off)
for pace in off 600 800 1000 1100 1200 1400 1800; do
  echo "=== pace $pace"
  if [ $pace = off ]; then unset FTK_KLT_PACE; else export FTK_KLT_PACE=$pace; fi
  python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:2000:6 basic:inverse:1000:10 --steps 200 || exit 1
done
