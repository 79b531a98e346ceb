#!/bin/bash
# A / B, same box, same library: the launch's common clock (progress balance, klt_common.h) off (FTK_KLT_PROGRESS=0) and on (default)
for rep in 1 2; do
for sw in 0 1; do
  echo "=== progress $sw"
  FTK_KLT_PROGRESS=$sw python scripts/time_variant.py basic:inverse:2000:10 basic:inverse:200:5 basic:inverse:2000:6 basic:inverse:1000:10 basic:inverse:4000:10 basic:inverse:300:6 --steps 200 || exit 1
  echo "--- real"; FTK_KLT_PROGRESS=$sw python scripts/time_variant.py basic:inverse:300:6 basic:inverse:2000:6 --real --steps 200 || exit 1
  FTK_KLT_PROGRESS=$sw python scripts/time_variant.py basic:inverse:25000:6 --size 1920x1080 --steps 50 || exit 1
done
done
