#!/bin/bash
# Dev: per-layer times (random operands / zero filters) of a DEV=1 build (prev/dev.so) under RGP_CP_ABLATE values
cd "$(dirname "$0")/.."
for a in "$@"; do
  echo "== RGP_CP_ABLATE=$a"
  RGP_CP_ABLATE=$a RGP_DEV_LIB=prev/dev.so python scripts/dev_zero_input.py 1024 random,zero-filters 2>&1 | grep -v amdgpu.ids | head -2
done
