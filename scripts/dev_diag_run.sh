#!/bin/bash
# Dev: per-layer times (random operands / zero filters) of the library and of the builds named on the command line
cd "$(dirname "$0")/.."
for lib in "" "$@"; do
  echo "== ${lib:-current}"
  RGP_DEV_LIB=$lib python scripts/dev_zero_input.py 1024 random,zero-filters 2>&1 | grep -v amdgpu.ids | head -2
done
