#!/bin/bash
# Dev: builds prev/diag<N>.so = the library with rgp_conv_patch.hip compiled under -DRGP_CP_DIAG=<N> (timing ablations of
# the patch kernels' K loop; results garbage).  usage: scripts/dev_build_diag.sh 1 2 4 ...
set -e
cd "$(dirname "$0")/../recurrent_gaze_prediction_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../prev
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -DRGP_CP_DIAG=$n -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include \
      -c rgp_conv_patch.hip -o /tmp/rgp_conv_patch_diag$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v rgp_conv_patch.o) /tmp/rgp_conv_patch_diag$n.o -o ../../prev/diag$n.so ) &
done
wait
ls -la ../../prev/
