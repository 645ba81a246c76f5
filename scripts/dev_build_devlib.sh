#!/bin/bash
# Dev: builds recurrent_gaze_prediction_amd/librgp_hip_dev.so = the library compiled with -DRGP_DEV_KNOBS (make DEV=1: the
# RGP_* development switches are read from the environment, extra kernel variants are instantiated) in a scratch copy of
# csrc/, without touching the product build's objects.  Run a script against it with scripts/dev_with_lib.py.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=${TMPDIR:-/tmp}/rgpdev
mkdir -p $W/pkg/csrc $W/include
cp $ROOT/include/rgp.h $W/include/
cp $ROOT/recurrent_gaze_prediction_amd/csrc/*.h $ROOT/recurrent_gaze_prediction_amd/csrc/*.hip $ROOT/recurrent_gaze_prediction_amd/csrc/Makefile $W/pkg/csrc/
make -C $W/pkg/csrc -j8 DEV=1 | grep -E "error|warning" || true
cp $W/pkg/librgp_hip.so $ROOT/recurrent_gaze_prediction_amd/librgp_hip_dev.so
ls -la $ROOT/recurrent_gaze_prediction_amd/librgp_hip_dev.so
