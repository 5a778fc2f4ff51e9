#!/bin/bash
# Build an experiment variant of libplmpm.so:  build_variant.sh NAME "EXTRA FLAGS"  ->  exp_libs/libplmpm_NAME.so
# (objects in plasticinelab_amd/csrc/build_NAME; both git-ignored, both travel with gpurun).  Select it at run time
# with PLMPM_LIB=exp_libs/libplmpm_NAME.so -- A/B runs of one gpurun call use the same box, clocks and day.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/exp_libs"
make -s -j 6 -C "$ROOT/plasticinelab_amd/csrc" OBJDIR=build_$NAME OUT="$ROOT/exp_libs/libplmpm_$NAME.so" EXTRA="$*"
echo "built exp_libs/libplmpm_$NAME.so ($*)"
