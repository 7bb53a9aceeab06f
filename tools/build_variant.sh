#!/bin/bash
# build_variant.sh NAME [EXTRA flags...]: build libeppm_hip.so with extra compiler flags into
# gpurun_variants/NAME/ (travels to the GPU box; git-ignored) for A/B timing with tools/gpu.sh ab / abstage / abk (the test library travels with it)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p gpurun_variants/$name
if [ -n "$TOL" ]; then      # TOL=1: a variant of the tolerance library (tools/gpu.sh ab with LIB=tol)
  make -C eppm_amd/csrc -j8 OUT=../../gpurun_variants/$name TOLFLAGS="$*" ../../gpurun_variants/$name/libeppm_hip_tol.so 2>&1 | grep -E "error|Error" || true
  ls -la gpurun_variants/$name/libeppm_hip_tol.so
  exit 0
fi
make -C eppm_amd/csrc -j8 OUT=../../gpurun_variants/$name EXTRA="$*" ../../gpurun_variants/$name/libeppm_hip.so ../../gpurun_variants/$name/libeppm_hip_test.so 2>&1 | grep -E "error|Error" || true
ls -la gpurun_variants/$name/libeppm_hip.so gpurun_variants/$name/libeppm_hip_test.so
