#!/bin/bash
# Regenerates the Panama FFM bindings of libjlamahip.so, exactly as the reference generates NativeSimd / NativeGPU
# (jlama-native/src/main/c/simd/jextract_vector_simd.sh:6-28).  Run from a Jlama checkout with this directory copied to
# jlama-native/src/main/c/hip/ next to include/jlama_hip.h and libjlamahip.so; needs jextract-22.
#
# The hand-written cnative/NativeHip.java in this directory is what jextract emits for the subset of entry points the
# provider uses (downcall handles over SymbolLookup.loaderLookup()); regenerate it with this script when the header grows.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
HDR=${JLAMA_HIP_HEADER:-$HERE/../include/jlama_hip.h}
OUT=${1:-$HERE/src/main/java22}
${JEXTRACT:-/usr/local/jextract-22/bin/jextract} \
  --output "$OUT" \
  -t com.github.tjake.jlama.tensor.operations.cnative \
  -I "$(dirname "$HDR")" \
  -l jlamahip \
  --header-class-name NativeHip \
  "$HDR"
