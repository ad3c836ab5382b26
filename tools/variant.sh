#!/bin/bash
# Builds a VARIANT of libcrt.so beside the product library, never in place of it:
#   tools/variant.sh LABEL "EXTRA defines" [make arguments, e.g. EXPERIMENTS=1]   ->   variants/LABEL/libcrt.so
# The sources are copied to a scratch directory and built there, so caitlynrenderer_amd/libcrt.so and its objects are not touched
# (an interrupted A/B can no longer leave a variant behind as the tested library).  Run it HERE (hipcc cross-compiles gfx950 without a
# GPU); the .so travels to the GPU box with the snapshot, and CRT_LIB=variants/LABEL/libcrt.so points the binding at it (tools/ab_run.sh).
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
L=$1; EX=${2:-}; shift; shift || true
W=$(mktemp -d /tmp/crt_variant_XXXXXX)
trap 'rm -rf "$W"' EXIT
mkdir -p "$W/caitlynrenderer_amd" "$W/include" "$R/variants/$L"
if [ -n "${VARIANT_REF:-}" ]; then      # VARIANT_REF=<commit>: that commit's sources instead of the working tree's
  git -C "$R" archive "$VARIANT_REF" caitlynrenderer_amd/csrc include | tar -x -C "$W"
else
  cp -r "$R/caitlynrenderer_amd/csrc" "$W/caitlynrenderer_amd/csrc"
  cp "$R"/include/*.h "$W/include/"
fi
find "$W" -name '*.o' -delete
make -C "$W/caitlynrenderer_amd/csrc" -s -j4 EXTRA="$EX" "$@" > "$R/variants/$L/build.log" 2>&1 || { echo "variant $L: build FAILED"; tail -20 "$R/variants/$L/build.log"; exit 1; }
cp "$W/caitlynrenderer_amd/libcrt.so" "$R/variants/$L/libcrt.so"
echo "$EX $*" > "$R/variants/$L/defines.txt"
echo "variant $L: $(stat -c %s "$R/variants/$L/libcrt.so") bytes  [$EX $*]"
