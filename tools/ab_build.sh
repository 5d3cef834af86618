#!/bin/bash
# Builds another branch's libheifhip.so beside the working tree (build/ab/<branch>/libheif_amd/libheifhip.so; build/ travels to the GPU box), so that
# tools/ab_bench.sh can measure it against the checked-out one in ONE gpurun call.  usage: bash tools/ab_build.sh [branch]   (default: candidates)
set -e
br=${1:-candidates}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=$ROOT/build/ab/$br
rm -rf "$out"; mkdir -p "$out"
git -C "$ROOT" archive "$br" libheif_amd/csrc include | tar -x -C "$out"
make -s -C "$out/libheif_amd/csrc" -j8
git -C "$ROOT" rev-parse --short "$br" > "$out/COMMIT"
ls -la "$out/libheif_amd/libheifhip.so"
