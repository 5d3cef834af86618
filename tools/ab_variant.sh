#!/bin/bash
# Builds a measurement variant of libheifhip.so from the WORKING TREE with extra compiler flags, beside the tree (build/ab/<name>/libheif_amd/libheifhip.so;
# build/ travels to the GPU box): tools/gpu_call.sh ab:<name>[,<name>] measures the main workload under each.  usage: bash tools/ab_variant.sh <name> "<flags>"
set -e
name=$1; flags=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=$ROOT/build/ab/$name
mkdir -p "$out/libheif_amd" "$out/obj"
make -s -j8 -C "$ROOT/libheif_amd/csrc" OUT="$out/libheif_amd/libheifhip.so" OBJDIR="$out/obj" EXTRA="$flags"
echo "$flags" > "$out/FLAGS"
ls -la "$out/libheif_amd/libheifhip.so"
