#!/bin/bash
# The whole GPU tier against tests/emu/libheifhip_emu.so (the product's sources compiled for the host: tests/test_product_on_emulator.py) - no GPU needed.
# Full-size shapes (4K / 8K, 1024 x 1080p) are left out: minutes each under the emulator.  usage: bash tools/gpu_tier_on_emulator.sh [pytest args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $ROOT/tests/emu libheifhip_emu.so || exit 1
cd $ROOT && HIPDEC_LIBRARY=$ROOT/tests/emu/libheifhip_emu.so HIPDEC_DEV_AB=1 python -m pytest tests -m gpu -q -n ${EMU_WORKERS:-8} --timeout 1200 -p no:cacheprovider \
  --deselect tests/test_full_shape_gpu.py "$@"
