#!/bin/bash
# AddressSanitizer over the product's HOST code: the whole library compiled for the host (tests/emu: decoder.hip, runtime.hip, plugin.hip, grid_rccl.hip +
# kernels under the SIMT emulator) with -fsanitize=address, then the GPU tier's host-orchestration tests against it - launch sets, look-ahead chains, both
# coalescers with their threads, DPB holds, plugin calls through the real libheif.  usage: bash tools/emu_asan_whole_library.sh [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/asan
C=$ROOT/libheif_amd/csrc; E=$ROOT/tests/emu
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Wno-unknown-pragmas -fno-strict-aliasing -w -DHIPDEC_HOST_EMU=1 -DHIPDEC_PARSE_INTER=1 \
    -DHIPDEC_NO_RCCL_HEADER -DHIPEMU_WHOLE_LIBRARY -I$E/shim -I$E -I$ROOT/include -I$C -shared -o $ROOT/build/asan/libheifhip_emu_asan.so \
    $E/parse_emu.cc $E/pipeline_emu.cc $E/color_emu.cc -x c++ $C/hevc_headers.hip $C/batch_layout.hip $C/residual_kernel.hip $C/recon_kernel.hip \
    $C/filter_kernels.hip $C/color.hip $C/transform.hip $C/inter_kernels.hip $C/decoder.hip $C/runtime.hip $C/plugin.hip $C/grid_rccl.hip -lpthread -ldl
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 \
  HIPDEC_LIBRARY=$ROOT/build/asan/libheifhip_emu_asan.so HIPDEC_DEV_AB=1 python -m pytest -m gpu -q --timeout 1500 -p no:cacheprovider \
  ${@:-tests/test_sequence_gpu.py tests/test_golden_sequences.py tests/test_plugin_dropin.py tests/test_resident_planes_gpu.py}
