# builds the ASAN-instrumented host emulation of the device kernels and runs tools/emu_asan_fuzz.py under it
# usage: [MODE=headers|inter] bash tools/emu_asan_fuzz.sh <seed> <count>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/asan
C=$ROOT/libheif_amd/csrc; E=$ROOT/tests/emu
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Wno-unknown-pragmas -fno-strict-aliasing -DHIPDEC_HOST_EMU=1 -DHIPDEC_PARSE_INTER=1 \
    -I$E/shim -I$E -I$ROOT/include -I$C -shared -o $ROOT/build/asan/libparse_emu_asan.so $E/parse_emu.cc $E/pipeline_emu.cc $E/color_emu.cc \
    -x c++ $C/hevc_headers.hip $C/batch_layout.hip $C/transform.hip $C/residual_kernel.hip $C/recon_kernel.hip $C/filter_kernels.hip $C/color.hip $C/inter_kernels.hip -lpthread
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python $ROOT/tools/emu_asan_fuzz${MODE:+_$MODE}.py ${1:-1} ${2:-200}
