#!/bin/bash
# ThreadSanitizer over the product's HOST code (CPU only): decoder.hip (both coalescers, look-ahead chains, DPB holds, the resident-plane registry),
# runtime.hip (pools), hevc_headers.hip, batch_layout.hip, plugin.hip, grid_rccl.hip compiled with -fsanitize=thread and linked with the kernels under
# the SIMT emulator - those WITHOUT the sanitizer: a workgroup's lanes are ucontext coroutines, which ThreadSanitizer cannot follow; HIPEMU_THREADS=1
# runs every launch on the launching thread - and with tests/emu/tsan_host.cc, a multi-threaded C++ host on the C ABI (stills and tracks of tests/golden
# decoded side by side by T threads, a share of them damaged, every picture compared with a serial pass).
# usage: bash tools/emu_tsan_host.sh [threads] [rounds] [seed] [damaged_percent]      (SAN=address|undefined builds the same host with another sanitizer)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=${SAN:-thread}
B=${TSAN_HOST_BUILD:-$ROOT/build/$SAN-host}
mkdir -p $B
C=$ROOT/libheif_amd/csrc; E=$ROOT/tests/emu
FLAGS="-std=c++17 -fPIC -Wno-unknown-pragmas -fno-strict-aliasing -w -DHIPDEC_HOST_EMU=1 -DHIPDEC_PARSE_INTER=1 -DHIPDEC_NO_RCCL_HEADER -DHIPEMU_WHOLE_LIBRARY -I$E/shim -I$E -I$ROOT/include -I$C"
pids=()
for f in $E/parse_emu.cc $E/pipeline_emu.cc $E/color_emu.cc $C/residual_kernel.hip $C/recon_kernel.hip $C/filter_kernels.hip $C/color.hip $C/transform.hip $C/inter_kernels.hip; do
  g++ -O2 -g $FLAGS -c -x c++ $f -o $B/$(basename $f).o & pids+=($!)
done
for f in $C/hevc_headers.hip $C/batch_layout.hip $C/decoder.hip $C/runtime.hip $C/plugin.hip $C/grid_rccl.hip $E/tsan_host.cc; do
  g++ -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer $FLAGS -c -x c++ $f -o $B/$(basename $f).o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -fsanitize=$SAN -o $B/tsan_host $B/*.o -lpthread -ldl
cd $ROOT
HIPEMU_THREADS=1 TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4" $B/tsan_host $ROOT/tests/golden "${@}"
