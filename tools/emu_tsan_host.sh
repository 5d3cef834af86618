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
B=${TSAN_HOST_BUILD:-${TMPDIR:-/tmp}/hipdec_sanitizer_builds/$SAN-host}   # (outside the tree: ~100 MB of objects per sanitizer must not travel with the repository)
mkdir -p $B
. $ROOT/tools/emu_tsan_objects.sh
g++ -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer $FLAGS -c $E/tsan_host.cc -o $B/tsan_host.o
g++ -fsanitize=$SAN -o $B/tsan_host $B/tsan_host.o $B/obj/*.o -lpthread -ldl
cd $ROOT
HIPEMU_THREADS=1 TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4" $B/tsan_host $ROOT/tests/golden "${@}"
