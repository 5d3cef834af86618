#!/bin/bash
# A stateful random walk over the decoder object's C ABI under a sanitizer (CPU only): tests/emu/api_fuzz.cc linked with the whole library for the host
# (tools/emu_tsan_objects.sh; SAN=address by default, ALL=1: kernels instrumented too).  Legal calls in any order - poll before push, push after flush,
# samples of other streams, random bytes, decode() between polls, plane reads at any time, instances dropped half way - must be answered, never crash.
# usage: [SAN=address|thread|undefined] bash tools/emu_api_fuzz.sh <seed> <walks> [steps per walk] [threads]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=${SAN:-address}
ALL=${ALL:-1}
B=${TSAN_HOST_BUILD:-${TMPDIR:-/tmp}/hipdec_sanitizer_builds/$SAN-host}   # (outside the tree: ~100 MB of objects per sanitizer must not travel with the repository)
mkdir -p $B
. $ROOT/tools/emu_tsan_objects.sh
g++ -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer $FLAGS -c $E/api_fuzz.cc -o $B/api_fuzz.o
CW=""; [ "$SAN" = "thread" ] && { gcc -O1 -fsanitize=thread -c $E/tsan_clockwait.c -o $B/tsan_clockwait.o; CW=$B/tsan_clockwait.o; }
g++ -fsanitize=$SAN -o $B/api_fuzz $B/api_fuzz.o $CW $B/obj/*.o -lpthread -ldl
cd $ROOT
HIPEMU_THREADS=${HIPEMU_THREADS:-2} ASAN_OPTIONS=detect_leaks=1:detect_stack_use_after_return=0 TSAN_OPTIONS="halt_on_error=0 history_size=4" \
  timeout ${FUZZ_TIMEOUT:-3000} $B/api_fuzz $ROOT/tests/golden "${@}"
