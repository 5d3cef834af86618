#!/bin/bash
# tools/emu_tsan_libheif.sh with a libheif that is compiled with ThreadSanitizer AS WELL: the reference (patched: libheif_amd/integration/*.cc are
# compiled INTO it) is built by oracle/Makefile.ref into a directory outside the tree with -fsanitize=thread, so that the integration sources - the colour op,
# the transformation and grid hooks - and libheif's own threads are watched, not only the plugin.  Needs /root/reference (this container only).
# usage: [RGB=0|1] [DROPIN_COLD=1] bash tools/emu_tsan_libheif_full.sh [threads] [seconds]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=${REF:-/root/reference}
SAN=${SAN:-thread}     # (SAN=address with FILES=<dir> DROPIN_TOLERATE=1: damaged files through the instrumented libheif + integration + plugin)
O=${TMPDIR:-/tmp}/hipdec_sanitizer_builds/ref_$SAN
mkdir -p $O
make -s -j8 -f $ROOT/oracle/Makefile.ref OUT=$O OBJ=$O/obj CXX=g++ \
  CXXFLAGS="-std=c++20 -O1 -g -fsanitize=$SAN -fPIC -w -pthread -ffp-contract=off -DLIBHEIF_EXPORTS -DHAVE_VISIBILITY -DENABLE_PLUGIN_LOADING=1 -DENABLE_MULTITHREADING_SUPPORT=1 -DENABLE_PARALLEL_TILE_DECODING=1 -I$O/gen -I$O/gen/libheif -I$R/libheif -I$R/libheif/api -I$R" \
  $O/libheif_hipcolor.so
SAN=$SAN LIBHEIF_OVERRIDE=$O/libheif_hipcolor.so ALL=${ALL:-1} RGB=${RGB:-1} bash $ROOT/tools/emu_tsan_libheif.sh "$@"
