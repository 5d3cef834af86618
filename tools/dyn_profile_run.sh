#!/bin/bash
# Estimated dynamic instruction profile of a kernel on the CPU (no GPU needed): builds the host emulation with --coverage under build/cov, runs
# one 1080p still of the bench's content through it, gcov's the kernel source and weights the static ISA of <kernel> with the line counts
# (tools/dyn_profile.py).  usage: dyn_profile_run.sh [parse|recon|residual] ;  env: SALU=1 (scalar instructions only), TOPN=<lines>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/libheif_amd/csrc; E=$ROOT/tests/emu; B=$ROOT/build/cov
what=${1:-parse}
mkdir -p $B; cd $B; rm -f *.gcda *.gcov
if [ ! -f libparse_emu_cov.so ] || [ -n "$(find $C $E -newer libparse_emu_cov.so -name '*.h' -o -newer libparse_emu_cov.so -name '*.hip' -o -newer libparse_emu_cov.so -name '*.cc' | head -1)" ]; then
  rm -f *.gcno
  g++ -O1 --coverage -std=c++17 -fPIC -Wno-unknown-pragmas -fno-strict-aliasing -DHIPDEC_HOST_EMU=1 -DHIPDEC_PARSE_INTER=0 -I$E/shim -I$E -I$ROOT/include -I$C -shared \
    -o libparse_emu_cov.so $E/parse_emu.cc $E/pipeline_emu.cc $E/color_emu.cc -x c++ $C/hevc_headers.hip $C/batch_layout.hip $C/transform.hip \
    $C/residual_kernel.hip $C/recon_kernel.hip $C/filter_kernels.hip $C/color.hip $C/inter_kernels.hip -lpthread 2>&1 | grep -E "error" || true
fi
cat > runcov.py <<PY
import ctypes as C, sys
sys.path.insert(0, "$ROOT")
from tools import streamgen
s = streamgen.make_stream(1920, 1080, seed=1001, bit_depth=8, wpp=1, qp=27)
L = C.CDLL("$B/libparse_emu_cov.so")
L.emu_create.restype = C.c_void_p
L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.emu_run_parse.argtypes = [C.c_void_p]; L.emu_run_pipeline.argtypes = [C.c_void_p, C.c_int]
for f in ():
    if hasattr(L, f): getattr(L, f).argtypes = [C.c_void_p]
arr = (C.c_char_p * 1)(s); sizes = (C.c_size_t * 1)(len(s)); err = C.create_string_buffer(512)
h = L.emu_create(1, arr, sizes, err, 512); assert h, err.value
print("parse status", L.emu_run_parse(h))
if "$what" != "parse": print("pipeline status", L.emu_run_pipeline(h, 15))
PY
python runcov.py
case $what in
  parse) gcov -o . libparse_emu_cov.so-parse_emu.gcda > /dev/null 2>&1; python $ROOT/tools/dyn_profile.py $C/parse_kernel.hip k_parse_occ8 parse_core.h $B/parse_core.h.gcov 2073600;;
  recon) gcov -o . libparse_emu_cov.so-recon_kernel.gcda > /dev/null 2>&1; python $ROOT/tools/dyn_profile.py $C/recon_kernel.hip k_recon8 recon_kernel.hip $B/recon_kernel.hip.gcov 2073600;;
  residual) gcov -o . libparse_emu_cov.so-residual_kernel.gcda > /dev/null 2>&1; python $ROOT/tools/dyn_profile.py $C/residual_kernel.hip k_residual residual_kernel.hip $B/residual_kernel.hip.gcov 2073600;;
esac
