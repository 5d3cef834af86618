#!/bin/bash
# ThreadSanitizer over the plugin INSIDE the real libheif (CPU only): the whole library for the host with its host sources under -fsanitize=thread
# (as tools/emu_tsan_host.sh) as a shared object, loaded with heif_load_plugin() into oracle/_ref/libheif*.so by tools/dropin_host.c compiled with the
# sanitizer: T application threads x heif_decode_image() on small HEIC files (golden stills, one of them as a 2 x 3 grid item) - the plugin's function
# table, plane hand-over into heif_image, both coalescers' still path; with RGB=1 through the patched libheif (the integration colour op -> hipdec_color_convert,
# resident planes / RGB, the grid hook).  libheif itself is not instrumented: only reports that name the plugin's sources count.
# The golden tracks go along as image-sequence files (moov / trak): heif_track_decode_next_image() plays them - Track_Visual pushes the samples into the plugin,
# look-ahead chains, the chain coalescer (TRACKS=0 leaves them out).
# usage: [RGB=0|1] [TRACKS=0] bash tools/emu_tsan_libheif.sh [threads] [seconds]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${TSAN_HOST_BUILD:-${TMPDIR:-/tmp}/hipdec_sanitizer_builds/thread-host}
SAN=${SAN:-thread}
[ "$SAN" = "thread" ] || B=${TSAN_HOST_BUILD:-${TMPDIR:-/tmp}/hipdec_sanitizer_builds/$SAN-host}
. $ROOT/tools/emu_tsan_objects.sh
mkdir -p $B/files
g++ -shared -fsanitize=$SAN -o $B/libheifhip_emu_tsan.so $B/obj/*.o -lpthread -ldl
gcc -O1 -g -fsanitize=$SAN -pthread $ROOT/tools/dropin_host.c $E/tsan_clockwait.c -ldl -lstdc++ -o $B/dropin_host_tsan   # (libstdc++ at start-up: AddressSanitizer resolves __cxa_throw when it initialises)
cd $ROOT
python - "$B/files" <<'PY'
import glob, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import heic_util
from oracle import pyoracle as orc
out = sys.argv[1]
n = 0
for f in sorted(glob.glob("tests/golden/*.hevc")):
    name = os.path.basename(f)
    if "reject" in name or name.startswith("ref_") or name.startswith("c4"):
        continue
    s = open(f, "rb").read()
    info = orc.decode(s)
    w, h, bd, cf = info["width"], info["height"], info["bit_depth_luma"], info["chroma_format_idc"]
    open(os.path.join(out, "%02d_%s.heic" % (n, name[:-5])), "wb").write(heic_util.build_heic([(s, w, h, cf)], bit_depth=bd, chroma_format_idc=cf))
    n += 1
    if name.startswith("default_"):
        open(os.path.join(out, "%02d_grid_2x3.heic" % n), "wb").write(heic_util.build_heic([(s, w, h, cf)] * 6, grid=(2, 3, 3 * w - 8, 2 * h - 4), bit_depth=bd, chroma_format_idc=cf))
        n += 1
if os.environ.get("TRACKS", "1") != "0":   # image-sequence files out of the golden tracks: heif_track_decode_next_image() on every picture
    import json
    idx = json.load(open("tests/golden/golden_sequences.json"))
    for name in sorted(idx):
        g = idx[name]
        blob = open("tests/golden/seq_%s.hevcs" % name, "rb").read()
        aus, p = [], 0
        while p < len(blob):
            k = int.from_bytes(blob[p:p + 4], "big"); aus.append(blob[p + 4:p + 4 + k]); p += 4 + k
        open(os.path.join(out, "%02d_seq_%s.heic" % (n, name)), "wb").write(
            heic_util.build_sequence(aus, g["width"], g["height"], bit_depth=g["bit_depth"], chroma_format_idc=g["chroma_format_idc"]))
        n += 1
print("%d HEIC files" % n)
PY
# FILES=<directory>: decode those files instead (e.g. the reference's fuzzing corpus with DROPIN_TOLERATE=1 and SAN=address)
[ -n "$FILES" ] && { rm -f $B/files/*; cp $FILES/* $B/files/; }
LIBHEIF=$ROOT/oracle/_ref/libheif.so
[ "${RGB:-0}" = "1" ] && LIBHEIF=$ROOT/oracle/_ref/libheif_hipcolor.so
# LIBHEIF_OVERRIDE: another build of libheif, e.g. one compiled with the sanitizer as well (make -f oracle/Makefile.ref OUT=<dir outside the tree> CXXFLAGS="... -fsanitize=thread"
# <dir>/libheif_hipcolor.so): then the integration sources (libheif_amd/integration/*.cc, compiled INTO libheif) and libheif's own threads are watched too
[ -n "$LIBHEIF_OVERRIDE" ] && LIBHEIF=$LIBHEIF_OVERRIDE
HIPEMU_THREADS=1 HIPEMU_DEVICES=${HIPEMU_DEVICES:-2} TSAN_OPTIONS="halt_on_error=0 history_size=4" \
  $B/dropin_host_tsan $LIBHEIF $B/libheifhip_emu_tsan.so ${1:-8} ${2:-10} ${RGB:-0} $B/files/*
