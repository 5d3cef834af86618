# sourced by tools/emu_tsan_host.sh and tools/emu_tsan_libheif.sh: the objects of the whole library for the host in $B - kernels and emulator support
# WITHOUT a sanitizer (ucontext lanes), the product's host sources with -fsanitize=$SAN.  An object newer than every source is kept.
C=$ROOT/libheif_amd/csrc; E=$ROOT/tests/emu
FLAGS="-std=c++17 -fPIC -Wno-unknown-pragmas -fno-strict-aliasing -w -DHIPDEC_HOST_EMU=1 -DHIPDEC_PARSE_INTER=1 -DHIPDEC_NO_RCCL_HEADER -DHIPEMU_WHOLE_LIBRARY -I$E/shim -I$E -I$ROOT/include -I$C"
[ "${ALL:-0}" = "1" ] && B=$B/all
mkdir -p $B/obj
fresh() { [ -f "$1" ] && [ -z "$(find $C $E $ROOT/include -maxdepth 3 \( -name '*.hip' -o -name '*.h' -o -name '*.cc' \) -newer "$1" -print -quit)" ]; }
pids=()
for f in $E/parse_emu.cc $E/pipeline_emu.cc $E/color_emu.cc $C/residual_kernel.hip $C/recon_kernel.hip $C/filter_kernels.hip $C/color.hip $C/transform.hip $C/inter_kernels.hip; do
  o=$B/obj/$(basename $f).o
  # ALL=1: the kernel translation units with the sanitizer as well - __global__ / __device__ functions are no_sanitize_thread in the shim, so this adds
  # the HOST functions that live beside the kernels (launchers, the colour planner and its capture state, hipdec_color_* / hipdec_image_transform)
  case "$f" in *.hip) [ "${ALL:-0}" = "1" ] && KSAN="-fsanitize=$SAN -fno-omit-frame-pointer" || KSAN="";; *) KSAN="";; esac
  fresh $o || { g++ -O2 -g $KSAN $FLAGS -c -x c++ $f -o $o & pids+=($!); }
done
for f in $C/hevc_headers.hip $C/batch_layout.hip $C/decoder.hip $C/runtime.hip $C/plugin.hip $C/grid_rccl.hip; do
  o=$B/obj/$(basename $f).o
  fresh $o || { g++ -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer $FLAGS -c -x c++ $f -o $o & pids+=($!); }
done
for p in "${pids[@]}"; do wait $p; done
