"""Header fuzz under AddressSanitizer: parameter sets, slice headers and NAL length fields are corrupted (and streams
truncated); the HOST front end (hevc_headers.hip / batch_layout.hip, product code that runs on the CPU) must reject or accept
cleanly, and whatever it accepts goes through the emulated device kernels.  Run through tools/emu_asan_fuzz.sh (MODE=headers)."""
import sys, ctypes as C, random, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import pyoracle as orc
L = C.CDLL(os.path.join(ROOT, 'build/asan/libparse_emu_asan.so'))
L.emu_create.restype = C.c_void_p
L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.emu_free.argtypes=[C.c_void_p]; L.emu_upload_hash.restype=C.c_uint64; L.emu_upload_hash.argtypes=[C.c_void_p]; L.emu_run_parse.argtypes=[C.c_void_p]; L.emu_run_pipeline.argtypes=[C.c_void_p, C.c_int]; L.emu_info.argtypes=[C.c_void_p, C.c_int, C.POINTER(C.c_int)]
def run(s):
    arr=(C.c_char_p*1)(s); sizes=(C.c_size_t*1)(len(s)); err=C.create_string_buffer(512)
    h=L.emu_create(1,arr,sizes,err,512)
    if not h: return 0
    info=(C.c_int*7)(); L.emu_info(h,0,info)
    r=1
    if info[0]*info[1] <= 600*600:
        h0=L.emu_upload_hash(h)
        st=L.emu_run_parse(h)
        if st==0: st=L.emu_run_pipeline(h,15)
        assert L.emu_upload_hash(h)==h0, "a kernel wrote into the read-only upload region"
        r = 2 if st==0 else 1
    L.emu_free(h); return r
rng=random.Random(int(sys.argv[1])); n=int(sys.argv[2])
cfgs=[dict(scaling_list=2, stress=1), dict(scaling_list=3), dict(), dict(stress=1, num_slices=3), dict(log2_ctb=4,log2_min_cb=3,log2_max_tb=4), dict(bit_depth=10, vui_matrix=9,vui_primaries=9,vui_transfer=16), dict(wpp=0, transform_skip=1), dict(pcm_pct=30), dict(pcm_pct=25, pcm_loop_filter_disabled=1, bit_depth=10), dict(dependent_segments=3, wpp=0, stress=1), dict(dependent_segments=2, num_slices=2, wpp=1)]
base=[orc.encode(orc.synth_image(136,72,c.get('bit_depth',8),1,seed=3+i),**c) for i,c in enumerate(cfgs)]
import heic_util as hu, struct
ref='/root/reference/examples/example.heic'
if os.path.exists(ref):
    h=hu.HeicFile(ref); base.append(h.plugin_stream(h.hevc_items()[1]))
# the hand-written hostile parameter sets / slice headers of tests/test_hostile_headers.py, as they are and as fuzz seeds
import test_hostile_headers as hh
for name, stream in sorted(hh.HOSTILE.items()):
    assert run(stream) == 0, "hostile stream accepted: " + name
base += [hh.sps() + hh.pps(wpp=1) + hh.idr(entry=(0, 0, [])), hh.sps(width=128) + hh.pps(wpp=1, tiles=dict(cols_m1=1, rows_m1=0, uniform=0, col_w_m1=[0])) + hh.idr(entry=(1, 7, [5]))]
acc=0
for it in range(n):
    s=bytearray(rng.choice(base))
    # header area: parameter sets + first slice header bytes; also length fields
    lim=min(len(s), 160)
    for _ in range(rng.choice([1,1,2,3,8])):
        p=rng.randrange(lim); m=rng.randrange(4)
        if m==0: s[p]^=1<<rng.randrange(8)
        elif m==1: s[p]=rng.randrange(256)
        elif m==2: s[p]=0xff
        else: s[p]=0
    if rng.random()<0.1: s=s[:rng.randrange(4,len(s))]   # truncation
    r=run(bytes(s)); acc+= r>0; ok = ok+(r==2) if 'ok' in dir() else (r==2)
print('accepted', acc, 'of', n, 'decoded with status 0:', ok)
