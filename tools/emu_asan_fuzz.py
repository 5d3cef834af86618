"""Memory-safety fuzz of the DEVICE code on the CPU: the kernel sources compiled for the host (tests/emu) with AddressSanitizer,
fed randomly corrupted streams (bit flips, random bytes, 0xff bytes in slice data and headers).  Every run must end in a host
rejection or a device status word - never in an out-of-bounds access.  Dev tool:
    bash tools/emu_asan_fuzz.sh <seed> <count>"""
import sys, ctypes as C, random, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import pyoracle as orc
L = C.CDLL(os.path.join(ROOT, 'build/asan/libparse_emu_asan.so'))
L.emu_create.restype = C.c_void_p
L.emu_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.emu_free.argtypes=[C.c_void_p]; L.emu_upload_hash.restype=C.c_uint64; L.emu_upload_hash.argtypes=[C.c_void_p]; L.emu_run_parse.argtypes=[C.c_void_p]; L.emu_run_pipeline.argtypes=[C.c_void_p, C.c_int]
def run(s):
    arr=(C.c_char_p*1)(s); sizes=(C.c_size_t*1)(len(s)); err=C.create_string_buffer(512)
    h=L.emu_create(1,arr,sizes,err,512)
    if not h: return 'rejected: '+err.value.decode()[:60]
    h0=L.emu_upload_hash(h)
    st=L.emu_run_parse(h)
    if st==0: st=L.emu_run_pipeline(h,15)
    assert L.emu_upload_hash(h)==h0, "a kernel wrote into the read-only upload region"
    L.emu_free(h)
    return 'status 0x%x'%(st&0xffffffff)
rng=random.Random(int(sys.argv[1]))
n=int(sys.argv[2])
cfgs=[dict(scaling_list=2, stress=1), dict(scaling_list=3), dict(), dict(stress=1, num_slices=2), dict(log2_ctb=4,log2_min_cb=3,log2_max_tb=4,stress=1), dict(bit_depth=10), dict(wpp=0, transform_skip=1, lossless_pct=10), dict(pcm_pct=30, stress=1), dict(pcm_pct=25, pcm_loop_filter_disabled=1, bit_depth=10), dict(dependent_segments=3, wpp=0, stress=1), dict(dependent_segments=2, num_slices=2, wpp=1)]
base=[orc.encode(orc.synth_image(136,72,c.get('bit_depth',8),1,seed=3+i),**c) for i,c in enumerate(cfgs)]
# 4:2:2 and 4:4:4 (HIPDEC_FUZZ_CHROMA=2 / 3 / "all"): the same tool mixes minus what the format does not take
cf_env=os.environ.get('HIPDEC_FUZZ_CHROMA','')
if cf_env:
    fmts=[2,3] if cf_env=='all' else [int(cf_env)]
    base=[] if cf_env!='all' else base
    for cf in fmts:
        for i,c in enumerate(cfgs):
            c=dict(c)
            base.append(orc.encode(orc.synth_image(136,72,c.get('bit_depth',8),cf,seed=3+i),**c))
print('clean:', [run(s) for s in base]); sys.stdout.flush()
res={}
for it in range(n):
    s=bytearray(rng.choice(base))
    k=rng.choice([1,1,2,4,16])
    for _ in range(k):
        p=rng.randrange(len(s)); 
        mode=rng.randrange(3)
        if mode==0: s[p]^=1<<rng.randrange(8)
        elif mode==1: s[p]=rng.randrange(256)
        else: s[p]=0xff
    r=run(bytes(s)); res[r.split(':')[0]]=res.get(r.split(':')[0],0)+1
print(res)
