"""Memory-safety fuzz of the P-picture device code on the CPU (MODE=inter bash tools/emu_asan_fuzz.sh <seed> <count>): short IPP
sequences from the test generator, the P pictures randomly corrupted (slice header, motion syntax, residuals), decoded by the
ASAN-instrumented host build of the kernels with the decoder's DPB logic around them (tests/emu: emu_seq_*).  Every picture must end in a
host rejection or a device status word, never in an out-of-bounds access; a picture that failed is not committed as a reference."""
import sys, ctypes as C, random, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import pyoracle as orc
from test_inter_oracle import make_frames
L = C.CDLL(os.path.join(ROOT, 'build/asan/libparse_emu_asan.so'))
L.emu_seq_new.restype = C.c_void_p
L.emu_seq_free.argtypes = [C.c_void_p]
L.emu_seq_create_picture.restype = C.c_void_p
L.emu_seq_create_picture.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
L.emu_seq_commit.argtypes = [C.c_void_p, C.c_void_p]
L.emu_run_parse.argtypes = [C.c_void_p]
L.emu_run_pipeline.argtypes = [C.c_void_p, C.c_int]
L.emu_free.argtypes = [C.c_void_p]
L.emu_seq_create_chain.restype = C.c_void_p
L.emu_seq_create_chain.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.emu_seq_commit_chain.argtypes = [C.c_void_p, C.c_void_p]
L.emu_run_pipeline_chain.argtypes = [C.c_void_p]
L.emu_num_items.argtypes = [C.c_void_p]


L.emu_seq_create_chains.restype = C.c_void_p
L.emu_seq_create_chains.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.emu_seq_commit_chains.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]


def parameter_sets(au):
    out, p = b"", 0
    while p + 4 <= len(au):
        n = int.from_bytes(au[p:p + 4], "big")
        if 32 <= ((au[p + 4] >> 1) & 63) <= 34:
            out += au[p:p + 4 + n]
        p += 4 + n
    return out


def run(aus, chain=False):
    ps = parameter_sets(aus[0])
    q = C.c_void_p(L.emu_seq_new())
    res = []
    try:
        if chain:   # the look-ahead's form (round 5): the first picture alone, everything behind it as ONE chain batch (references inside the batch)
            err = C.create_string_buffer(512)
            b = L.emu_seq_create_picture(q, aus[0], len(aus[0]), err, 512)
            if not b:
                return 'rejected'
            b = C.c_void_p(b)
            if L.emu_run_parse(b) or L.emu_run_pipeline(b, 15):
                return 'status'
            L.emu_seq_commit(q, b)
            group = [ps + a for a in aus[1:]]
            ptrs = (C.c_char_p * len(group))(*group)
            sizes = (C.c_size_t * len(group))(*[len(a) for a in group])
            b = L.emu_seq_create_chain(q, len(group), ptrs, sizes, err, 512)
            if not b:
                return 'ok,chain rejected'
            b = C.c_void_p(b)
            st = 0
            if L.emu_num_items(b):
                st = L.emu_run_parse(b)
                if st == 0:
                    st = L.emu_run_pipeline_chain(b)
            if st == 0:
                L.emu_seq_commit_chain(q, b)
            return 'ok,chain ' + ('ok' if st == 0 else 'status')
        for i, au in enumerate(aus):
            data = au if i == 0 else ps + au
            err = C.create_string_buffer(512)
            b = L.emu_seq_create_picture(q, data, len(data), err, 512)
            if not b:
                res.append('rejected')
                continue
            b = C.c_void_p(b)
            st = L.emu_run_parse(b)
            if st == 0:
                st = L.emu_run_pipeline(b, 15)
            if st == 0:
                L.emu_seq_commit(q, b)     # (the sequence keeps the batch: its planes are reference pictures)
                res.append('ok')
            else:
                res.append('status')      # (the sequence object owns every batch it created; a failed picture is simply not committed)
    finally:
        L.emu_seq_free(q)
    return ','.join(res)


def run_tracks(tracks):
    """the chain coalescer's form: every track's first picture alone, then the rest of ALL tracks as one batch (layout_batch_plan_chains)"""
    qs = [C.c_void_p(L.emu_seq_new()) for _ in tracks]
    shared = None
    try:
        err = C.create_string_buffer(512)
        data, first, count = [], [], []
        for q, aus in zip(qs, tracks):
            b = L.emu_seq_create_picture(q, aus[0], len(aus[0]), err, 512)
            if not b:
                return 'tracks: first rejected'
            b = C.c_void_p(b)
            if L.emu_run_parse(b) or L.emu_run_pipeline(b, 15):
                return 'tracks: first status'
            L.emu_seq_commit(q, b)
            ps = parameter_sets(aus[0])
            group = [ps + a for a in aus[1:]]
            first.append(len(data)); count.append(len(group)); data += group
        n = len(tracks)
        qarr = (C.c_void_p * n)(*qs)
        b = L.emu_seq_create_chains(qarr, n, (C.c_int * n)(*first), (C.c_int * n)(*count), (C.c_char_p * len(data))(*data),
                                    (C.c_size_t * len(data))(*[len(a) for a in data]), err, 512)
        if not b:
            return 'tracks: rejected'
        shared = C.c_void_p(b)
        st = 0
        if L.emu_num_items(shared):
            st = L.emu_run_parse(shared)
            if st == 0:
                st = L.emu_run_pipeline_chain(shared)
        if st == 0:
            L.emu_seq_commit_chains(qarr, n, shared)
        return 'tracks: ' + ('ok' if st == 0 else 'status')
    finally:
        for q in qs:
            L.emu_seq_free(q)
        if shared:
            L.emu_free(shared)


rng = random.Random(int(sys.argv[1]))
n = int(sys.argv[2])
cfgs = [dict(), dict(amp=1, inter_num_refs=2), dict(inter_num_refs=3, lists_modification=1, max_merge_cand=3), dict(wpp=0, tile_cols=2, tile_rows=2),
        dict(num_slices=3, parallel_merge_level=4), dict(bit_depth=10, amp=1), dict(lossless_pct=20, transform_skip=1, log2_ctb=5),
        dict(pcm_pct=10, inter_intra_pct=40, cu_qp_delta=1), dict(dependent_segments=3, num_slices=2, wpp=0),
        dict(b_frames=1, temporal_mvp=1), dict(b_frames=2, b_ref=1, temporal_mvp=1, weighted_pred=1, inter_num_refs=2, mvd_l1_zero=1),
        dict(temporal_mvp=1, weighted_pred=1, inter_num_refs=3), dict(b_frames=1, weighted_pred=1, lists_modification=1, num_slices=2, amp=1),
        # round 5: long-term reference pictures (three syntax forms), constrained intra prediction, scaling lists in P / B pictures
        dict(long_term_ref=1, inter_num_refs=2, temporal_mvp=1), dict(long_term_ref=2, b_frames=1, temporal_mvp=1), dict(long_term_ref=3, lists_modification=1, inter_num_refs=3),
        dict(constrained_intra_pred=1, inter_intra_pct=45, log2_ctb=4, log2_max_tb=4), dict(scaling_list=2, inter_intra_pct=30, b_frames=1), dict(scaling_list=3, temporal_mvp=1)]
base = []
for i, c in enumerate(cfgs):
    frames = make_frames(104, 72, 4 if c.get('b_frames') else 3, c.get('bit_depth', 8))
    base.append(orc.encode_sequence(frames, qp=26, global_mv_x=-6, global_mv_y=4, seed=11 + i, **c))
print('clean:', [run(a) for a in base]); sys.stdout.flush()
print('clean (chains):', [run(a, True) for a in base]); sys.stdout.flush()
res = {}
for it in range(n):
    src = rng.choice(base)
    aus = [bytes(a) for a in src]
    victim = rng.randrange(1, len(aus))
    s = bytearray(aus[victim])
    for _ in range(rng.choice([1, 1, 2, 4, 16])):
        # the first bytes are the slice header (reference picture set, list sizes, merge candidates): hit them more often
        p = rng.randrange(4, min(len(s), 24)) if rng.random() < 0.35 else rng.randrange(len(s))
        mode = rng.randrange(3)
        if mode == 0: s[p] ^= 1 << rng.randrange(8)
        elif mode == 1: s[p] = rng.randrange(256)
        else: s[p] = 0xff
    aus[victim] = bytes(s)
    if it % 3 == 1:                         # every third case beside two clean tracks of the same bit depth in ONE batch
        same = [b for i, b in enumerate(base) if (cfgs[i].get('bit_depth', 8) > 8) == (cfgs[[id(x) for x in base].index(id(src))].get('bit_depth', 8) > 8)]
        r = run_tracks([rng.choice(same), aus, rng.choice(same)])
    else:
        r = run(aus, chain=(it % 3 == 2))   # every third case through the chain form
    res[r] = res.get(r, 0) + 1
print(res)
