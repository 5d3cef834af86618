"""The decoder's sample queue on the CPU (libheif_amd/csrc/hevc_headers.h: SampleQueue, through the host build of tests/emu): what
hipdec_decoder_push_data does with pushed bytes before anything reaches the GPU.  libde265 takes any number of NAL units per push
(libheif/plugins/decoder_libde265.cc:322-368); libheif pushes a still in one piece and a track sample by sample with a user_data each
(codecs/decoder.cc:436-447, sequences/track_visual.cc:200-280); heif_plugin.h:113-115 allows several pushes per picture.  The queue splits all of
that into access units (7.4.2.4.4), keeps the parameter sets for the samples that come without them and attributes push_data2's user_data."""
import ctypes as C
import random
import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parse_emu import emu
from test_inter_oracle import make_frames


def _lib():
    L = emu()
    L.emu_sq_new.restype = C.c_void_p
    L.emu_sq_free.argtypes = [C.c_void_p]
    L.emu_sq_push.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint64]
    L.emu_sq_close_first.argtypes = [C.c_void_p]
    L.emu_sq_drop.argtypes = [C.c_void_p, C.c_size_t]
    L.emu_sq_count.argtypes = [C.c_void_p]
    L.emu_sq_get.restype = C.c_size_t
    L.emu_sq_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    return L


def nals(stream):
    out, p = [], 0
    while p + 4 <= len(stream):
        n = int.from_bytes(stream[p:p + 4], "big")
        out.append(stream[p:p + 4 + n])
        p += 4 + n
    return out


def is_ps(x):
    return 32 <= ((x[4] >> 1) & 63) <= 34


class Queue:
    def __init__(self):
        self.L = _lib()
        self.q = C.c_void_p(self.L.emu_sq_new())

    def push(self, data, user_data=None):
        return self.L.emu_sq_push(self.q, data, len(data), 0 if user_data is None else 1, 0 if user_data is None else user_data)

    def get(self, i):
        ud, vcl = C.c_uint64(0), C.c_int(0)
        n = self.L.emu_sq_get(self.q, i, None, 0, C.byref(ud), C.byref(vcl))
        buf = C.create_string_buffer(max(1, n))
        self.L.emu_sq_get(self.q, i, buf, n, None, None)
        return buf.raw[:n], ud.value, bool(vcl.value)

    def count(self):
        return self.L.emu_sq_count(self.q)

    def free(self):
        self.L.emu_sq_free(self.q)


@pytest.fixture(scope="module")
def track():
    return orc.encode_sequence(make_frames(72, 56, 6), qp=30, b_frames=1, num_slices=2, wpp=0)


def _check(q, aus, user_data=None):
    """first access unit = sample 0 as pushed; queued sample k = the parameter sets seen so far + the slice NAL units of sample k + 1"""
    first, ud0, vcl0 = q.get(-1)
    assert first == aus[0] and vcl0
    assert q.count() == len(aus) - 1
    ps = b"".join(x for x in nals(aus[0]) if is_ps(x))
    assert q.get(-2)[0] == ps
    for k in range(1, len(aus)):
        blob, ud, vcl = q.get(k - 1)
        assert blob == ps + aus[k] and vcl, "sample %d" % k
        if user_data is not None:
            assert ud == user_data[k]
    if user_data is not None:
        assert ud0 == user_data[0]


def test_one_push_per_sample_with_user_data(track):
    q = Queue()
    uds = [1000 + 7 * k for k in range(len(track))]
    for au, ud in zip(track, uds):
        assert q.push(au, ud) == 0
    _check(q, track, uds)
    q.free()


def test_whole_track_in_one_push_and_in_arbitrary_pieces(track):
    q = Queue()
    assert q.push(b"".join(track)) == 0
    _check(q, track)
    q.free()
    rng = random.Random(5)
    allnals = [x for au in track for x in nals(au)]
    for trial in range(20):       # the same NAL units cut into random pushes (whole NAL units each: the framing is validated per push)
        q = Queue()
        i = 0
        while i < len(allnals):
            n = rng.randint(1, 4)
            assert q.push(b"".join(allnals[i:i + n])) == 0
            i += n
        _check(q, track)
        q.free()


def test_truncated_framing_is_refused_and_leaves_the_queue_alone(track):
    q = Queue()
    assert q.push(track[0]) == 0
    assert q.push(track[1][:-3]) == -2 and q.push(track[1][:3]) == -2 and q.push(b"\xff\xff\xff\xff\x00") == -2
    assert q.count() == 0 and q.get(-1)[0] == track[0]
    assert q.push(track[1]) == 0 and q.count() == 1
    q.free()


def test_repeated_and_changed_parameter_sets(track):
    """a sample that repeats the parameter sets does not grow the remembered set; a CHANGED parameter set (same id) goes behind the old one - the parser
    takes the last one it reads - and a repetition of the OLD one moves it to the end again"""
    ps = [x for x in nals(track[0]) if is_ps(x)]
    slices = [b"".join(x for x in nals(au) if not is_ps(x)) for au in track]
    q = Queue()
    q.push(track[0])
    q.push(b"".join(ps) + slices[1])                      # repeats all three
    assert q.get(-2)[0] == b"".join(ps)
    pps = bytearray(ps[2]); pps[-1] ^= 0x40               # a different PPS payload (never parsed here)
    pps = bytes(pps)
    q.push(pps + slices[2])
    assert q.get(-2)[0] == b"".join(ps) + pps
    assert q.get(1)[0] == b"".join(ps) + pps + slices[2]
    q.push(ps[2] + slices[3])                             # the old PPS again: it is the newest now
    assert q.get(-2)[0] == ps[0] + ps[1] + pps + ps[2]
    assert q.get(2)[0] == ps[0] + ps[1] + pps + ps[2] + slices[3]
    q.free()


def test_user_data_follows_the_samples_of_the_last_push_only(track):
    q = Queue()
    q.push(track[0], 1)
    q.push(track[1] + track[2], 2)                        # two samples in one push: both carry its user_data
    q.push(track[3], 3)
    for x in nals(track[4]):                              # one sample in pieces: the last piece's user_data is the sample's
        q.push(x, 4)
    assert [q.get(k)[1] for k in range(q.count())] == [2, 2, 3, 4] and q.get(-1)[1] == 1
    q.L.emu_sq_drop(q.q, 2)                               # a chain of two was decoded
    q.push(track[5], 5)
    assert [q.get(k)[1] for k in range(q.count())] == [3, 4, 5]
    q.free()


def test_random_bytes_never_break_the_queue():
    """fuzz: random framed garbage in random pushes (the framing check passes, the NAL headers are arbitrary): no crash, bounded growth"""
    rng = random.Random(11)
    q = Queue()
    total = 0
    for _ in range(400):
        parts = []
        for _ in range(rng.randint(1, 5)):
            n = rng.choice([0, 1, 2, 3, 5, 40])
            parts.append(n.to_bytes(4, "big") + bytes(rng.randrange(256) for _ in range(n)))
        data = b"".join(parts)
        total += len(data)
        assert q.push(data, rng.randrange(1 << 40)) == 0
    sizes = [len(q.get(k)[0]) for k in range(q.count())]
    assert sum(sizes) <= total * (2 + len(q.get(-2)[0])) and len(q.get(-1)[0]) <= total
    q.free()
