"""One rank of the SPMD grid path on the CPU (tests/test_product_on_emulator.py): the emulated library (HIPDEC_LIBRARY), one emulated device per rank
(HIPEMU_DEVICES, hipdec_init(rank)), tests/emu/libfake_rccl.so as the transport (HIPDEC_RCCL_LIBRARY).  usage: rccl_rank.py <rank> <world> <id file>"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
import libheif_amd
from libheif_amd import HipDecError
from libheif_amd._capi import check
lib = libheif_amd.load_library()
check(lib.hipdec_init(rank))                 # one process per (emulated) GPU
from oracle import pyoracle as orc
from libheif_amd.grid import GridDecoderRccl, GridLayout, RcclComm
import test_grid_sharding as T


def exchange(raw):      # rank 0's unique id to the others: what torch.distributed's broadcast does in bench.py
    if rank == 0:
        open(idfile + ".part", "wb").write(raw); os.rename(idfile + ".part", idfile)
        return raw
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "rank 0 never published the unique id"
        time.sleep(0.01)
    return open(idfile, "rb").read()


vui = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
L = GridLayout(2, 3, 128, 128, 380, 250)
streams = {t: orc.encode(orc.synth_image(128, 128, 8, 1, seed=90 + t), wpp=t % 2, **vui) for t in range(6)}
comm = RcclComm(rank, world, exchange)
g = GridDecoderRccl(streams, L, comm)
for it in range(2):      # twice over the same buffers
    g.decode(); g.wait()
    if rank == 0:
        got = g.planes()
        exp = T._expected_canvas(L, streams)
        for c in range(3):
            assert (got[c] == exp[c]).all(), "canvas component %d (decode %d)" % (c, it)
if rank == 0:
    rgb = g.to_rgb(10)
    assert (rgb == orc.color_420_to_rgb24(exp[0], exp[1], exp[2], (1, 13, 6, 1)).reshape(250, -1)).all()
g.free()
# a tile that rank 1 % world owns is damaged: EVERY rank's wait reports the failure (one status all-reduce per decode) - rank 0 must not hand out a canvas
# with an undecoded tile, and nobody may be left waiting in a collective
bad = bytearray(streams[1])
for k in range(200, 260):
    bad[k] ^= 0x55
damaged = dict(streams); damaged[1] = bytes(bad)
g = GridDecoderRccl(damaged, L, comm)
try:
    g.decode(); g.wait()
    raise SystemExit("rank %d: the damaged tile went unnoticed" % rank)
except HipDecError:
    pass
g.free()
# and the communicator is still usable
g = GridDecoderRccl(streams, L, comm)
g.decode(); g.wait()
if rank == 0:
    assert (g.planes()[0] == exp[0]).all()
g.free(); comm.free()
print("RANK %d OK" % rank)
