"""Multi-GPU grid path (libheif_amd/grid.py): tile -> rank partition, the gather of decoded tile planes
to the root and the paste into the canvas.  CPU part: pure partition logic + a world_size-2 `gloo` run
in which each rank's tiles are produced by the oracle (standing in for the device decode) and go
through the SAME gather / paste code as on the GPUs.  GPU part: the real GridDecoder at world size 1."""
import os
import numpy as np
import pytest

from oracle import pyoracle as orc
from libheif_amd import grid as G


def _tiles(layout, seed0=40):
    return {t: orc.encode(orc.synth_image(layout.tile_w, layout.tile_h, layout.bit_depth, 1, seed=seed0 + t), wpp=t % 2)
            for t in range(layout.n_tiles)}


def _expected_canvas(layout, streams):
    es = layout.sample_bytes
    full = [np.zeros((layout.rows * layout.tile_h, layout.cols * layout.tile_w), np.uint16),
            np.zeros((layout.rows * layout.tile_h // 2, layout.cols * layout.tile_w // 2), np.uint16),
            np.zeros((layout.rows * layout.tile_h // 2, layout.cols * layout.tile_w // 2), np.uint16)]
    for t, s in streams.items():
        ref = orc.decode(s)
        x0, y0 = layout.origin(t)
        for c in range(3):
            sub = 1 if c == 0 else 2
            full[c][y0 // sub:(y0 + layout.tile_h) // sub, x0 // sub:(x0 + layout.tile_w) // sub] = ref["planes"][c]
    dt = np.uint16 if es == 2 else np.uint8
    cw, ch = (layout.out_w + 1) // 2, (layout.out_h + 1) // 2
    return [full[0][:layout.out_h, :layout.out_w].astype(dt), full[1][:ch, :cw].astype(dt), full[2][:ch, :cw].astype(dt)]


def _pack(ref_planes, dt):
    return np.concatenate([np.ascontiguousarray(p, dtype=dt).reshape(-1).view(np.uint8) for p in ref_planes])


def test_partition_and_paste_plan():
    L = G.GridLayout(6, 8, 1024, 1024, 8192, 6144)
    assert L.n_tiles == 48 and L.tile_bytes == 1024 * 1024 * 3 // 2
    for world in (1, 2, 4, 8, 5):
        owned = [G.shard(48, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(48))
        assert max(len(o) for o in owned) == G.slots_per_rank(48, world)
        plan = G.paste_plan(L, world)
        assert len(plan) == 48
        for t, r, slot, x0, y0, w, h in plan:
            assert owned[r][slot] == t and (x0, y0) == ((t % 8) * 1024, (t // 8) * 1024) and (w, h) == (1024, 1024)
    # 8 GPUs: every rank owns one tile column
    assert all(set(t % 8 for t in G.shard(48, r, 8)) == {r} for r in range(8))
    clipped = G.paste_plan(G.GridLayout(2, 2, 128, 128, 200, 130), 2)
    assert [(p[5], p[6]) for p in clipped] == [(128, 128), (72, 128), (128, 2), (72, 2)]
    with pytest.raises(ValueError):
        G.GridLayout(2, 2, 128, 128, 300, 100)


def _gloo_worker(rank, world, port, bit_depth, result_file):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = G.GridLayout(2, 3, 64, 64, 180, 120, bit_depth)
        streams = _tiles(L)
        dt = np.uint16 if bit_depth > 8 else np.uint8
        send = torch.zeros((G.slots_per_rank(L.n_tiles, world), L.tile_bytes), dtype=torch.uint8)
        for slot, t in enumerate(G.shard(L.n_tiles, rank, world)):      # this rank "decodes" only its own tiles
            send[slot] = torch.from_numpy(_pack(orc.decode(streams[t])["planes"], dt))
        gathered = G.gather_tiles(send, L, rank, world)
        if rank == 0:
            canvas = G.alloc_canvas(L, "cpu")
            G.paste_tiles(gathered, L, world, canvas)
            exp = _expected_canvas(L, streams)
            ok = all(np.array_equal(canvas[c].numpy().view(dt), exp[c]) for c in range(3))
            open(result_file, "w").write("ok" if ok else "mismatch")
        else:
            assert gathered is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bit_depth", [(2, 8), (3, 10)])
def test_gather_and_paste_over_gloo(tmp_path, world, bit_depth):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    result = str(tmp_path / "result.txt")
    mp.spawn(_gloo_worker, args=(world, port, bit_depth, result), nprocs=world, join=True)
    assert open(result).read() == "ok"


@pytest.mark.gpu
def test_grid_decoder_single_gpu_matches_oracle():
    import torch
    L = G.GridLayout(2, 3, 128, 128, 380, 250)
    streams = _tiles(L)
    gd = G.GridDecoder(streams, L, rank=0, world=1)
    canvas = gd.decode()
    exp = _expected_canvas(L, streams)
    for c in range(3):
        np.testing.assert_array_equal(canvas[c].cpu().numpy(), exp[c], err_msg="canvas component %d" % c)
    rgb = gd.to_rgb((1, 13, 6, 1)).cpu().numpy()
    np.testing.assert_array_equal(rgb, orc.color_420_to_rgb24(exp[0], exp[1], exp[2], (1, 13, 6, 1)).reshape(L.out_h, -1))


# ---- the product path: hipdec_grid_* in C++, one process over several devices (include/heif_hipdec.h) ---------------------------------
def test_grid_c_api_without_gpu_fails_loudly():
    import libheif_amd
    from libheif_amd import HipDecError
    from libheif_amd.grid import GridDecoderC, GridLayout
    if libheif_amd.load_library().hipdec_device_count() > 0:
        pytest.skip("a GPU is present")
    s = orc.encode(orc.synth_image(64, 64, 8, 1, seed=1))
    with pytest.raises(HipDecError) as e:
        GridDecoderC({0: s, 1: s}, GridLayout(1, 2, 64, 64, 128, 64))
    assert e.value.code == -6


def _c_grid_case(rows, cols, tw, th, ow, oh, devices, seed0=70, **cfg):
    from libheif_amd.grid import GridDecoderC, GridLayout
    streams = [orc.encode(orc.synth_image(tw, th, cfg.get("bit_depth", 8), 1, seed=seed0 + t), **cfg) for t in range(rows * cols)]
    es = np.uint16 if cfg.get("bit_depth", 8) > 8 else np.uint8
    canvas = [np.zeros((rows * th, cols * tw), es), np.zeros((rows * th // 2, cols * tw // 2), es), np.zeros((rows * th // 2, cols * tw // 2), es)]
    for t, s in enumerate(streams):
        ref = orc.decode(s)
        r, c = divmod(t, cols)
        for k in range(3):
            d = 1 if k == 0 else 2
            canvas[k][r * th // d:(r + 1) * th // d, c * tw // d:(c + 1) * tw // d] = ref["planes"][k]
    g = GridDecoderC({t: s for t, s in enumerate(streams)}, GridLayout(rows, cols, tw, th, ow, oh, cfg.get("bit_depth", 8)), devices)
    g.decode(); g.wait()
    got = g.planes()
    np.testing.assert_array_equal(got[0], canvas[0][:oh, :ow])
    np.testing.assert_array_equal(got[1], canvas[1][:(oh + 1) // 2, :(ow + 1) // 2])
    np.testing.assert_array_equal(got[2], canvas[2][:(oh + 1) // 2, :(ow + 1) // 2])
    return g, [c for c in canvas]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0], [0, 0], [0, 0, 0, 0, 0]], ids=["all_visible", "one", "two_shards", "five_shards"])
def test_grid_c_api_shards_paste_into_the_canvas(devices):
    """tiles t mod G per shard (several shards on the one device of the GPU box exercise the partition, the per-shard batches and
    the strided pastes), clipped output, fused colour stage over the canvas"""
    vui = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    g, canvas = _c_grid_case(2, 3, 128, 128, 380, 250, devices, **vui)
    rgb = g.to_rgb(10)
    want = orc.color_420_to_rgb24(canvas[0][:250, :380], canvas[1][:125, :190], canvas[2][:125, :190], (1, 13, 6, 1)).reshape(250, -1)
    np.testing.assert_array_equal(rgb, want)
    g.decode(); g.wait()          # a second decode over the same arenas
    np.testing.assert_array_equal(g.planes()[0], canvas[0][:250, :380])
    g.free()


@pytest.mark.gpu
def test_grid_c_api_main10_and_errors():
    from libheif_amd import HipDecError
    from libheif_amd.grid import GridDecoderC, GridLayout
    g, canvas = _c_grid_case(2, 2, 136, 72, 272, 144, [0, 0], bit_depth=10, vui_matrix=9, vui_primaries=9, vui_transfer=16)
    rgb = g.to_rgb(14)
    np.testing.assert_array_equal(rgb, orc.color_420_to_rrggbb(canvas[0], canvas[1], canvas[2], 10, (9, 16, 9, 0), little_endian=True))
    g.free()
    a = orc.encode(orc.synth_image(64, 64, 8, 1, seed=1)); b = orc.encode(orc.synth_image(128, 64, 8, 1, seed=2))
    with pytest.raises(HipDecError):           # tiles of different size (grid.cc rejects such grids)
        GridDecoderC({0: a, 1: b}, GridLayout(1, 2, 64, 64, 128, 64))
    with pytest.raises(HipDecError):           # a device that does not exist
        GridDecoderC({0: a, 1: a}, GridLayout(1, 2, 64, 64, 128, 64), [0, 99])


# ---- the SPMD product path: hipdec_grid_*_rccl (one process per GPU, RCCL gather inside libheifhip.so) --------------------------------
def test_grid_rccl_entry_points_without_gpu_fail_loudly():
    import ctypes as C
    import libheif_amd
    lib = libheif_amd.load_library()
    if lib.hipdec_device_count() > 0:
        pytest.skip("a GPU is present")
    assert lib.hipdec_rccl_available() in (0, 1)
    h = C.c_void_p()
    lib.hipdec_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
    buf = C.create_string_buffer(128)
    assert lib.hipdec_rccl_comm_create(C.byref(h), 1, 0, buf) != 0 and not h.value            # no device (or no RCCL): an error, never a fake communicator
    assert lib.hipdec_rccl_comm_create(C.byref(h), 2, 2, buf) == -1                              # rank out of range
    lib.hipdec_grid_create_rccl.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    assert lib.hipdec_grid_create_rccl(C.byref(h), None, 0, 1, 1, 2, 128, 64, None, None, 0) == -1   # NULL communicator / tiles
    lib.hipdec_grid_rccl_decode.argtypes = [C.c_void_p]
    assert lib.hipdec_grid_rccl_decode(None) == -1


@pytest.mark.gpu
def test_grid_rccl_world1_matches_oracle_and_the_one_process_form():
    """the GPU box has one GPU (RCCL refuses two ranks on one device), so this runs the whole SPMD path - communicator from
    hipdec_rccl_comm_create, the geometry all-reduce, decode, paste, colour stage - at world size 1; the N-rank sends / receives first run in
    `bench.py --gpus N` (grid_sharded.rccl), which compares their canvas with this same one-GPU result"""
    import libheif_amd
    from libheif_amd.grid import GridDecoderC, GridDecoderRccl, GridLayout, RcclComm
    if not libheif_amd.load_library().hipdec_rccl_available():
        pytest.skip("librccl is not loadable here")
    vui = dict(vui_primaries=1, vui_transfer=13, vui_matrix=6, vui_full_range=1)
    L = GridLayout(2, 3, 128, 128, 380, 250)
    streams = {t: orc.encode(orc.synth_image(128, 128, 8, 1, seed=90 + t), wpp=t % 2, **vui) for t in range(6)}
    comm = RcclComm(0, 1)
    g = GridDecoderRccl(streams, L, comm)
    g.decode(); g.wait()
    got = g.planes()
    exp = _expected_canvas(L, streams)
    for c in range(3):
        np.testing.assert_array_equal(got[c], exp[c], err_msg="canvas component %d" % c)
    rgb = g.to_rgb(10)
    ref = GridDecoderC(streams, L, [0])
    ref.decode(); ref.wait()
    np.testing.assert_array_equal(rgb, ref.to_rgb(10))
    g.decode(); g.wait()      # again over the same buffers
    np.testing.assert_array_equal(g.planes()[0], exp[0])
    g.free(); ref.free(); comm.free()
