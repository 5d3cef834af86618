"""Grid ('grid' item) decode with the tiles sharded over the GPUs of one node.

Device-side form of libheif/image-items/grid.cc: ImageItem_Grid::decode_full_grid_image (:250-468)
fans the tiles out to worker threads and decode_and_paste_tile_image (:482-577) pastes each decoded
tile into the canvas with HeifPixelImage::copy_image_to (image/pixelimage.cc:1115-1172).  Here:

  * tile t = row * cols + col (the order of the 'dimg' references, grid.cc:193,319) is owned by rank
    t mod world_size; every rank decodes its tiles as ONE hipdec batch (all CABAC substreams and CTB
    rows of all its tiles in one set of launches);
  * the one exchange step of the path is a gather of the decoded tile planes to the root rank
    (RCCL over xGMI through torch.distributed; `gloo` on CPU in the tests) — 1.5 bytes per pixel for
    8-bit 4:2:0, each tile packed Y|Cb|Cr so one message per rank carries all its tiles;
  * the root pastes the tiles at (col * tile_w, row * tile_h), clipped to the output size, and runs the
    single fused colour conversion over the canvas (bilinear chroma taps cross tile borders, so the
    colour stage must see the whole canvas — SURVEY.md §8e).

There is no all-reduce anywhere; single stills do not shard (they run as replicas).
"""
import ctypes as C

import numpy as np

from ._capi import check, load_library, ImageInfo


class GridLayout:
    def __init__(self, rows, cols, tile_w, tile_h, out_w, out_h, bit_depth=8):
        self.rows, self.cols, self.tile_w, self.tile_h = int(rows), int(cols), int(tile_w), int(tile_h)
        self.out_w, self.out_h, self.bit_depth = int(out_w), int(out_h), int(bit_depth)
        if self.out_w > self.cols * self.tile_w or self.out_h > self.rows * self.tile_h:
            raise ValueError("grid output size exceeds the tiled area (libheif rejects such grids, grid.cc:270-282)")
        if (self.tile_w | self.tile_h) & 1:
            raise ValueError("4:2:0 tiles must have even dimensions")

    @property
    def n_tiles(self):
        return self.rows * self.cols

    @property
    def sample_bytes(self):
        return 2 if self.bit_depth > 8 else 1

    @property
    def tile_bytes(self):
        return (self.tile_w * self.tile_h * 3 // 2) * self.sample_bytes

    def origin(self, t):
        r, c = divmod(t, self.cols)
        return c * self.tile_w, r * self.tile_h


def owner(t, world):
    return t % world


def shard(n_tiles, rank, world):
    """tile indices owned by `rank`, in decode / message order"""
    return [t for t in range(n_tiles) if owner(t, world) == rank]


def slots_per_rank(n_tiles, world):
    return (n_tiles + world - 1) // world


def gather_tiles(local, layout, rank, world, group=None):
    """local: uint8 tensor [slots_per_rank, tile_bytes] holding this rank's packed tiles (unused slots
    are padding).  Returns, on rank 0, a tensor [world, slots, tile_bytes]; None elsewhere."""
    import torch
    if world == 1:
        return local.unsqueeze(0)
    import torch.distributed as dist
    if rank == 0:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(out.unbind(0)), dst=0, group=group)
        return out
    dist.gather(local, gather_list=None, dst=0, group=group)
    return None


def paste_plan(layout, world):
    """[(tile, rank, slot, x0, y0, w, h)] with (w, h) clipped to the output size — the arithmetic of
    copy_image_to (pixelimage.cc:1115-1172): chroma offsets / sizes are the luma ones halved."""
    plan = []
    for t in range(layout.n_tiles):
        x0, y0 = layout.origin(t)
        w = min(layout.tile_w, layout.out_w - x0)
        h = min(layout.tile_h, layout.out_h - y0)
        if w <= 0 or h <= 0:
            continue
        r = owner(t, world)
        plan.append((t, r, shard(layout.n_tiles, r, world).index(t), x0, y0, w, h))
    return plan


def alloc_canvas(layout, device):
    import torch
    es = layout.sample_bytes
    cw, ch = (layout.out_w + 1) // 2, (layout.out_h + 1) // 2
    return [torch.zeros((layout.out_h, layout.out_w * es), dtype=torch.uint8, device=device),
            torch.zeros((ch, cw * es), dtype=torch.uint8, device=device),
            torch.zeros((ch, cw * es), dtype=torch.uint8, device=device)]


def paste_tiles(gathered, layout, world, canvas):
    """root only: copies every tile's planes from the gathered buffer into the canvas planes"""
    es = layout.sample_bytes
    tw, th = layout.tile_w, layout.tile_h
    ysz, csz = tw * th * es, (tw // 2) * (th // 2) * es
    on_gpu = gathered.is_cuda
    lib = load_library() if on_gpu else None
    for (t, r, slot, x0, y0, w, h) in paste_plan(layout, world):
        tile = gathered[r, slot]
        for c, (off, pw, ph, px, py, src_w) in enumerate(((0, w, h, x0, y0, tw),
                                                          (ysz, (w + 1) // 2, (h + 1) // 2, x0 // 2, y0 // 2, tw // 2),
                                                          (ysz + csz, (w + 1) // 2, (h + 1) // 2, x0 // 2, y0 // 2, tw // 2))):
            dst = canvas[c]
            if on_gpu:
                check(lib.hipdec_copy2d_d2d(dst.data_ptr() + py * dst.stride(0) + px * es, dst.stride(0),
                                            tile.data_ptr() + off, src_w * es, pw * es, ph, None))
            else:
                src = tile[off:off + src_w * es * (th if c == 0 else th // 2)].view(-1, src_w * es)
                dst[py:py + ph, px * es:(px + pw) * es] = src[:ph, :pw * es]
    if on_gpu:
        check(lib.hipdec_stream_synchronize(None))


class GridDecoder:
    """Decodes one grid image with its tiles sharded over `world` ranks (one process per GPU)."""

    def __init__(self, tile_streams, layout, rank=0, world=1, group=None):
        """tile_streams: {tile index: plugin-framed HEVC stream} for at least the tiles this rank owns"""
        import torch
        from .decoder import Batch, _bind
        self.layout, self.rank, self.world, self.group = layout, rank, world, group
        self.mine = shard(layout.n_tiles, rank, world)
        self.lib = _bind(load_library())
        self.batch = Batch([tile_streams[t] for t in self.mine]) if self.mine else None
        if self.batch is not None:
            for i in range(len(self.mine)):
                d = self.batch.info(i)
                if (d["width"], d["height"]) != (layout.tile_w, layout.tile_h) or d["bit_depth_luma"] != layout.bit_depth:
                    raise ValueError("tile %d does not match the grid's tile size / bit depth" % self.mine[i])
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.send = torch.zeros((slots_per_rank(layout.n_tiles, world), layout.tile_bytes), dtype=torch.uint8, device=self.device)
        self.canvas = alloc_canvas(layout, self.device) if rank == 0 else None

    def decode(self):
        """asynchronous decode of the local tiles, gather, paste.  Returns the canvas planes (torch
        uint8 tensors on the root's GPU; rows are bytes) on rank 0, None elsewhere."""
        import torch
        if self.batch is not None:
            self.batch.run()
            for i in range(len(self.mine)):
                check(self.lib.hipdec_batch_pack_item(self.batch._h, i, self.send[i].data_ptr(), self.layout.tile_bytes, None))
            self.batch.status()   # synchronises the library stream; device-side decode errors are loud
        torch.cuda.current_stream().synchronize()
        gathered = gather_tiles(self.send, self.layout, self.rank, self.world, self.group)
        if self.rank != 0:
            return None
        torch.cuda.current_stream().synchronize()
        paste_tiles(gathered, self.layout, self.world, self.canvas)
        return self.canvas

    def to_rgb(self, nclx=(1, 13, 6, 1)):
        """root only: the fused colour stage over the canvas (8-bit: interleaved RGB24 rows)"""
        import ctypes as C
        import torch
        from ._capi import Nclx
        L = self.layout
        if L.bit_depth != 8:
            raise ValueError("to_rgb: 8-bit canvases only (use hipdec_color_420_to_rrggbb for HDR)")
        out = torch.empty((L.out_h, L.out_w * 3), dtype=torch.uint8, device=self.device)
        ns = Nclx(1, *[int(v) for v in nclx])
        y, cb, cr = self.canvas
        m = 6 if nclx[2] == 2 else nclx[2]
        if nclx[3] and m not in (0, 8):
            check(self.lib.hipdec_color_420_to_rgb24(y.data_ptr(), y.stride(0), cb.data_ptr(), cb.stride(0), cr.data_ptr(), cr.stride(0),
                                                     L.out_w, L.out_h, C.byref(ns), out.data_ptr(), out.stride(0), 0, None))
        else:
            check(self.lib.hipdec_color_ycbcr_to_rgb24_float(y.data_ptr(), y.stride(0), cb.data_ptr(), cb.stride(0), cr.data_ptr(), cr.stride(0),
                                                             L.out_w, L.out_h, 1, C.byref(ns), out.data_ptr(), out.stride(0), 0, None))
        check(self.lib.hipdec_stream_synchronize(None))
        return out


class GridDecoderC:
    """The same grid decode through the C ABI's hipdec_grid_* (include/heif_hipdec.h): ONE process, the tiles sharded over `devices`
    (device indices, entries may repeat), every decoded tile pasted into the canvas on devices[0] by a strided peer copy.  This is the
    product path; GridDecoder above is the multi-process (torch.distributed) test driver of the same partition."""

    def __init__(self, tile_streams, layout, devices=None, max_image_size_pixels=0):
        import ctypes as C
        from .decoder import _bind
        self._C = C
        self.layout = layout
        self.lib = lib = _bind(load_library())
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        lib.hipdec_grid_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, C.POINTER(C.c_char_p), C.POINTER(sz), C.POINTER(ci), ci, C.c_uint64]
        lib.hipdec_grid_free.argtypes = [vp]
        lib.hipdec_grid_decode.argtypes = [vp]
        lib.hipdec_grid_wait.argtypes = [vp]
        lib.hipdec_grid_read_plane.argtypes = [vp, ci, vp, sz]
        lib.hipdec_grid_to_rgb.argtypes = [vp, ci, ci, ci, vp, sz, ci]
        lib.hipdec_grid_canvas_plane.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(sz), C.POINTER(ci)]
        lib.hipdec_grid_info.argtypes = [vp, C.POINTER(ImageInfo), C.POINTER(ci)]
        n = layout.n_tiles
        self._keep = [bytes(tile_streams[t]) for t in range(n)]
        arr = (C.c_char_p * n)(*self._keep)
        sizes = (sz * n)(*[len(s) for s in self._keep])
        devs = None
        if devices is not None:
            devs = (ci * len(devices))(*[int(d) for d in devices])
        self._h = vp()
        check(lib.hipdec_grid_create(C.byref(self._h), layout.rows, layout.cols, layout.out_w, layout.out_h, arr, sizes, devs,
                                     len(devices) if devices is not None else 0, int(max_image_size_pixels)))

    def decode(self):
        check(self.lib.hipdec_grid_decode(self._h))

    def wait(self):
        check(self.lib.hipdec_grid_wait(self._h))

    def planes(self):
        L = self.layout
        dt = np.uint16 if L.bit_depth > 8 else np.uint8
        out = []
        info = ImageInfo()
        check(self.lib.hipdec_grid_info(self._h, C.byref(info), None))
        for c in range(3 if info.chroma_format_idc else 1):
            w, h = (L.out_w, L.out_h) if c == 0 else (info.chroma_width, info.chroma_height)     # (4:2:2 / 4:4:4 tiles: the canvas keeps their subsampling)
            a = np.empty((h, w), dt)
            check(self.lib.hipdec_grid_read_plane(self._h, c, a.ctypes.data, w * a.itemsize))
            out.append(a)
        return out

    def to_rgb(self, out_chroma=10, upsampling=1, only_preferred=False, out_dev=None):
        """host array (h, w * bytes per pixel) — or, with out_dev = (device pointer, stride), converts into that device buffer"""
        L = self.layout
        bpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_chroma]
        if out_dev is not None:
            check(self.lib.hipdec_grid_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), out_dev[0], out_dev[1], 1))
            return None
        a = np.empty((L.out_h, L.out_w * bpp), np.uint8)
        check(self.lib.hipdec_grid_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), a.ctypes.data, L.out_w * bpp, 0))
        return a

    def free(self):
        if self._h:
            self.lib.hipdec_grid_free(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _torch_rccl_path():
    """PyTorch-ROCm wheels bundle their own librccl next to their HIP runtime; when it is there the product's dlopen() takes that copy
    (one RCCL per process), otherwise the system's librccl.so.1"""
    import importlib.util
    import os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        p = os.path.join(loc, "lib", "librccl.so")
        if os.path.exists(p):
            return p
    return None


class RcclComm:
    """An RCCL communicator made by the library itself (hipdec_rccl_comm_create = ncclCommInitRank on the hipdec_init device).  `exchange_id`
    carries rank 0's 128-byte unique id to the other ranks: a callable bytes -> bytes (e.g. a torch.distributed broadcast); world 1 needs none."""

    def __init__(self, rank=0, world=1, exchange_id=None):
        import os
        p = _torch_rccl_path()
        if p and "HIPDEC_RCCL_LIBRARY" not in os.environ:
            os.environ["HIPDEC_RCCL_LIBRARY"] = p
        self.lib = lib = load_library()
        lib.hipdec_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
        lib.hipdec_rccl_comm_destroy.argtypes = [C.c_void_p]
        lib.hipdec_rccl_unique_id.argtypes = [C.c_void_p]
        self.rank, self.world = int(rank), int(world)
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            check(lib.hipdec_rccl_unique_id(buf))
        raw = bytes(buf.raw)
        if self.world > 1:
            if exchange_id is None:
                raise ValueError("RcclComm: world > 1 needs exchange_id to distribute rank 0's unique id")
            raw = exchange_id(raw)
        self._h = C.c_void_p()
        check(lib.hipdec_rccl_comm_create(C.byref(self._h), self.world, self.rank, C.c_char_p(raw)))

    def free(self):
        if self._h:
            self.lib.hipdec_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()


class GridDecoderRccl:
    """The SPMD product path: hipdec_grid_create_rccl / _decode / _to_rgb (include/heif_hipdec.h) - one process per GPU, this rank's tiles decoded on its
    device, the packed tiles gathered to rank 0 by one grouped ncclSend / ncclRecv inside libheifhip.so, pasted and colour-converted there."""

    def __init__(self, tile_streams, layout, comm, max_image_size_pixels=0):
        self.layout, self.comm = layout, comm
        self.lib = lib = load_library()
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        lib.hipdec_grid_create_rccl.argtypes = [C.POINTER(vp), vp, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_uint64]
        lib.hipdec_grid_rccl_free.argtypes = [vp]
        lib.hipdec_grid_rccl_decode.argtypes = [vp]
        lib.hipdec_grid_rccl_wait.argtypes = [vp]
        lib.hipdec_grid_rccl_read_plane.argtypes = [vp, ci, vp, sz]
        lib.hipdec_grid_rccl_to_rgb.argtypes = [vp, ci, ci, ci, vp, sz, ci]
        n = layout.n_tiles
        mine = set(shard(n, comm.rank, comm.world))
        self._keep = [bytes(tile_streams[t]) if t in mine else None for t in range(n)]
        arr = (C.c_char_p * n)(*self._keep)
        sizes = (sz * n)(*[len(s) if s is not None else 0 for s in self._keep])
        self._h = vp()
        check(lib.hipdec_grid_create_rccl(C.byref(self._h), comm._h, comm.rank, comm.world, layout.rows, layout.cols, layout.out_w, layout.out_h,
                                          arr, sizes, int(max_image_size_pixels)))

    def decode(self):
        rc = self.lib.hipdec_grid_rccl_decode(self._h)
        if rc != 0:
            # a rank whose shard could not be queued still joins the status exchange of wait() - its peers are inside that all-reduce;
            # wait() then returns this rank's own error again
            self.lib.hipdec_grid_rccl_wait(self._h)
        check(rc)

    def wait(self):
        check(self.lib.hipdec_grid_wait(self._h))

    def planes(self):
        L = self.layout
        dt = np.uint16 if L.bit_depth > 8 else np.uint8
        out = []
        info = ImageInfo()
        check(self.lib.hipdec_grid_info(self._h, C.byref(info), None))
        for c in range(3 if info.chroma_format_idc else 1):
            w, h = (L.out_w, L.out_h) if c == 0 else (info.chroma_width, info.chroma_height)     # (4:2:2 / 4:4:4 tiles: the canvas keeps their subsampling)
            a = np.empty((h, w), dt)
            check(self.lib.hipdec_grid_read_plane(self._h, c, a.ctypes.data, w * a.itemsize))
            out.append(a)
        return out

    def to_rgb(self, out_chroma=10, upsampling=1, only_preferred=False, out_dev=None):
        """host array (h, w * bytes per pixel) — or, with out_dev = (device pointer, stride), converts into that device buffer"""
        L = self.layout
        bpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_chroma]
        if out_dev is not None:
            check(self.lib.hipdec_grid_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), out_dev[0], out_dev[1], 1))
            return None
        a = np.empty((L.out_h, L.out_w * bpp), np.uint8)
        check(self.lib.hipdec_grid_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), a.ctypes.data, L.out_w * bpp, 0))
        return a

    def free(self):
        if self._h:
            self.lib.hipdec_grid_free(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _torch_rccl_path():
    """PyTorch-ROCm wheels bundle their own librccl next to their HIP runtime; when it is there the product's dlopen() takes that copy
    (one RCCL per process), otherwise the system's librccl.so.1"""
    import importlib.util
    import os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        p = os.path.join(loc, "lib", "librccl.so")
        if os.path.exists(p):
            return p
    return None


class RcclComm:
    """An RCCL communicator made by the library itself (hipdec_rccl_comm_create = ncclCommInitRank on the hipdec_init device).  `exchange_id`
    carries rank 0's 128-byte unique id to the other ranks: a callable bytes -> bytes (e.g. a torch.distributed broadcast); world 1 needs none."""

    def __init__(self, rank=0, world=1, exchange_id=None):
        import os
        p = _torch_rccl_path()
        if p and "HIPDEC_RCCL_LIBRARY" not in os.environ:
            os.environ["HIPDEC_RCCL_LIBRARY"] = p
        self.lib = lib = load_library()
        lib.hipdec_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
        lib.hipdec_rccl_comm_destroy.argtypes = [C.c_void_p]
        lib.hipdec_rccl_unique_id.argtypes = [C.c_void_p]
        self.rank, self.world = int(rank), int(world)
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            check(lib.hipdec_rccl_unique_id(buf))
        raw = bytes(buf.raw)
        if self.world > 1:
            if exchange_id is None:
                raise ValueError("RcclComm: world > 1 needs exchange_id to distribute rank 0's unique id")
            raw = exchange_id(raw)
        self._h = C.c_void_p()
        check(lib.hipdec_rccl_comm_create(C.byref(self._h), self.world, self.rank, C.c_char_p(raw)))

    def free(self):
        if self._h:
            self.lib.hipdec_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()


class GridDecoderRccl:
    """The SPMD product path: hipdec_grid_create_rccl / _decode / _to_rgb (include/heif_hipdec.h) - one process per GPU, this rank's tiles decoded on its
    device, the packed tiles gathered to rank 0 by one grouped ncclSend / ncclRecv inside libheifhip.so, pasted and colour-converted there."""

    def __init__(self, tile_streams, layout, comm, max_image_size_pixels=0):
        self.layout, self.comm = layout, comm
        self.lib = lib = load_library()
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        lib.hipdec_grid_create_rccl.argtypes = [C.POINTER(vp), vp, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_uint64]
        lib.hipdec_grid_rccl_free.argtypes = [vp]
        lib.hipdec_grid_rccl_decode.argtypes = [vp]
        lib.hipdec_grid_rccl_wait.argtypes = [vp]
        lib.hipdec_grid_rccl_read_plane.argtypes = [vp, ci, vp, sz]
        lib.hipdec_grid_rccl_to_rgb.argtypes = [vp, ci, ci, ci, vp, sz, ci]
        n = layout.n_tiles
        mine = set(shard(n, comm.rank, comm.world))
        self._keep = [bytes(tile_streams[t]) if t in mine else None for t in range(n)]
        arr = (C.c_char_p * n)(*self._keep)
        sizes = (sz * n)(*[len(s) if s is not None else 0 for s in self._keep])
        self._h = vp()
        check(lib.hipdec_grid_create_rccl(C.byref(self._h), comm._h, comm.rank, comm.world, layout.rows, layout.cols, layout.out_w, layout.out_h,
                                          arr, sizes, int(max_image_size_pixels)))

    def decode(self):
        rc = self.lib.hipdec_grid_rccl_decode(self._h)
        if rc != 0:
            # a rank whose shard could not be queued still joins the status exchange of wait(): its peers are inside that all-reduce
            msg = self.lib.hipdec_last_error()
            self.lib.hipdec_grid_rccl_wait(self._h)
            raise RuntimeError("hipdec_grid_rccl_decode failed (%d): %s" % (rc, msg.decode() if isinstance(msg, bytes) else msg))

    def wait(self):
        check(self.lib.hipdec_grid_rccl_wait(self._h))

    def planes(self):
        """rank 0: the canvas planes (4:2:0 layouts)"""
        L = self.layout
        dt = np.uint16 if L.bit_depth > 8 else np.uint8
        out = []
        for c in range(3):
            w, h = (L.out_w, L.out_h) if c == 0 else ((L.out_w + 1) // 2, (L.out_h + 1) // 2)
            a = np.empty((h, w), dt)
            check(self.lib.hipdec_grid_rccl_read_plane(self._h, c, a.ctypes.data, w * a.itemsize))
            out.append(a)
        return out

    def to_rgb(self, out_chroma=10, upsampling=1, only_preferred=False, out_dev=None):
        L = self.layout
        bpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_chroma]
        if out_dev is not None:
            check(self.lib.hipdec_grid_rccl_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), out_dev[0], out_dev[1], 1))
            return None
        a = np.empty((L.out_h, L.out_w * bpp), np.uint8)
        check(self.lib.hipdec_grid_rccl_to_rgb(self._h, out_chroma, upsampling, int(only_preferred), a.ctypes.data, L.out_w * bpp, 0))
        return a

    def free(self):
        if self._h:
            self.lib.hipdec_grid_rccl_free(self._h)
            self._h = C.c_void_p()
