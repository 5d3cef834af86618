"""Test tooling: a minimal ISO-BMFF/HEIF reader that extracts, for every `hvc1` item, the exact byte
string libheif hands to a decoder plugin through push_data2() — the hvcC parameter-set NALs followed
by the item's iloc payload, each NAL prefixed with a 4-byte big-endian length
(reference: libheif/codecs/decoder.cc:275-308, libheif/codecs/hevc_boxes.cc:288-309).

Container parsing is NOT part of the product (the product sits behind libheif, which does this
itself); this module exists so the tests can feed the reference's own .heic fixtures to the oracle
and to the HIP decoder without libheif in the loop.
"""
import struct


def _boxes(buf, start, end):
    pos = start
    while pos + 8 <= end:
        size, typ = struct.unpack(">I4s", buf[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr or pos + size > end:
            break
        yield typ.decode("latin1"), pos + hdr, pos + size
        pos += size


def _find(buf, start, end, typ):
    for t, s, e in _boxes(buf, start, end):
        if t == typ:
            return s, e
    return None


class HeicFile:
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray)):
            self.buf = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, "rb") as f:
                self.buf = f.read()
        buf = self.buf
        meta = _find(buf, 0, len(buf), "meta")
        if meta is None:
            raise ValueError("no meta box")
        ms, me = meta[0] + 4, meta[1]  # FullBox
        self.items = {}       # id -> type
        self.iloc = {}        # id -> (construction_method, [(offset, length)])
        self.props = []       # list of (type, payload bytes)
        self.assoc = {}       # id -> [property index (1-based)]
        self.refs = {}        # (type, from_id) -> [to_ids]
        self.primary = None
        self.idat = b""
        for t, s, e in _boxes(buf, ms, me):
            if t == "pitm":
                ver = buf[s]
                self.primary = struct.unpack(">H" if ver == 0 else ">I", buf[s + 4:s + (6 if ver == 0 else 8)])[0]
            elif t == "iinf":
                ver = buf[s]
                p = s + 4 + (2 if ver == 0 else 4)
                for t2, s2, e2 in _boxes(buf, p, e):
                    if t2 != "infe":
                        continue
                    v = buf[s2]
                    if v == 2:
                        iid = struct.unpack(">H", buf[s2 + 4:s2 + 6])[0]
                        ityp = buf[s2 + 8:s2 + 12].decode("latin1")
                    elif v == 3:
                        iid = struct.unpack(">I", buf[s2 + 4:s2 + 8])[0]
                        ityp = buf[s2 + 10:s2 + 14].decode("latin1")
                    else:
                        continue
                    self.items[iid] = ityp
            elif t == "iloc":
                self._parse_iloc(s, e)
            elif t == "iprp":
                ipco = _find(buf, s, e, "ipco")
                if ipco:
                    for t2, s2, e2 in _boxes(buf, ipco[0], ipco[1]):
                        self.props.append((t2, buf[s2:e2]))
                for t2, s2, e2 in _boxes(buf, s, e):
                    if t2 == "ipma":
                        self._parse_ipma(s2, e2)
            elif t == "iref":
                ver = buf[s]
                for t2, s2, e2 in _boxes(buf, s + 4, e):
                    fmt, sz = (">H", 2) if ver == 0 else (">I", 4)
                    frm = struct.unpack(fmt, buf[s2:s2 + sz])[0]
                    cnt = struct.unpack(">H", buf[s2 + sz:s2 + sz + 2])[0]
                    to = [struct.unpack(fmt, buf[s2 + sz + 2 + i * sz:s2 + sz + 2 + (i + 1) * sz])[0] for i in range(cnt)]
                    self.refs.setdefault((t2, frm), []).extend(to)
            elif t == "idat":
                self.idat = buf[s:e]

    def _parse_iloc(self, s, e):
        buf = self.buf
        ver = buf[s]
        a, b = buf[s + 4], buf[s + 5]
        off_sz, len_sz, base_sz, idx_sz = a >> 4, a & 15, b >> 4, (b & 15) if ver in (1, 2) else 0
        p = s + 6
        if ver < 2:
            n = struct.unpack(">H", buf[p:p + 2])[0]; p += 2
        else:
            n = struct.unpack(">I", buf[p:p + 4])[0]; p += 4

        def rd(sz):
            nonlocal p
            v = int.from_bytes(buf[p:p + sz], "big") if sz else 0
            p += sz
            return v
        for _ in range(n):
            iid = rd(2 if ver < 2 else 4)
            cm = 0
            if ver in (1, 2):
                cm = rd(2) & 15
            rd(2)  # data_reference_index
            base = rd(base_sz)
            ext = []
            for _ in range(rd(2)):
                if ver in (1, 2) and idx_sz:
                    rd(idx_sz)
                o = rd(off_sz); l = rd(len_sz)
                ext.append((base + o, l))
            self.iloc[iid] = (cm, ext)

    def _parse_ipma(self, s, e):
        buf = self.buf
        ver, flags = buf[s], int.from_bytes(buf[s + 1:s + 4], "big")
        p = s + 4
        n = struct.unpack(">I", buf[p:p + 4])[0]; p += 4
        for _ in range(n):
            if ver < 1:
                iid = struct.unpack(">H", buf[p:p + 2])[0]; p += 2
            else:
                iid = struct.unpack(">I", buf[p:p + 4])[0]; p += 4
            cnt = buf[p]; p += 1
            lst = []
            for _ in range(cnt):
                if flags & 1:
                    v = struct.unpack(">H", buf[p:p + 2])[0] & 0x7FFF; p += 2
                else:
                    v = buf[p] & 0x7F; p += 1
                lst.append(v)
            self.assoc[iid] = lst

    def item_data(self, iid):
        cm, ext = self.iloc[iid]
        src = self.idat if cm == 1 else self.buf
        return b"".join(src[o:o + (l if l else len(src) - o)] for o, l in ext)

    def item_property(self, iid, typ):
        for idx in self.assoc.get(iid, []):
            if 1 <= idx <= len(self.props) and self.props[idx - 1][0] == typ:
                return self.props[idx - 1][1]
        return None

    def hevc_items(self):
        return [i for i, t in sorted(self.items.items()) if t == "hvc1"]

    def ispe(self, iid):
        p = self.item_property(iid, "ispe")
        if p is None:
            return None
        return struct.unpack(">II", p[4:12])

    def plugin_stream(self, iid):
        """bytes exactly as libheif pushes them into heif_decoder_plugin.push_data2()."""
        hvcc = self.item_property(iid, "hvcC")
        if hvcc is None:
            raise ValueError("item %d has no hvcC" % iid)
        length_size = (hvcc[21] & 3) + 1
        out = bytearray()
        p = 23
        for _ in range(hvcc[22]):
            p += 1  # array_completeness / NAL type
            n = struct.unpack(">H", hvcc[p:p + 2])[0]; p += 2
            for _ in range(n):
                l = struct.unpack(">H", hvcc[p:p + 2])[0]; p += 2
                out += struct.pack(">I", l) + hvcc[p:p + l]
                p += l
        data = self.item_data(iid)
        if length_size == 4:
            out += data
        else:
            q = 0
            while q + length_size <= len(data):
                l = int.from_bytes(data[q:q + length_size], "big"); q += length_size
                out += struct.pack(">I", l) + data[q:q + l]
                q += l
        return bytes(out)

    def grid(self, iid):
        """(rows, cols, out_w, out_h, [tile item ids]) for a 'grid' item (libheif grid.cc:34-73)."""
        d = self.item_data(iid)
        flags = d[1]
        rows, cols = d[2] + 1, d[3] + 1
        if flags & 1:
            w, h = struct.unpack(">II", d[4:12])
        else:
            w, h = struct.unpack(">HH", d[4:8])
        return rows, cols, w, h, self.refs.get(("dimg", iid), [])


# ------------------------------------------------------------------------------------------------
# minimal HEIC writer (test inputs only): wraps plugin-framed HEVC streams into single-item or grid
# files the REAL libheif can open, so that the drop-in tests drive heif_decode_image() end to end
# without reading /root/reference (ISO/IEC 23008-12; box layouts as libheif/box.cc parses them).
# ------------------------------------------------------------------------------------------------
def _box(typ, payload):
    return struct.pack(">I4s", 8 + len(payload), typ.encode("latin1")) + payload


def _fullbox(typ, version, flags, payload):
    return _box(typ, struct.pack(">I", (version << 24) | flags) + payload)


def split_nals(stream):
    """[4-byte BE length][NAL]... -> list of NAL byte strings"""
    out, p = [], 0
    while p + 4 <= len(stream):
        n = struct.unpack(">I", stream[p:p + 4])[0]
        out.append(stream[p + 4:p + 4 + n])
        p += 4 + n
    return out


def _unescape(nal):
    out, zeros = bytearray(), 0
    for b in nal:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def _hvcc(nals, chroma_format_idc=1, bit_depth=8):
    sps = [n for n in nals if (n[0] >> 1) & 63 == 33][0]
    r = _unescape(sps[2:])
    ptl = r[1:13]                      # general profile_tier_level: 12 bytes after the 1-byte sps header fields
    body = bytes([1]) + ptl[0:1] + ptl[1:5] + ptl[5:11] + ptl[11:12]
    body += struct.pack(">H", 0xF000) + bytes([0xFC, 0xFC | chroma_format_idc, 0xF8 | (bit_depth - 8), 0xF8 | (bit_depth - 8)])
    body += struct.pack(">H", 0) + bytes([0x0F])   # avgFrameRate, constantFrameRate=0 numTemporalLayers=1 nested=1 lengthSizeMinusOne=3
    arrays = b""
    count = 0
    for t in (32, 33, 34):
        sel = [n for n in nals if (n[0] >> 1) & 63 == t]
        if not sel:
            continue
        count += 1
        arrays += bytes([0x80 | t]) + struct.pack(">H", len(sel))
        for n in sel:
            arrays += struct.pack(">H", len(n)) + n
    return _box("hvcC", body + bytes([count]) + arrays)


def _payload(nals):
    return b"".join(struct.pack(">I", len(n)) + n for n in nals if (n[0] >> 1) & 63 < 32)


def build_heic(items, grid=None, bit_depth=8, chroma_format_idc=1, alpha_of=None, transforms=None):
    """items: list of (stream, width, height[, chroma_format_idc]) coded items (ids 1..n).  grid: None, or
    (rows, cols, out_w, out_h) -> an extra 'grid' item (id n+1, primary) referencing all items in order.
    alpha_of: {alpha item id: master item id} -> the alpha item becomes a hidden auxiliary image of its master ('auxC' property
    urn:mpeg:hevc:2015:auxid:1 + 'auxl' reference), what libheif attaches as the alpha channel (image_item.cc:949-1081).
    transforms: transformative properties of the primary item, applied in order (essential): ("irot", quarter turns ccw 0..3),
    ("imir", axis byte 0 / 1), ("clap", (width, height, left, top)) -> the 'clap' whose rounded edges are exactly that window."""
    alpha_of = alpha_of or {}
    cfs = [it[3] if len(it) > 3 else chroma_format_idc for it in items]
    items = [it[:3] for it in items]
    n = len(items)
    payloads = [_payload(split_nals(s)) for s, _, _ in items]
    grid_data = b""
    if grid is not None:
        rows, cols, ow, oh = grid
        grid_data = bytes([0, 0, rows - 1, cols - 1]) + struct.pack(">HH", ow, oh)
    primary = n + 1 if grid is not None else 1
    # ---- properties: per item hvcC + ispe (+ ispe for the grid) ----
    props, assoc = [], {}
    for i, (s, w, h) in enumerate(items):
        props.append(_hvcc(split_nals(s), cfs[i], bit_depth))
        props.append(_fullbox("ispe", 0, 0, struct.pack(">II", w, h)))
        assoc[i + 1] = [(len(props) - 1, True), (len(props), False)]
        if i + 1 in alpha_of:
            props.append(_fullbox("auxC", 0, 0, b"urn:mpeg:hevc:2015:auxid:1\0"))
            assoc[i + 1].append((len(props), True))
    if grid is not None:
        props.append(_fullbox("ispe", 0, 0, struct.pack(">II", grid[2], grid[3])))
        assoc[n + 1] = [(len(props), False)]
    for kind, arg in (transforms or []):
        pid = n + 1 if grid is not None else 1
        pw, ph = (grid[2], grid[3]) if grid is not None else (items[0][1], items[0][2])
        if kind == "irot":
            props.append(_box("irot", bytes([arg & 3])))
        elif kind == "imir":
            props.append(_box("imir", bytes([arg & 1])))
        elif kind == "clap":   # left = horizOff + (W - 1) / 2 - (cw - 1) / 2 (box.cc Box_clap::left_rounded): offsets in halves
            cw, ch, left, top = arg
            props.append(_box("clap", struct.pack(">IIIIiIiI", cw, 1, ch, 1, 2 * left + (cw - 1) - (pw - 1), 2, 2 * top + (ch - 1) - (ph - 1), 2)))
        else:
            raise ValueError(kind)
        assoc[pid].append((len(props), True))
    ipco = _box("ipco", b"".join(props))
    ipma = struct.pack(">I", len(assoc))
    for iid in sorted(assoc):
        ipma += struct.pack(">HB", iid, len(assoc[iid]))
        for idx, essential in assoc[iid]:
            ipma += bytes([(0x80 if essential else 0) | idx])
    iprp = _box("iprp", ipco + _fullbox("ipma", 0, 0, ipma))
    infes = b""
    for i in range(n):
        hidden = 1 if (grid is not None or (i + 1) in alpha_of) else 0
        infes += _fullbox("infe", 2, hidden, struct.pack(">HH4s", i + 1, 0, b"hvc1") + b"\0")
    if grid is not None:
        infes += _fullbox("infe", 2, 0, struct.pack(">HH4s", n + 1, 0, b"grid") + b"\0")
    n_items = n + (1 if grid is not None else 0)
    iinf = _fullbox("iinf", 0, 0, struct.pack(">H", n_items) + infes)
    pitm = _fullbox("pitm", 0, 0, struct.pack(">H", primary))
    hdlr = _fullbox("hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 12 + b"\0")
    iref = b""
    refs = b""
    if grid is not None:
        tiles = [i + 1 for i in range(n) if (i + 1) not in alpha_of]
        refs += _box("dimg", struct.pack(">HH", n + 1, len(tiles)) + b"".join(struct.pack(">H", t) for t in tiles))
    for a, m in sorted(alpha_of.items()):
        refs += _box("auxl", struct.pack(">HHH", a, 1, m))
    if refs:
        iref = _fullbox("iref", 0, 0, refs)
    ftyp = _box("ftyp", b"heic" + struct.pack(">I", 0) + b"mif1heic")

    def make_meta(offsets):
        iloc = struct.pack(">BBH", 0x44, 0x00, n_items)   # offset_size 4, length_size 4, base_offset_size 0
        blobs = payloads + ([grid_data] if grid is not None else [])
        for i, blob in enumerate(blobs):
            iloc += struct.pack(">HHH", i + 1, 0, 1) + struct.pack(">II", offsets[i], len(blob))
        return _fullbox("meta", 0, 0, hdlr + pitm + _fullbox("iloc", 0, 0, iloc) + iinf + iref + iprp)

    blobs = payloads + ([grid_data] if grid is not None else [])
    meta_len = len(make_meta([0] * len(blobs)))
    pos = len(ftyp) + meta_len + 8
    offsets = []
    for blob in blobs:
        offsets.append(pos)
        pos += len(blob)
    return ftyp + make_meta(offsets) + _box("mdat", b"".join(blobs))


def build_tili(tiles, rows, cols, tile_w, tile_h, out_w, out_h, bit_depth=8, chroma_format_idc=1):
    """A 'tili' tiled-image item (libheif/image-items/tiled.cc): ONE item whose data is an offset table ([u32 offset][u32 size] per tile,
    row-major, offsets relative to the item data) followed by the tiles' coded slice data; the tiles share one decoder configuration
    (the 'hvcC' child of the item's 'tilC' property, version 0 layout: tiled.cc:244-340), so every tile stream must carry the same
    parameter sets.  tiles: plugin-framed HEVC streams, row-major."""
    assert len(tiles) == rows * cols
    nal_sets = [split_nals(s) for s in tiles]
    ps0 = [n for n in nal_sets[0] if (n[0] >> 1) & 63 >= 32]
    for ns in nal_sets[1:]:
        assert [n for n in ns if (n[0] >> 1) & 63 >= 32] == ps0, "tili tiles must share their parameter sets"
    payloads = [_payload(ns) for ns in nal_sets]
    table_len = 8 * len(tiles)
    table, pos = b"", table_len
    for p in payloads:
        table += struct.pack(">II", pos, len(p))
        pos += len(p)
    item_data = table + b"".join(payloads)
    hvcc = _hvcc(nal_sets[0], chroma_format_idc, bit_depth)
    tile_ispe = _fullbox("ispe", 0, 0, struct.pack(">II", tile_w, tile_h))
    # tilC: flags bits 0-1 offset field length (0 = 32 bit), bits 2-3 size field length (2 = 32 bit)
    tilc = _fullbox("tilC", 0, 0x08, struct.pack(">II4sB", tile_w, tile_h, b"hvc1", 0) + bytes([2]) + hvcc + tile_ispe)
    ispe = _fullbox("ispe", 0, 0, struct.pack(">II", out_w, out_h))
    ipco = _box("ipco", tilc + ispe)
    ipma = struct.pack(">I", 1) + struct.pack(">HB", 1, 2) + bytes([0x80 | 1, 2])
    iprp = _box("iprp", ipco + _fullbox("ipma", 0, 0, ipma))
    iinf = _fullbox("iinf", 0, 0, struct.pack(">H", 1) + _fullbox("infe", 2, 0, struct.pack(">HH4s", 1, 0, b"tili") + b"\0"))
    pitm = _fullbox("pitm", 0, 0, struct.pack(">H", 1))
    hdlr = _fullbox("hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 12 + b"\0")
    ftyp = _box("ftyp", b"heic" + struct.pack(">I", 0) + b"mif1heic")

    def make_meta(offset):
        iloc = struct.pack(">BBH", 0x44, 0x00, 1) + struct.pack(">HHH", 1, 0, 1) + struct.pack(">II", offset, len(item_data))
        return _fullbox("meta", 0, 0, hdlr + pitm + _fullbox("iloc", 0, 0, iloc) + iinf + iprp)

    off = len(ftyp) + len(make_meta(0)) + 8
    return ftyp + make_meta(off) + _box("mdat", item_data)


def build_sequence(samples, width, height, bit_depth=8, chroma_format_idc=1, timescale=30, composition_offsets=None, chunk_size=0):
    """An image-sequence file the REAL libheif opens as a visual track (ISO/IEC 14496-12 movie structure, brand 'msf1'; boxes as
    libheif/sequences/seq_boxes.cc parses them, requirements of Track::load, sequences/track.cc:208-460): ftyp + moov(mvhd, trak(tkhd, mdia(mdhd, hdlr
    'pict', minf(vmhd, dinf, stbl(stsd(hvc1 + hvcC), stts, [ctts], stsc, stsz, stco))))) + mdat.
    samples: access units in DECODING order, plugin framing; the parameter sets of the first one go into the sample entry's hvcC (libheif pushes them
    with a chunk's first sample, codecs/decoder.cc:422) and every sample is stored with its slice NAL units only.
    composition_offsets: per sample (B tracks: decoding order != output order) -> a version-0 'ctts'; libheif pushes samples in file order and
    takes the output order from the decoder.  chunk_size: samples per chunk (0: one chunk)."""
    n = len(samples)
    nals0 = split_nals(samples[0])
    payloads = [_payload(split_nals(s)) for s in samples]
    hvcc = _hvcc(nals0, chroma_format_idc, bit_depth)
    entry = (b"\0" * 6 + struct.pack(">H", 1) + struct.pack(">HH", 0, 0) + b"\0" * 12 + struct.pack(">HH", width, height) +
             struct.pack(">II", 0x00480000, 0x00480000) + struct.pack(">I", 0) + struct.pack(">H", 1) + b"\0" * 32 + struct.pack(">Hh", 24, -1))
    stsd = _fullbox("stsd", 0, 0, struct.pack(">I", 1) + _box("hvc1", entry + hvcc))
    stts = _fullbox("stts", 0, 0, struct.pack(">I", 1) + struct.pack(">II", n, 1))
    ctts = b""
    if composition_offsets is not None:
        ctts = _fullbox("ctts", 0, 0, struct.pack(">I", n) + b"".join(struct.pack(">II", 1, int(o)) for o in composition_offsets))
    per = chunk_size if chunk_size and chunk_size > 0 else n
    chunks = [list(range(a, min(a + per, n))) for a in range(0, n, per)]
    stsc_entries = []
    for ci, ch in enumerate(chunks):
        if not stsc_entries or stsc_entries[-1][1] != len(ch):
            stsc_entries.append((ci + 1, len(ch), 1))
    stsc = _fullbox("stsc", 0, 0, struct.pack(">I", len(stsc_entries)) + b"".join(struct.pack(">III", *e) for e in stsc_entries))
    stsz = _fullbox("stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(p)) for p in payloads))
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _fullbox("mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, n) + struct.pack(">IH", 0x10000, 0x100) + b"\0" * 10 + matrix + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _fullbox("tkhd", 0, 3, struct.pack(">IIII", 0, 0, 1, 0) + struct.pack(">I", n) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) + matrix +
                    struct.pack(">II", width << 16, height << 16))
    mdhd = _fullbox("mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, n) + struct.pack(">HH", 0x55C4, 0))
    hdlr = _fullbox("hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 12 + b"\0")
    vmhd = _fullbox("vmhd", 0, 1, b"\0" * 8)
    dinf = _box("dinf", _fullbox("dref", 0, 0, struct.pack(">I", 1) + _fullbox("url ", 0, 1, b"")))
    ftyp = _box("ftyp", b"msf1" + struct.pack(">I", 0) + b"msf1heic")

    def make_moov(offsets):
        stco = _fullbox("stco", 0, 0, struct.pack(">I", len(offsets)) + b"".join(struct.pack(">I", o) for o in offsets))
        stbl = _box("stbl", stsd + stts + ctts + stsc + stsz + stco)
        minf = _box("minf", vmhd + dinf + stbl)
        mdia = _box("mdia", mdhd + hdlr + minf)
        return _box("moov", mvhd + _box("trak", tkhd + mdia))

    moov_len = len(make_moov([0] * len(chunks)))
    pos = len(ftyp) + moov_len + 8
    offsets = []
    for ch in chunks:
        offsets.append(pos)
        pos += sum(len(payloads[i]) for i in ch)
    return ftyp + make_moov(offsets) + _box("mdat", b"".join(payloads))
