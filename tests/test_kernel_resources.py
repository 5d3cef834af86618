"""Resource budgets of the compiled gfx950 kernels, read from the code objects embedded in libheifhip.so (no GPU needed).
Guards the two things that silently cost performance on this path: scratch memory in a streaming kernel (a dynamically indexed
register array once made k_sao write 9 B/px) and a kernel sliding over an occupancy step (VGPRs / LDS per workgroup)."""
import os
import re
import struct
import subprocess
import tempfile
import pytest

import libheif_amd

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernels():
    so = libheif_amd.library_path()
    if not os.path.exists(so) or not os.path.exists(READELF):
        pytest.skip("built library or llvm-readelf not available")
    data = open(so, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = {}, 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p); p += 24
            triple = data[p:p + tl].decode(); p += tl
            if "amdgcn" in triple and size:
                with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                    f.write(data[i + off:i + off + size]); name = f.name
                txt = subprocess.run([READELF, "--notes", name], capture_output=True, text=True).stdout
                os.unlink(name)
                for blk in txt.split("- .agpr_count")[1:]:
                    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                    if g("name"):
                        out[g("name").group(1)] = dict(vgpr=int(g("vgpr_count").group(1)), scratch=int(g("private_segment_fixed_size").group(1)),
                                                       lds=int(g("group_segment_fixed_size").group(1)))
        pos = i + 32
    return out


def _find(ks, *parts):
    hits = [v for k, v in ks.items() if all(p in k for p in parts)]
    assert hits, "kernel %s not found in libheifhip.so" % (parts,)
    return hits


def test_streaming_kernels_use_no_scratch_memory():
    ks = _kernels()
    for parts in (("k_sao",), ("k_deblock",), ("k_ycbcr_to_rgb",), ("k_bilinear",), ("k_to_sdr",), ("k_residual",), ("k_recon16",)):
        for k in _find(ks, *parts):
            assert k["scratch"] == 0, parts


def test_occupancy_budgets():
    ks = _kernels()
    parse8 = _find(ks, "k_parse_occ8")[0]
    assert parse8["vgpr"] <= 64 and parse8["scratch"] <= 184          # the pool-mode variant, 8 waves / SIMD; the spills (kernel arguments and
                                                                      # per-picture bases around the inlined row parser) sit in per-CTB / per-row code
                                                                      # (120 B before pcm_sample and the dependent-slice-segment state hand-over, 148 B with them, 164 B with the; 180 B in round 6: LDS-resident contexts,
                                                                      #  operand registers of the hand-scheduled CABAC statements of round 3)
    parse6 = _find(ks, "k_parse_occ6")[0]
    assert parse6["vgpr"] <= 80 and parse6["scratch"] <= 128
    gen8 = _find(ks, "k_parse_gen_occ8")[0]                               # the build with the 4:2:2 / 4:4:4 paths (batches that hold such pictures)
    assert gen8["vgpr"] <= 64 and gen8["scratch"] <= 192
    recon8 = _find(ks, "8k_recon8E")[0]
    # 7 waves / SIMD by registers, 26 one-wave groups per CU by LDS (512 B granules); the spills sit in the per-wave / per-CTB code around the block
    # loop (scalar registers parked in VGPR lanes, 172 B of scratch; 244 B with the register-resident small-block forms of round 6), not in the block functions
    assert recon8["vgpr"] <= 72 and recon8["scratch"] <= 280 and recon8["lds"] <= 6144
    residual = _find(ks, "k_residual")[0]
    assert residual["lds"] <= 23040 and residual["vgpr"] <= 72          # 7 workgroups of 4 waves per CU: LDS is handed out in 512 B granules, 7 x 23040 <= 160 KB;
                                                                          # 7 waves per SIMD by registers (round 6: the thread-per-row form with v_dot2 butterflies, 70 VGPRs)
    assert _find(ks, "k_parse")[0]["scratch"] == 0      # (the unconstrained variant lone stills run)
    for name, k in ks.items():
        if "k_sao_rgb_lean" in name:
            # staged tiles + 16 KB of colour terms per tile: 6 workgroups per CU by LDS, 4 waves per SIMD by registers (the next tile's words travel in
            # registers while the current one is worked on)
            assert k["lds"] <= 24576 and k["vgpr"] <= 128 and k["scratch"] == 0
        elif "k_sao" in name:
            assert k["lds"] <= 10240
