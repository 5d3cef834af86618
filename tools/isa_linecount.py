"""Static instruction histogram of one kernel by source line (hipcc -S -g output); dev tool.
usage: isa_linecount.py <file.hip> <kernel-symbol-substring> [min-count]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, sym = sys.argv[1], sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 10
out = "/tmp/isa_%s.s" % os.path.basename(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-g", "-std=c++17", "-ffp-contract=off", "-I%s/include" % ROOT,
                       "-I%s/libheif_amd/csrc" % ROOT, "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(sym), l)][0]
print(lines[start].split(":")[0])
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cnt = collections.Counter(); cur = None; total = 0
for l in lines[start:]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if l.startswith(".Lfunc_end"): break
    if re.match(r"\s+[a-z_0-9]+(\s|$)", l) and not l.strip().startswith((".", ";")):
        cnt[cur] += 1; total += 1
print("total instructions", total)
text = open(src).read().split("\n")
agg = collections.Counter(); other = 0
for (f, ln), c in cnt.items():
    if files.get(f, "").endswith(os.path.basename(src)): agg[ln] += c
    else: other += c
print("from other files / line 0:", other + agg.get(0, 0))
for ln in sorted(agg):
    if ln and agg[ln] >= thr: print("%5d %4d  %s" % (ln, agg[ln], text[ln - 1].strip()[:110]))
for l in lines:
    if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name):", l) and ("count" in l or "size" in l or sym in l): print(l.strip())
