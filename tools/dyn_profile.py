"""Estimated DYNAMIC instruction profile of a kernel by source line: static ISA count per line (hipcc -S -g) x execution
count per line from a gcov run of the host emulation (tests/emu built with --coverage).  Lines inside the emulated-lane
loops of parse_core.h (PC_VEC_BEGIN..PC_VEC_END) run 64x per wave step on the host and are scaled back.  Dev tool.
usage: dyn_profile.py <kernel.hip> <kernel-symbol-substring> <source-with-gcov> <file.gcov> [pixels]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern, sym, srcname, gcovf = sys.argv[1:5]
px = float(sys.argv[5]) if len(sys.argv) > 5 else None
out = "/tmp/isa_%s.s" % os.path.basename(kern)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-g", "-std=c++17", "-ffp-contract=off", "-I%s/include" % ROOT,
                       "-I%s/libheif_amd/csrc" % ROOT, "-S", "--cuda-device-only", "-o", out, kern], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(sym), l)][0]
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
static = collections.Counter(); static_cls = collections.Counter(); cur = None
for l in lines[start:]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), ""), int(m.group(2))); continue
    if l.startswith(".Lfunc_end"): break
    if re.match(r"\s+[a-z_0-9]+(\s|$)", l) and not l.strip().startswith((".", ";")):
        static[cur] += 1
        op = l.split()[0]
        cls = "B" if op.startswith(("s_cbranch", "s_branch")) else ("N" if op.startswith(("s_nop", "s_waitcnt")) else ("S" if op.startswith("s_") else ("V" if op.startswith("v_") else "M")))
        # round 6 (profiles/r06_issue_model_*.txt): a VALU instruction that reads or writes the scalar register file (SGPR operand, VCC, v_readlane,
        # v_cmp, v_cndmask ...) issues at half the rate of one that only touches VGPRs / inline constants: class "Vs" against "Vp"
        if cls == "V":
            args = l.split(";")[0].split(None, 1)[1] if len(l.split(";")[0].split(None, 1)) > 1 else ""
            sg = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0|scc)(?![a-z0-9_])", args)) or op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane"))
            static_cls[(cur, "Vs" if sg else "Vp")] += 1
            # SGPR spills live in VGPR lanes: a reload is v_readlane sN, vK, <constant lane>, a spill v_writelane vK, sN, <constant lane>
            if re.match(r"v_readlane_b32\s+s\d+,\s*v\d+,\s*\d+\s*$", l.split(";")[0].strip()): static_cls[(cur, "spill_reload")] += 1
            if re.match(r"v_writelane_b32\s+v\d+,\s*s\d+,\s*\d+\s*$", l.split(";")[0].strip()): static_cls[(cur, "spill_store")] += 1
        static_cls[(cur, cls)] += 1
st_line = collections.Counter()
unattributed = 0
for (f, ln), c in static.items():
    if f.endswith(srcname) and ln: st_line[ln] += c
    else: unattributed += c
print("static instructions: %d in %s, %d elsewhere / line 0" % (sum(st_line.values()), srcname, unattributed))
# gcov
dyn = {}
text = {}
for l in open(gcovf):
    m = re.match(r"\s*([0-9#=\-\*]+)\*?:\s*(\d+):(.*)", l)
    if not m: continue
    ln = int(m.group(2)); text[ln] = m.group(3)
    c = m.group(1)
    if c.rstrip("*").isdigit(): dyn[ln] = int(c.rstrip("*"))
# vector regions
invec = set(); on = False
for ln in sorted(text):
    t = text[ln]
    if "PC_VEC_BEGIN" in t and "define" not in t: on = True
    if on: invec.add(ln)
    if "PC_VEC_END" in t and "define" not in t: on = False
def count(ln):
    # nearest executed line at or before ln (declarations / continuation lines carry no count)
    for k in range(ln, max(ln - 6, 0), -1):
        if k in dyn:
            c = dyn[k]
            return c / 64.0 if k in invec else c   # the emulated-lane loop runs its header 65x and its body 64x per wave step
    return 0
# enclosing function of every line (heuristic) and the number of inlined copies of each function: every copy
# contributes at least one instruction to each executed statement, so the smallest static count of a function's
# lines estimates the number of copies; static counts are divided by it before weighting with the execution counts
func_at = {}; curf = "?"
for ln in sorted(text):
    t = text[ln]
    m = re.match(r"(?:PC_DEV|__device__|static|template|inline)\b.*?\b([a-z_0-9]+)\s*\(", t)
    if m and not t.startswith(" "): curf = m.group(1)
    func_at[ln] = curf
copies = {}
for ln, c in st_line.items():
    if count(ln) > 0:
        f = func_at.get(ln, "?"); copies[f] = min(copies.get(f, 1 << 30), c)
tot = 0; rows = []
cls_tot = collections.Counter()
for ln, c in st_line.items():
    k = max(1, copies.get(func_at.get(ln, "?"), 1))
    d = c / k * count(ln); tot += d; rows.append((d, ln, c / k, count(ln)))
    for (key, cls), cc in static_cls.items():
        if key[1] == ln and key[0].endswith(srcname): cls_tot[cls] += cc / k * count(ln)
print("by class (S salu, V valu = Vs scalar-file-touching + Vp pure, B branch, N nop/waitcnt, M memory/lds):", {k: round(v / (px or 1), 2) for k, v in cls_tot.items()})
srows = []
for (key, cls), cc in static_cls.items():
    if cls == "S" and key[0].endswith(srcname) and key[1]:
        ln = key[1]; k = max(1, copies.get(func_at.get(ln, "?"), 1))
        srows.append((cc / k * count(ln), ln, cc / k, count(ln)))
srows.sort(reverse=True)
if os.environ.get("SPILL"):
    for what in ("spill_reload", "spill_store"):
        vrows = []
        for (key, cls), cc in static_cls.items():
            if cls == what and key[0].endswith(srcname) and key[1]:
                ln = key[1]; k = max(1, copies.get(func_at.get(ln, "?"), 1))
                vrows.append((cc / k * count(ln), ln, cc / k, count(ln)))
        vrows.sort(reverse=True)
        print("-- %s: %.2f per pixel; top lines" % (what, sum(r[0] for r in vrows) / (px or 1)))
        for d, ln, c, k in vrows[:int(os.environ.get("TOPN", "45"))]: print("%5.2f/px line %4d static %5.1f x %9.0f  %s" % (d / (px or 1), ln, c, k, text[ln].strip()[:100]))
if os.environ.get("VS"):
    vrows = []
    for (key, cls), cc in static_cls.items():
        if cls == "Vs" and key[0].endswith(srcname) and key[1]:
            ln = key[1]; k = max(1, copies.get(func_at.get(ln, "?"), 1))
            vrows.append((cc / k * count(ln), ln, cc / k, count(ln)))
    vrows.sort(reverse=True)
    vbyf = collections.Counter()
    for d, ln, c, k in vrows: vbyf[func_at.get(ln, "?")] += d
    print("-- scalar-file-touching VALU by function:", {f: round(d / (px or 1), 2) for f, d in vbyf.most_common(14)})
    print("-- top scalar-file-touching VALU lines")
    for d, ln, c, k in vrows[:int(os.environ.get("TOPN", "45"))]: print("%5.2f/px line %4d static %5.1f x %9.0f  %s" % (d / (px or 1), ln, c, k, text[ln].strip()[:100]))
if os.environ.get("SALU"):
    sbyf = collections.Counter()
    for d, ln, c, k in srows: sbyf[func_at.get(ln, "?")] += d
    print("-- SALU by function:", {f: round(d / (px or 1), 2) for f, d in sbyf.most_common(12)})
    print("-- top SALU lines")
    for d, ln, c, k in srows[:int(os.environ.get("TOPN", "45"))]: print("%5.2f/px line %4d static %5.1f x %9.0f  %s" % (d / (px or 1), ln, c, k, text[ln].strip()[:100]))
rows.sort(reverse=True)
print("estimated dynamic wave-instructions: %.3g%s" % (tot, (" = %.1f per pixel" % (tot / px)) if px else ""))
byf = collections.Counter()
for d, ln, c, k in rows: byf[func_at.get(ln, "?")] += d
print("-- by function")
for f, d in byf.most_common(25): print("%6.2f%%  %s" % (100 * d / tot, f))
print("-- top lines")
for d, ln, c, k in rows[:int(os.environ.get("TOPN", "45"))]: print("%5.2f%% line %4d static %5.1f x %9.0f  %s" % (100 * d / tot, ln, c, k, text[ln].strip()[:100]))
