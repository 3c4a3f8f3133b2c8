"""Static instruction mix of k_thompson_pack<512> PER SECTION of the level code (measurement aid, not product).

    python profiles/micro/th_static_sections.py > profiles/r06_thompson_sections.md

mp_thompson.hip is compiled with the product's flags + -gline-tables-only -S; every instruction of the kernel is attributed to the
source line of its innermost `.loc` (thompson_lane.inc / thompson_math.h / glibc_*.h / column_comm.h lines are attributed to the
thompson_lane.inc line that inlined them when the .loc carries an inlined-at chain; otherwise to their own file, listed as "callee
code"), and the lines are bucketed by the section boundaries of profiles/micro/th_sections.py (the same anchors as the s_memtime
stamps of round 5, whose wall-clock shares are quoted next to them).  STATIC counts: rocprofv3 PC sampling is not supported on this
pool and the counters are per dispatch, so there is no dynamic per-section count; branches skip parts of every section."""
import os, re, subprocess, sys, tempfile, collections
sys.path.insert(0, ".")
from icar_amd import build as B
sys.path.insert(0, os.path.join("profiles", "micro"))
import importlib.util
spec = importlib.util.spec_from_file_location("th_sections", os.path.join("profiles", "micro", "th_sections.py"))
src_sections = open(os.path.join("profiles", "micro", "th_sections.py")).read()
ANCHORS = eval(src_sections[src_sections.index("ANCHORS = [") + len("ANCHORS = "):src_sections.index("]\n\nSTAMP")] + "]")
inc = open(os.path.join(B.CSRC, "thompson_lane.inc")).read()


def line_of(anchor, occ):
    pos = -1
    for _ in range(occ):
        pos = inc.index(anchor, pos + 1)
    return inc.count("\n", 0, pos) + 1


bounds = [(name, line_of(a, occ)) for name, a, occ, f in ANCHORS if f == "inc"]          # section ENDS at its anchor line
bounds.sort(key=lambda x: x[1])


def section(line):
    for name, end in bounds:
        if line < end:
            return name
    return "finish: tendencies applied, stores"


with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "th.s")
    subprocess.check_call([B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS["mp_thompson.hip"] + ["-gline-tables-only", "--cuda-device-only", "-S",
                          os.path.join(B.CSRC, "mp_thompson.hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()
files, on, cur = {}, False, ("?", 0)
rows = collections.defaultdict(collections.Counter)
DIV = re.compile(r"v_div_(scale|fmas|fixup)|v_rcp_f")
for ln in text:
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", ln)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2)); continue
    if re.match(r"^_ZN12_GLOBAL__N_115k_thompson_packILi512E.*:", ln): on = True; continue
    if on and ln.strip().startswith(".end_amdhsa_kernel"): break
    if on and re.match(r"^_Z\w+:", ln): break
    if not on: continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    s = ln.strip()
    if not s or s[0] in ".;" or s.endswith(":"): continue
    op = s.split()[0]
    sec = section(cur[1]) if cur[0] == "thompson_lane.inc" else "callee code: " + cur[0]
    c = rows[sec]
    if op.startswith("v_"):
        c["VALU"] += 1
        if DIV.search(op): c["division helpers"] += 1
        elif op.startswith("v_cvt"): c["conversions"] += 1
        elif re.match(r"v_(mov|cndmask|cmp|readfirstlane|readlane|writelane|accvgpr|swap|perm|bfe|and|or|xor|not|lshl|lshr|ashr|add_u|sub_u|add_co|subb|addc|mad_u|mul_lo|mul_hi|min_[iu]|max_[iu]|ldexp|frexp)", op): c["moves / selects / compares / integer"] += 1
        elif "f64" in op: c["FP64 arithmetic"] += 1
        elif re.match(r"v_(exp|log|sqrt|rsq|sin|cos)_", op): c["transcendental"] += 1
        else: c["FP32 arithmetic"] += 1
    elif op.startswith("s_"):
        c["SALU"] += 1
        if op.startswith("s_cbranch") or op == "s_branch": c["branches"] += 1
    elif op.startswith("ds_"): c["LDS"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["VMEM"] += 1
        if op.startswith("scratch_"): c["scratch"] += 1
cols = ["VALU", "FP32 arithmetic", "FP64 arithmetic", "moves / selects / compares / integer", "division helpers", "conversions", "transcendental", "SALU", "branches", "LDS", "VMEM", "scratch"]
print(__doc__.split("\n\n")[0] + "\n")
print("Wall-clock shares of the same sections (s_memtime stamps): `profiles/r05_steps.md`.  Division helpers = v_div_scale / v_div_fmas / v_div_fixup / v_rcp (the fma\nsteps of a division are counted as arithmetic: an FP32 division is 11 VALU, 5 of them here).\n")
print("| section (ends at the anchor of th_sections.py) | " + " | ".join(cols) + " |"); print("|---" * (len(cols) + 1) + "|")
order = [n for n, _ in bounds] + ["finish: tendencies applied, stores"] + sorted(k for k in rows if k.startswith("callee"))
tot = collections.Counter()
for k in order:
    if k not in rows: continue
    print("| " + k + " | " + " | ".join(str(rows[k][c]) for c in cols) + " |"); tot.update(rows[k])
print("| **whole kernel** | " + " | ".join(str(tot[c]) for c in cols) + " |")
