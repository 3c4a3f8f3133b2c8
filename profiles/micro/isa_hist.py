"""Opcode histogram of the hottest loop of a kernel in a hipcc -save-temps .s file (measurement aid, not product).

usage: python profiles/micro/isa_hist.py file.s kernel_symbol_substring [--loop N] [--dump]
Finds the backward branches of the kernel, ranks the loops by instruction count and prints the opcode histogram of
loop N (default 0 = the largest): VALU / SALU / VMEM / LDS totals, the v_pk_* count and the v_mov count.
"""
import collections
import re
import sys


def kernel_lines(path, sym):
    out, on = [], False
    for ln in open(path):
        if re.match(r"^[A-Za-z_.$][\w.$]*:", ln) and not ln.startswith(".L") and not ln.startswith("; "):
            name = ln.split(":")[0]
            if on and not name.startswith(".L"):
                break
            if sym in name:
                on = True
                continue
        if on:
            if ln.strip().startswith(".section") or ln.strip().startswith(".end_amdhsa_kernel"):
                break
            out.append(ln.rstrip("\n"))
    return out


def parse(lines):
    ins, labels = [], {}
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB[\w]+):", s)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(s.split(";")[0].strip())
    return ins, labels


def klass(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    which = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 0
    ins, labels = parse(kernel_lines(path, sym))
    print(f"kernel {sym}: {len(ins)} instructions")
    loops = []
    for i, s in enumerate(ins):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)", s)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= i:
                loops.append((i - tgt + 1, tgt, i))
    loops.sort(reverse=True)
    for n, (sz, a, b) in enumerate(loops[:6]):
        print(f"  loop {n}: instructions {a}..{b} ({sz})")
    if not loops:
        return
    sz, a, b = loops[which]
    body = ins[a:b + 1]
    h = collections.Counter(x.split()[0] for x in body)
    kl = collections.Counter()
    for op, c in h.items():
        kl[klass(op)] += c
    print(f"loop {which}: {sz} instructions:", dict(kl))
    dpp = sum(1 for x in body if "dpp" in x.split()[0] or "row_" in x or "wave_sh" in x)
    print("  v_pk_*:", sum(c for op, c in h.items() if op.startswith("v_pk_")), " v_mov_b32:", h.get("v_mov_b32_e32", 0) + h.get("v_mov_b32", 0),
          " dpp-modified:", dpp, " v_rcp:", h.get("v_rcp_f32_e32", 0), " v_cndmask:", sum(c for op, c in h.items() if op.startswith("v_cndmask")),
          " readlane/writelane:", sum(c for op, c in h.items() if "lane" in op))
    for op, c in h.most_common(60):
        print(f"    {c:5d}  {op}")
    if "--dump" in sys.argv:
        for x in body:
            print(x)


if __name__ == "__main__":
    main()
