"""Aggregate rocprofv3 PC-sampling CSVs on the GPU box (the raw files are too big to bring back).

usage: python profiles/micro/pcs_agg.py <dir> <out.md> [kernel_substring]
Prints the header + a few raw rows of every *pc_sampling*.csv under <dir>, then sample counts per column value (columns
with few distinct values), per source line (Instruction_Comment) and per instruction text.  Measurement aid, not product.
"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
root, out = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(root, "**", "*pc_sampling*.csv"), recursive=True))
lines = []
P = lines.append
for f in files:
    P(f"## {f} ({os.path.getsize(f) / 1e6:.1f} MB)")
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd, None)
        if not hdr:
            continue
        P("columns: " + ", ".join(hdr))
        cols = {h: i for i, h in enumerate(hdr)}
        per_col = [collections.Counter() for _ in hdr]
        by_line = collections.Counter()
        by_line_issued = collections.Counter()
        by_line_stall = collections.defaultdict(collections.Counter)
        by_inst = collections.Counter()
        ci = cols.get("Instruction_Comment")
        ii = cols.get("Instruction")
        wi = cols.get("Wave_Issued_Instruction")
        si = cols.get("Stall_Reason")
        n = 0
        for row in rd:
            if n < 3:
                P("row: " + " | ".join(row))
            n += 1
            for j, v in enumerate(row):
                if len(per_col[j]) < 3000:
                    per_col[j][v] += 1
            if ci is not None:
                key = row[ci]
                by_line[key] += 1
                if wi is not None and row[wi] in ("1", "true", "True"):
                    by_line_issued[key] += 1
                if si is not None:
                    by_line_stall[key][row[si]] += 1
            if ii is not None:
                by_inst[row[ii].split(" ")[0]] += 1
        P(f"samples: {n}")
        for j, h in enumerate(hdr):
            if 1 < len(per_col[j]) < 200:
                P(f"### by {h}")
                for v, c in per_col[j].most_common(60):
                    P(f"{c:9d} {100.0 * c / max(n, 1):6.2f}%  {v}")
        if by_inst:
            P("### by opcode")
            for v, c in by_inst.most_common(80):
                P(f"{c:9d} {100.0 * c / max(n, 1):6.2f}%  {v}")
        if by_line:
            P("### by source line (samples, share, issued share of the line, top stall reasons)")
            for v, c in sorted(by_line.items(), key=lambda kv: -kv[1])[:400]:
                st = ", ".join(f"{k}:{m}" for k, m in by_line_stall[v].most_common(3)) if v in by_line_stall else ""
                P(f"{c:9d} {100.0 * c / max(n, 1):6.2f}%  iss {100.0 * by_line_issued[v] / c:5.1f}%  {v}  [{st}]")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
