#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (<prefix>_counter_collection.csv): mean counter value per dispatch of each
kernel + mean duration, as a markdown table.  Several CSVs (separate --pmc passes) may be given; they are merged by
kernel name.  usage: summarize_pmc.py out.md a_counter_collection.csv [b_counter_collection.csv ...]

Derived columns (MI355X: 256 CUs x 4 SIMDs, SQ_ACTIVE_INST_* in quad-cycles -- MI355X_MICROARCH.md):
  VALUBusy% = 100 * 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   (gfx9 derived-metric formula;
              GRBM_GUI_ACTIVE is summed over the 8 XCDs: it reads 8 x 2.1 GHz x duration)
  cyc/VALU  = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU   (average issue cycles per VALU instruction)
  HBM read GB/s  = 2 * FETCH_SIZE(KB) * 1024 / duration   (gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B)
  HBM write GB/s = WRITE_SIZE(KB) * 1024 / duration       (uncalibrated)"""
import csv
import sys
from collections import defaultdict


def main(out, files):
    val = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            val[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    names = sorted({c for k in val for c in val[k]})
    rows = []
    for k in val:
        d = sum(dur[k]) / len(dur[k])
        m = {c: sum(v) / len(v) for c, v in val[k].items()}
        extra = {}
        if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            extra["VALUBusy%"] = 100 * 4 * m["SQ_ACTIVE_INST_VALU"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_INSTS_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_INSTS_VALU"] > 0:
            extra["cyc/VALU"] = 4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"] > 0:
            extra["VALU/wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "FETCH_SIZE" in m:
            extra["HBM_rd_GB/s(x2)"] = 2 * m["FETCH_SIZE"] * 1024 / d
            extra["HBM_rd_MB(x2)"] = 2 * m["FETCH_SIZE"] / 1024
        if "WRITE_SIZE" in m:
            extra["HBM_wr_GB/s"] = m["WRITE_SIZE"] * 1024 / d
            extra["HBM_wr_MB"] = m["WRITE_SIZE"] / 1024
        rows.append((sum(dur[k]), k, len(dur[k]), d, m, extra))
    rows.sort(reverse=True)
    ex = sorted({e for r in rows for e in r[5]})
    hdr = ["kernel", "calls", "avg us"] + names + ex
    lines = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
    for _, k, n, d, m, e in rows:
        if d < 5e3:
            continue
        cells = [f"`{k[:60]}`", str(n), f"{d/1e3:.1f}"] + [f"{m[c]:.4g}" if c in m else "" for c in names] + [f"{e[x]:.4g}" if x in e else "" for x in ex]
        lines.append("| " + " | ".join(cells) + " |")
    txt = "\n".join(lines)
    print(txt)
    if out != "-":
        open(out, "a").write(txt + "\n\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
