#!/usr/bin/env python
"""gpurun_out/parity/*.jsonl (written by the -m gpu parity tests on the MI355X box, tests/util.py:parity_record) ->
profiles/<round>_parity.json: per test label and field the MEASURED deviation from the CPU oracle
(bit-different fraction / cells, fraction beyond rtol 1e-5, max |d| / max|field|, max |d| / local scale), plus the worst
value per label.  The bounds asserted in tests/ are <= 2.5x these.  usage: python profiles/collect_parity.py [round, default r04]"""
import glob, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "parity", "*.jsonl"))):
    name = os.path.splitext(os.path.basename(f))[0]
    recs = {}
    for l in open(f):
        r = json.loads(l)
        recs.setdefault(r["label"], {}).update(r["fields"])  # (clear gpurun_out/parity before a run: records accumulate)
    summary = {}
    for lab, fields in recs.items():
        w = {}
        for st in fields.values():
            for k, v in st.items():
                if k != "cells" and isinstance(v, (int, float)):
                    w[k] = max(w.get(k, 0), v)
        summary[lab] = w
    out[name] = {"worst_per_label": summary, "per_field": recs}
json.dump(out, open(os.path.join(root, "profiles", RND + "_parity.json"), "w"), indent=1, sort_keys=True)
for name, v in out.items():
    for lab, w in v["worst_per_label"].items():
        print(f"{name:10s} {lab:70s} " + "  ".join(f"{k}={x:.3g}" for k, x in sorted(w.items())))
