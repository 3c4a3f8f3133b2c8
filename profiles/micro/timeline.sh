#!/bin/bash
# kernel timeline of the last step of a bench run (rocprofv3 --kernel-trace): timeline.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/tl_$tag; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/t.log 2>&1
python - "$O" "$tag" <<'PY'
import sqlite3, glob, sys, re
fn = glob.glob(sys.argv[1] + '/t/**/*.db', recursive=True)[0]
con = sqlite3.connect(fn)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in con.execute(f"pragma table_info({kd})")]
q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else 'nid')
rows = con.execute(f"select s.kernel_name, d.start, d.end, d.{q} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    return re.sub(r'^_Z+N?\d*_?GLOBAL__N_1?\d*', '', n)[:40]
# last full step: from the second-to-last k_diag_cell to the last one
idx = [i for i, r in enumerate(rows) if 'k_diag_cell' in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
print("==", sys.argv[2], "step of %.1f us" % ((rows[b][1] - t0) / 1e3))
prev_end = {}
for n, s, e, qid in rows[a:b]:
    gap = (s - prev_end[qid]) / 1e3 if qid in prev_end else 0.0
    print(f"  q{qid} {short(n):40s} start {(s - t0)/1e3:8.1f} dur {(e - s)/1e3:8.1f} gap-on-queue {gap:7.1f}")
    prev_end[qid] = e
PY
