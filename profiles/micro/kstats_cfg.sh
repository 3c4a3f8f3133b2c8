#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) for one bench configuration: kstats_cfg.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
tag=$1; shift
O=gpurun_out/ks_$tag; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $O/t.log 2>&1
python - "$O" "$tag" <<'PY'
import sqlite3, glob, sys
fn = glob.glob(sys.argv[1] + '/t/**/*.db', recursive=True)[0]
con = sqlite3.connect(fn)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.start), max(d.end) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
print("==", sys.argv[2])
for n, c, a, t, _, _ in rows[:16]:
    if 'qr_acr' in n or 'freez' in n: continue
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    import re
    n = re.sub(r'^_Z+N?\d*_?GLOBAL__N_1?\d*', '', n)[:44]
    print(f"  {n:44s} {c:5d} {a/1e3:9.1f} us")
PY
