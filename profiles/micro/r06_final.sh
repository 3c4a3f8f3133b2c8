export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.4e ms %.4f advect %.4f frac %.4f mp %.4f traffic %.3e cpu %s' % (d['value'], d['ms_per_step'], r['avg_ms'], r['frac'], r['mp_ms_per_step'], r['traffic'] or 0, d['cpu_baseline']['value']))"
