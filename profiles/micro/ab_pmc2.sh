#!/bin/bash
# I-cache / fetch counters of k_thompson_pack (interior launch) for the product library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_LEVEL\|INSTS_\)[A-Z_0-9]*" | sort -u | tr '\n' ' '; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"; do
  O=gpurun_out/abpmc2/$(echo $set | cut -d' ' -f1); mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/log 2>&1
  python - "$O" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
if not f: print('no csv', open(sys.argv[1]+'/log').read()[-600:]); sys.exit()
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'thompson_pack' not in r['Kernel_Name'] or int(r['Grid_Size']) < 5000000: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print({c: round(acc[c]/cnt[c]) for c in acc})
P
done
