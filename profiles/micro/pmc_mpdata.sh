#!/bin/bash
# instruction classes of k_mpdata_fused, per wave
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1)); O=gpurun_out/pmc_mpdata/s$i; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/log 2>&1
  python - "$O" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'mpdata_fused' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
w=acc.get('SQ_WAVES',0)/max(cnt.get('SQ_WAVES',1),1) or 1984
print({c: round(acc[c]/cnt[c]/1984, 1) for c in acc})
P
done
