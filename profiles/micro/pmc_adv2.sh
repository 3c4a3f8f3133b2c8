#!/bin/bash
# SQ counters of k_mpdata_fused with nothing beside it (profiles/prof_advect.py), two passes; per-wave means printed
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
tag=${1:-adv}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); O=gpurun_out/pmc_$tag/s$i; rm -rf $O; mkdir -p $O
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p -- python profiles/prof_advect.py 512 3 > $O/log 2>&1
  python - "$O" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'mpdata_fused' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print({c: round(acc[c]/cnt[c]/1944, 1) for c in acc}, "(per wave, 243 blocks x 8 waves)")
P
done
