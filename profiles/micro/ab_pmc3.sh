#!/bin/bash
# VALU instruction classes of k_thompson_pack (interior launch) for one library (ICAR_HIP_LIB) -- per-wave averages
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); O=gpurun_out/abpmc3/s$i; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o p -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/log 2>&1
  python - "$O" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
if not f: print('no csv', open(sys.argv[1]+'/log').read()[-600:]); sys.exit()
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'thompson_pack' not in r['Kernel_Name'] or int(r['Grid_Size']) < 5000000: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print({c: round(acc[c]/cnt[c]/172720, 1) for c in acc})
P
done
