#!/bin/bash
# profiles/micro/ab_pmc.sh -- SQ instruction counters of k_thompson_pack per library variant under icar_amd/lib/ab/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
for so in icar_amd/lib/ab/lib_*.so; do
  n=$(basename $so .so); O=gpurun_out/abpmc/$n; mkdir -p $O
  ICAR_HIP_LIB=$R/$so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d $O -o p -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/log 2>&1
  python - "$O" "$n" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'thompson_pack' not in k: continue
    if int(r['Grid_Size']) < 5000000: continue
    acc[r['Counter_Name']]['v']+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print(sys.argv[2], {c: round(acc[c]['v']/cnt[c]) for c in acc})
P
done
