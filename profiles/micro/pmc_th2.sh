#!/bin/bash
# lane activity and wait shares of k_thompson_pack for the builds under icar_amd/lib/ab (prof_thompson.py 512)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; export TMPDIR=/tmp
for so in icar_amd/lib/ab/lib_*.so; do
  n=$(basename $so .so); O=gpurun_out/pmc_th_$n; rm -rf $O; mkdir -p $O
  ICAR_HIP_LIB=$R/$so timeout -s KILL 100 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- python profiles/prof_thompson.py ${NX:-256} > $O/log 2>&1
  ICAR_HIP_LIB=$R/$so timeout -s KILL 100 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/b -o p -- python profiles/prof_thompson.py ${NX:-256} > $O/log2 2>&1
  python - "$O" "$n" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv', recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter(); mx=0
for fn in f:
  for r in csv.DictReader(open(fn)):
    if 'thompson_pack' not in r['Kernel_Name']: continue
    if int(r['Grid_Size']) < 1000000: continue          # the interior launch only
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
a={c: acc[c]/max(cnt[c],1) for c in acc}
print(sys.argv[2], {k: '%.3g'%v for k,v in a.items()})
print('   lanes active per VALU instr: %.1f %%' % (100*a['SQ_THREAD_CYCLES_VALU']/(64*a['SQ_ACTIVE_INST_VALU'])), ' wait_any/wave_cycles %.1f %%' % (100*a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES']), ' VALU/wave %.0f' % (a['SQ_INSTS_VALU']/a['SQ_WAVES']))
P
done
