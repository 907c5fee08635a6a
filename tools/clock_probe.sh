#!/bin/bash
# GPU box: effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of demod_rows for library variants.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in $VARIANTS; do
  for wl in ${WLS:-c3}; do
  OUT=$REPO/gpurun_out/clk_${v}_$wl; rm -rf $OUT; mkdir -p $OUT
  DEMOD_HIP_LIBRARY=$REPO/demodulator_amd/variants/libdemod_hip_$v.so rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 5 --warmup 2 --workload $wl --no-cpu-baseline --no-extra > $OUT/log.txt 2>&1
  python3 - $OUT $v $wl <<'PY'
import csv,glob,sys,collections
out,v,wl=sys.argv[1:4]
dur={}
for f in glob.glob(out+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'demod_rows' in r['Kernel_Name']:
            dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
acc=collections.defaultdict(list)
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'demod_rows' in r['Kernel_Name']:
            acc[r['Counter_Name']].append((r['Dispatch_Id'],float(r['Counter_Value'])))
if not dur: print(v,wl,'no trace'); sys.exit()
ids=sorted(dur)[2:]
d=sum(dur[i] for i in ids)/len(ids)
def avg(c):
    vals=[x for i,x in acc[c] if i in ids]
    return sum(vals)/len(vals) if vals else float('nan')
g=avg('GRBM_GUI_ACTIVE')
print(f"{v} {wl}: dur {d/1e6:.3f} ms  clock {g/8/d:.3f} GHz  VALU insts {avg('SQ_INSTS_VALU'):.4g}  valu-active(quad) {avg('SQ_ACTIVE_INST_VALU'):.4g}  wave-cycles(quad) {avg('SQ_WAVE_CYCLES'):.4g}")
PY
  done
done
