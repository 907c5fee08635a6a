#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes
# of one short bench.py run.  Output: gpurun_out/prof_<tag>/ (scratch); copy the
# summaries you want judged into profiles/.
#   bash tools/profile.sh <tag> [bench args...]
set -e
TAG=${1:-r1}; shift || true
ARGS=${@:-"--steps 5 --warmup 2 --no-cpu-baseline --no-extra"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
echo "== pmc SQ"; rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || tail -5 $OUT/pmc_sq.log
echo "== pmc SQ2"; rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || tail -5 $OUT/pmc_sq2.log
echo "== pmc FETCH"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || tail -5 $OUT/pmc_fetch.log
echo "== pmc WRITE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || tail -5 $OUT/pmc_write.log
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1 || true
cat $OUT/summary.md
