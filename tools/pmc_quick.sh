#!/bin/bash
# two PMC passes (SQ busy / MFMA counters, GRBM cycles) of the default bench: tools/pmc_quick.sh outdir [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pass$i -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/pass$i.json 2> $out/pass$i.err || { echo "pass $i failed"; tail -5 $out/pass$i.err; }
done
python tools/pmc_summary.py $out
