#!/bin/bash
# round 4, GPU call 33: PMC counters of the small-blocks kernel on the CPL joint likelihood (what bounds its 24 us: instructions issued or waiting?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_33; mkdir -p $O
timeout -k 10 900 bash tools/pmc_profile.sh $O/pmc --workload desi_cmb_des5y --fde cpl > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
rm -rf $O/pmc/pass*/
grep -A32 "small_blocks" $O/pmc/pmc_summary.txt | head -40
