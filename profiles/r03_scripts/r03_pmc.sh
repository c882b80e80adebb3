#!/bin/bash
# round 3 PMC passes (separate --pmc runs, --kernel-trace only): config 2 in full (HBM traffic for roofline.traffic), and the config-3 CPL
# joint with the lean and with the generic walker kernel (two quick passes each)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_pmc; mkdir -p $O
tools/pmc_profile.sh $O/config2 > $O/config2.log 2>&1; tail -3 $O/config2.log
tools/pmc_profile.sh $O/config3_cpl --workload desi_cmb_des5y --fde cpl > $O/config3_cpl.log 2>&1; tail -3 $O/config3_cpl.log
CF_WALKER_GENERIC=1 tools/pmc_quick.sh $O/config3_cpl_generic_walker --workload desi_cmb_des5y --fde cpl > $O/config3_cpl_generic.log 2>&1; tail -3 $O/config3_cpl_generic.log
for d in config2 config3_cpl config3_cpl_generic_walker; do echo "== $d"; cat $O/$d/pmc_summary.txt | grep -A40 "walker" | head -45; done > $O/walker_pmc.txt
find $O -name "*.csv" -size +200k -delete; find $O -name "*agent_info.csv" -delete
ls $O/config2
