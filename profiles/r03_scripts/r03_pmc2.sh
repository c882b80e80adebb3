#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_pmc; mkdir -p $O
CF_WALKER_GENERIC=1 bash tools/pmc_quick.sh $O/config3_cpl_generic_walker --workload desi_cmb_des5y --fde cpl > $O/config3_cpl_generic.log 2>&1; tail -3 $O/config3_cpl_generic.log
find $O -name "*.csv" -size +200k -delete; find $O -name "*agent_info.csv" -delete
