#!/bin/bash
# end of round 2: PMC counters of the headline configuration again (the solve's epilogue changed since the first collection)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02pmc2; mkdir -p $O
bash tools/pmc_profile.sh $O/config2_w4096 > $O/config2_w4096.log 2>&1; tail -3 $O/config2_w4096.log
cp $O/config2_w4096/pmc_traffic.json $O/pmc_traffic_config2_w4096.json; cp $O/config2_w4096/pmc_summary.txt $O/pmc_summary_config2_w4096.txt; rm -rf $O/config2_w4096/pass*/
cat $O/pmc_traffic_config2_w4096.json | head -30
