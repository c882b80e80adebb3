#!/bin/bash
# round 3, GPU call 4: GPU tests (table-driven exp, timing split), bench lines of configs 2 / 3 (Lambda, CPL) / 5 with the CPU leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_config2.json 2> $O/bench_config2.err || { tail -5 $O/bench_config2.err; exit 1; }
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --workload desi_cmb_des5y > $O/bench_config3_lcdm.json 2> $O/bench_config3_lcdm.err || { tail -5 $O/bench_config3_lcdm.err; exit 1; }
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --workload desi_cmb_des5y --fde cpl > $O/bench_config3_cpl.json 2> $O/bench_config3_cpl.err || { tail -5 $O/bench_config3_cpl.err; exit 1; }
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --workload desi_des5y_bbn_theta_star > $O/bench_config5.json 2> $O/bench_config5.err || { tail -5 $O/bench_config5.err; exit 1; }
python - <<PY
import json
for n in ("config2","config3_lcdm","config3_cpl","config5"):
    d=json.load(open("$O/bench_%s.json"%n))
    c=d.get("cpu_baseline",{})
    print(n, "value %.3e ms/step %.4f kernels %s frac %.3f | cpu %.3e (%s cores) parity %.2e"%(d["value"],d["ms_per_step"],{k:(round(v,4) if v else v) for k,v in d["kernels_ms"].items()},d["roofline"]["frac"],c.get("value",0),c.get("cores"),c.get("parity_max_rel",-1)))
PY
