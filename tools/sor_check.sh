#!/bin/bash
# developer loop for the config-3 filters: parity tests, soak, per-kernel times of tools/bench_configs.py
timeout -k 10 400 python -u -m pytest tests/test_gpu_filters.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
timeout -k 10 200 python tests/stress_filters.py --seconds ${SOAK:-30} 2>&1 | tail -1
export TMPDIR=/tmp; rm -rf gpurun_out/prof_sorc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sorc -- python3 tools/bench_configs.py --frames 64 2>/dev/null | grep "^{" | head -1 | cut -c1-330
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/prof_sorc/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "sor" in r["Name"]: print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
P
