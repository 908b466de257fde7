#!/bin/bash
# developer loop for stage 1: parity tests, soak, kernel times under rocprofv3
set -o pipefail
TAG=${1:-s1}
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_multicam.py -x -q --timeout 600 --timeout-method thread -k "not fps" > gpurun_out/${TAG}_t.log 2>&1; echo rc=$? >> gpurun_out/${TAG}_t.log; tail -3 gpurun_out/${TAG}_t.log
timeout -k 10 200 python tests/stress_stage1.py --seconds ${SOAK:-40} 2>&1 | tail -1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-verify > gpurun_out/prof_${TAG}.log 2>&1
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].split("(")[0]
    if any(t in n for t in ("k1_","fps_","gather")): print(f'{n[:60]:60s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"])/1e3:9.1f} us')
P
