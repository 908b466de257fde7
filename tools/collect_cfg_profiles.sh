#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 evidence for BASELINE configs[2] / configs[4] (tools/bench_configs.py):
# kernel-trace stats, then FETCH_SIZE / WRITE_SIZE / LDS+SQ counters in separate passes.
#   bash tools/collect_cfg_profiles.sh r02_cfg35 [--frames N]
set -uo pipefail
TAG=${1:?tag}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/bench_configs.py $* > "$OUT.plain.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/bench_configs.py $* > "$OUT.trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/bench_configs.py $* > "$OUT.fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/bench_configs.py $* > "$OUT.write.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/sq" -- python3 tools/bench_configs.py $* > "$OUT.sq.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/sq2" -- python3 tools/bench_configs.py $* > "$OUT.sq2.log" 2>&1
python3 tools/summarise_profiles.py "$OUT" "$TAG"
grep "^{" "$OUT.plain.log" > "$PWD/gpurun_out/${TAG}_bench_lines.json"
