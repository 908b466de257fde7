#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py,
# then tools/summarise_profiles.py folds them into gpurun_out/<tag>_* files to be copied into profiles/.
#   bash tools/collect_profiles.sh r01_c [bench args...]
set -euo pipefail
TAG=${1:?tag}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --no-secondary --no-verify $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT.trace.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-secondary --no-verify $* > "$OUT.fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-secondary --no-verify $* > "$OUT.write.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-secondary --no-verify $* > "$OUT.sq.log" 2>&1
# one L2 hit/miss pass (TCC_HIT / TCC_MISS summed over the channels)
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/l2" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode --no-secondary --no-verify $* > "$OUT.l2.log" 2>&1 || echo "l2 pass failed (see $OUT.l2.log)"
python3 tools/summarise_profiles.py "$OUT" "$TAG"
