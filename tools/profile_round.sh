#!/bin/bash
# Profiles the bench workload on the GPU box; writes raw output under gpurun_out/<tag>/.
#   tools/profile_round.sh r01
# Three separate rocprofv3 runs (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes (the TCC counters do not fit in one pass), then
# tools/profile_summary.py condenses them into the files committed under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
grep -h '"metric"' "$OUT"/*.log | head -4
find "$OUT" -name "*.csv" | head -20
