#!/bin/bash
# One GPU-box visit: the -m gpu suite (every failure listed, not only the first), smoke(), and the default bench line.
#   tools/gpu_check.sh <tag> [pytest args...]
set -u
TAG=${1:-check}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python -m pytest tests -m gpu -q -rf --durations=15 "$@" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" >> "$OUT/smoke.log"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" >> "$OUT/bench.err"
tail -30 "$OUT/pytest.log"; tail -5 "$OUT/smoke.log"; cut -c1-1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
