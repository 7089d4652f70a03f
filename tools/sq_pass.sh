#!/bin/bash
# One rocprofv3 counter pass (SQ set of the corner profile) per target; summaries to gpurun_out/sq/<name>.json.
# usage: tools/sq_pass.sh <name> <script> <args...>
set -o pipefail
R="$GRAFT_REPO_ROOT"
name="$1"; shift
OUT="$R/gpurun_out/sq"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
rm -rf "/tmp/sq_$name"
timeout -s KILL 150 rocprofv3 --pmc $CTRS -d "/tmp/sq_$name" -o sq -- python3 "$R/tools/$1" "${@:2}" > "$OUT/$name.log" 2>&1 || exit 1
db=$(find "/tmp/sq_$name" -name "sq_results.db" | head -1)
python3 "$R/tools/sq_counters.py" "$db" "$OUT/$name.json" "rocprofv3 --pmc $CTRS -- python3 tools/$*" vslam orb_pack || exit 1
rm -rf "/tmp/sq_$name"
