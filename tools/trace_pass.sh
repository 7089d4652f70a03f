#!/bin/bash
# One rocprofv3 kernel trace of a target; per-kernel totals to gpurun_out/trace/<name>_kernel_stats.csv.
# usage: tools/trace_pass.sh <name> <script> <args...>
set -o pipefail
R="$GRAFT_REPO_ROOT"
name="$1"; shift
OUT="$R/gpurun_out/trace"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/tr_$name"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "/tmp/tr_$name" -o tr -- python3 "$R/tools/$1" "${@:2}" > "$OUT/$name.log" 2>&1 || exit 1
db=$(find "/tmp/tr_$name" -name "tr_results.db" | head -1)
python3 - "$db" "$OUT/${name}_kernel_stats.csv" <<'PY'
import sys
sys.path.insert(0, __import__("os").environ["GRAFT_REPO_ROOT"] + "/tools")
from condense_profiles import kernel_stats
st = kernel_stats(sys.argv[1], sys.argv[2])
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["avg_ns"] * kv[1]["calls"])[:14]:
    print("%-90s %6d calls %10.1f us avg %10.1f us total" % (k[:90], v["calls"], v["avg_ns"] / 1e3, v["avg_ns"] * v["calls"] / 1e3))
PY
rm -rf "/tmp/tr_$name"
