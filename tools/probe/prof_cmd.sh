#!/bin/bash
# kernel-time summary of one tool run: bash tools/probe/prof_cmd.sh <name> <python script and args>
set -o pipefail
R="$GRAFT_REPO_ROOT"; NAME="$1"; shift
OUT="$R/gpurun_out/prof_$NAME"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$R/$1" "${@:2}" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
tail -2 "$OUT/log.txt"
python3 - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms" % (tot / 1e6))
    for r in rows[:22]:
        print("%-60s %6d  %9.3f ms  avg %8.1f us" % (r[0][:60], r[1], r[2] / 1e6, r[3] / 1e3))
PY
