#!/bin/bash
# Kernel trace of back-to-back point log-likelihood calls: medians of the two kernels' durations and of the idle
# time between them (terms -> assemble inside a call, assemble -> next call's terms).
# usage: tools/loglik_gaps.sh [instances] [landmarks]
set -o pipefail
R="$GRAFT_REPO_ROOT"
B="${1:-512}"; L="${2:-150}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_gaps
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/tr_gaps -o tr -- python3 "$R/tools/run_loglik.py" "$B" "$L" 60 > /tmp/tr_gaps.log 2>&1 || { tail -5 /tmp/tr_gaps.log; exit 1; }
db=$(find /tmp/tr_gaps -name "tr_results.db" | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "loglik_" in r[0]]
rows = rows[20:]                                   # past the first calls
med = lambda v: sorted(v)[len(v) // 2] / 1e3 if v else float("nan")
dur = {"terms": [], "assemble": []}
gap_in, gap_out, period = [], [], []
for i, (name, s, e) in enumerate(rows):
    k = "terms" if "terms" in name else "assemble"
    dur[k].append(e - s)
    if i + 1 < len(rows):
        g = rows[i + 1][1] - e
        (gap_in if k == "terms" else gap_out).append(g)
    if k == "terms" and i + 2 < len(rows):
        period.append(rows[i + 2][1] - s)
print("terms %.2f us, assemble %.2f us, gap terms->assemble %.2f us, gap assemble->next terms %.2f us, period %.2f us"
      % (med(dur["terms"]), med(dur["assemble"]), med(gap_in), med(gap_out), med(period)))
PY
rm -rf /tmp/tr_gaps
