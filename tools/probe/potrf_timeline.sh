# usage: bash tools/probe/potrf_timeline.sh [L]: kernel trace of one device-resident update of L landmarks; prints, for one
# blocked Cholesky in the middle of it, the time per kernel in each tenth of its launches
cd /tmp && export TMPDIR=/tmp
L=${1:-2000}
D=$GRAFT_REPO_ROOT/gpurun_out/prof_tmp
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t -- python3 $GRAFT_REPO_ROOT/tools/run_update_dev1.py $L 1 > $GRAFT_REPO_ROOT/gpurun_out/trace_cmd.log 2>&1 || exit 1
python3 - $(find $D -name '*.db' | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if 'potf2_upper' in r[0]]
i0 = idx[len(idx) * 3 // 4]
i1 = min([i for i in idx if i > i0] + [len(rows)])
seq = [r for r in rows[i0:i1] if 'trsm' in r[0] or 'chol_update' in r[0] or 'potf2' in r[0]]
print("factorisation: %d launches, span %.1f us" % (len(seq), (seq[-1][2] - seq[0][1]) / 1e3))
n = len(seq)
for t in range(10):
    part = seq[n * t // 10: n * (t + 1) // 10]
    tr = sum(r[2] - r[1] for r in part if 'trsm' in r[0]) / 1e3
    up = [(r[2] - r[1]) / 1e3 for r in part if 'chol_update' in r[0]]
    big = [u for r, u in zip([r for r in part if 'chol_update' in r[0]], up)]
    span = (part[-1][2] - part[0][1]) / 1e3
    print("tenth %d: span %7.1f us, trsm %6.1f us in %d, updates %7.1f us in %d (min %.1f max %.1f)" % (t, span, tr, sum('trsm' in r[0] for r in part), sum(up), len(up), min(up), max(up)))
PY
rm -rf $D
