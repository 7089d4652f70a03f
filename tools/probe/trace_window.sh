# usage: bash tools/probe/trace_window.sh <first> <count> <python script and arguments ...>: kernel trace of one command; prints `count` consecutive launches (start relative to the first, duration, gap to the previous end) starting `first` launches before the end
cd /tmp && export TMPDIR=/tmp
FIRST=$1; COUNT=$2; shift; shift
D=$GRAFT_REPO_ROOT/gpurun_out/prof_tmp
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_cmd.log 2>&1 || exit 1
python3 - $FIRST $COUNT $(find $D -name '*.db' | head -1) <<'PY'
import sqlite3, sys
first, count = int(sys.argv[1]), int(sys.argv[2]); db = sqlite3.connect(sys.argv[3])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[len(rows) - first:][:count]
t0, prev = rows[0][1], rows[0][1]
for name, a, b, gx, gy in rows:
    print("%9.1f us  +%7.1f us  gap %6.1f  grid %6d %4d  %s" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev) / 1e3, gx, gy, name[:48]))
    prev = b
PY
rm -rf $D
