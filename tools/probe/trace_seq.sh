# usage: bash tools/probe/trace_seq.sh <min_us> <python script and arguments ...>: kernel trace of one command; prints, in launch order, every kernel of the last third of the run that took at least min_us
cd /tmp && export TMPDIR=/tmp
MIN=$1; shift
D=$GRAFT_REPO_ROOT/gpurun_out/prof_tmp
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_cmd.log 2>&1 || exit 1
python3 - $MIN $(find $D -name '*.db' | head -1) <<'PY'
import sqlite3, sys
mn = float(sys.argv[1]); db = sqlite3.connect(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
rows = rows[len(rows) * 2 // 3:]
t0 = rows[0][1]
for name, a, b, gx, gy, gz in rows:
    if (b - a) / 1e3 >= mn: print("%10.1f us  +%9.1f us  grid %6d %4d %3d  %s" % ((a - t0) / 1e3, (b - a) / 1e3, gx, gy, gz, name[:60]))
PY
rm -rf $D
