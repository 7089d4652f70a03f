cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_qr
timeout -k 10 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_qr -o qr -- python3 $GRAFT_REPO_ROOT/tools/qr_bench.py 6012 6012 2 > $GRAFT_REPO_ROOT/gpurun_out/qr.log 2>&1
python3 - <<'PY'
import sqlite3, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_qr/**/*.db', recursive=True)[0]
db = sqlite3.connect(f)
rows = [r for r in db.execute("select name, start, end from kernels order by start") if 'vslam' in r[0]]
n = len(rows) // 3
last = rows[-n:]
t0, t1 = last[0][1], max(r[2] for r in last)
print("span %.3f ms" % ((t1 - t0) / 1e6))
tot = {}
for name, a, b in last:
    k = name.split('(')[0][-40:]
    c = tot.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += (b - a) / 1e3
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]: print("%-42s %4d %9.1f us avg %7.1f" % (k, c, us, us / c))
sel = [r for r in last if 'qr_block_kernel' in r[0]]
print([round((b - a) / 1e3) for _, a, b in sel])
print([round((y[1] - x[2]) / 1e3) for x, y in zip(sel, sel[1:])])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_qr
