# usage: bash tools/probe/trace_cmd.sh <top> <python script and arguments ...>: kernel trace of one command, per-kernel totals printed, database dropped
cd /tmp && export TMPDIR=/tmp
TOP=$1; shift
D=$GRAFT_REPO_ROOT/gpurun_out/prof_tmp
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o t -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_cmd.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/dbstats.py $(find $D -name '*.db' | head -1) $TOP
rm -rf $D
