#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box; summaries land in
# gpurun_out/prof/ and are condensed into profiles/ by tools/condense_profiles.py afterwards.
# Kernel-trace and counter passes are separate runs, one counter group per run.
set -o pipefail
R="$GRAFT_REPO_ROOT"
OUT="$R/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/bench" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-dense --steps 20 --warmup 3 > "$OUT/bench.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python3 "$R/tools/run_loglik.py" 512 150 3 > "$OUT/fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write -- python3 "$R/tools/run_loglik.py" 512 150 3 > "$OUT/write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/corner_sq" -o sq -- python3 "$R/tools/run_corner.py" 64 3 > "$OUT/corner_sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/corner_fetch" -o fetch -- python3 "$R/tools/run_corner.py" 64 3 > "$OUT/corner_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/corner_write" -o write -- python3 "$R/tools/run_corner.py" 64 3 > "$OUT/corner_write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fast_fetch" -o fetch -- python3 "$R/tools/run_fast9.py" 64 3 > "$OUT/fast_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/fast_write" -o write -- python3 "$R/tools/run_fast9.py" 64 3 > "$OUT/fast_write.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/qr" -o qr -- python3 "$R/tools/qr_bench.py" 6012 6012 2 > "$OUT/qr.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA -d "$OUT/qr_mfma" -o mfma -- python3 "$R/tools/qr_bench.py" 6012 6012 1 > "$OUT/qr_mfma.log" 2>&1 || echo "mfma counter pass failed (counter names may differ)" > "$OUT/qr_mfma.failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/bank_affine" -o bank_affine -- python3 "$R/tools/run_bank_affine.py" 512 150 3 > "$OUT/bank_affine.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/bank_iekf" -o bank_iekf -- python3 "$R/tools/run_iekf_bank.py" 150 512 --batched > "$OUT/bank_iekf.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/c4" -o c4 -- python3 "$R/tools/run_update.py" 2000 2 > "$OUT/c4.log" 2>&1 || echo "config-4 trace failed" > "$OUT/c4.failed"
TAG="${1:-r03}"
# front-end kernel trace and the SQ counter passes of the kernels the front end and the headline call spend their time in
bash "$R/tools/trace_pass.sh" orb run_orb.py 64 3 > "$OUT/orb_trace.log" 2>&1 || echo "orb trace failed" > "$OUT/orb_trace.failed"
bash "$R/tools/sq_pass.sh" loglik run_loglik.py 512 150 3 > "$OUT/sq_loglik.log" 2>&1 || echo "loglik sq pass failed" > "$OUT/sq_loglik.failed"
bash "$R/tools/sq_pass.sh" fast9 run_fast9.py 64 3 > "$OUT/sq_fast9.log" 2>&1 || echo "fast9 sq pass failed" > "$OUT/sq_fast9.failed"
bash "$R/tools/sq_pass.sh" orb run_orb.py 64 3 > "$OUT/sq_orb.log" 2>&1 || echo "orb sq pass failed" > "$OUT/sq_orb.failed"
cd /tmp
mkdir -p "$R/gpurun_out/profiles_out"
for f in loglik fast9 orb; do [ -f "$R/gpurun_out/sq/$f.json" ] && cp "$R/gpurun_out/sq/$f.json" "$R/gpurun_out/profiles_out/${TAG}_${f}_sq_counters.json"; done
[ -f "$R/gpurun_out/trace/orb_kernel_stats.csv" ] && cp "$R/gpurun_out/trace/orb_kernel_stats.csv" "$R/gpurun_out/profiles_out/${TAG}_orb_kernel_stats.csv"
# the databases are too large to travel back: condensed here, summaries only are kept
VSLAM_PROFILES_DST="$R/gpurun_out/profiles_out" python3 "$R/tools/condense_profiles.py" "$TAG" || exit 1
rm -rf "$OUT"/*/
ls "$R/gpurun_out/profiles_out"
