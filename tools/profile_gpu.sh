#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats and PMC passes over
# bench.py, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --pmc is never combined with the trace domains gpurun refuses).
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py --cpu-sample 0 --steps 3 --warmup 1 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $BENCH > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.log" || { echo "stats pass failed"; tail -5 "$OUT/bench_stats.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 $BENCH > /dev/null 2> "$OUT/pmc_$name.log" || { echo "pmc pass $grp failed"; tail -3 "$OUT/pmc_$name.log"; }
done
# calibration of FETCH_SIZE on this kernel's own access pattern: with --scan-ablate 3 the filter kernel issues only
# its table stream (12 B per row with the compact layout, 16 B with the SoA one: known exactly), no gate loads, no list writes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_FETCH_SIZE" -- python3 $BENCH --scan-ablate 3 > /dev/null 2> "$OUT/cal_FETCH_SIZE.log" || echo "calibration pass failed"
cd - > /dev/null
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
