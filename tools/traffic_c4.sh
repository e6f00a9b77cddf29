#!/bin/bash
# Run ON THE GPU BOX (through gpurun): HBM bytes per launch group of the whole-genome scan's kernels (sub-slice form), from rocprofv3
# PMC passes over `bench.py --workload c4` -- FETCH_SIZE and WRITE_SIZE in passes of their own (they do not fit one, and --pmc is never
# combined with a trace domain), plus a calibration pass: with --scan-ablate 256 pass one issues its row stream (12 B per row, known
# exactly) and no ticket stores, which gives the factor FETCH_SIZE needs on this kernel's 8-byte loads (MI355X_MICROARCH.md, HBM:
# FETCH_SIZE counts 64 B per request on gfx950; calibrate on a known byte count in the kernel's own access pattern).
# usage: tools/traffic_c4.sh <tag> [bench args...]   -> gpurun_out/traffic_<tag>/{traffic_scan_c4.json, summary.txt}
set -o pipefail
TAG=${1:-r04}; shift
OUT=$PWD/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py --workload c4 --steps 1 --warmup 0 --cpu-sample 0 --sustained-s 0 $*"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log" || { echo "FETCH_SIZE pass failed"; tail -3 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $BENCH > /dev/null 2> "$OUT/write.log" || { echo "WRITE_SIZE pass failed"; tail -3 "$OUT/write.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal" -- python3 $BENCH --scan-ablate 256 > /dev/null 2> "$OUT/cal.log" || echo "calibration pass failed"
cd - > /dev/null
python3 tools/traffic_c4.py "$OUT" | tee "$OUT/summary.txt"
python3 tools/traffic_blocks.py "$OUT" c4 | tee -a "$OUT/summary.txt"      # the same passes hold the record loop's kernels
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/cal"   # (hundreds of MB of per-dispatch rows: gpurun copies back at most 64 MiB)
