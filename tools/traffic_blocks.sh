#!/bin/bash
# Run ON THE GPU BOX (through gpurun): HBM bytes per record loop (block cut, the three tiers, likelihoods) from rocprofv3 PMC passes over
# `bench.py --workload <c4|c5>` -- FETCH_SIZE and WRITE_SIZE in passes of their own, never combined with a trace domain.
# usage: tools/traffic_blocks.sh <tag> <workload> [bench args...]   -> gpurun_out/traffic_<tag>/{traffic_blocks_<workload>.json, summary.txt}
set -o pipefail
TAG=${1:-r04}; shift
WL=${1:-c5}; shift
OUT=$PWD/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --sustained-s 0 $*"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log" || { echo "FETCH_SIZE pass failed"; tail -3 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $BENCH > /dev/null 2> "$OUT/write.log" || { echo "WRITE_SIZE pass failed"; tail -3 "$OUT/write.log"; exit 1; }
cd - > /dev/null
python3 tools/traffic_blocks.py "$OUT" "$WL" | tee "$OUT/summary.txt"
rm -rf "$OUT/fetch" "$OUT/write"
