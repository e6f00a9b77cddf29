#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats over any python command of this repository.
# usage: tools/prof_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv, out.txt, err.txt} and a table on stdout
set -o pipefail
TAG=${1:-x}; shift
SCRIPT=$PWD/$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $SCRIPT "$@" > "$OUT/out.txt" 2> "$OUT/err.txt" || { echo "stats pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
cd - > /dev/null
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
print("%-70s %7s %12s %12s %6s" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
for r in rows[:24]:
    n = r["Name"]
    n = n.replace("(anonymous namespace)::", "").split("(")[0][:70]
    print("%-70s %7s %12.1f %12.3f %6s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r.get("Percentage", "")))
PY
