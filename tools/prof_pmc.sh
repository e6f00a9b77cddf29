#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 PMC passes (one per counter group, never combined with a trace domain)
# over one bench.py command; per-kernel means of every counter for the kernels whose name matches KERNELS.
# usage: KERNELS="scan_probe|scan_hits" tools/prof_pmc.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-x}; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py --cpu-sample 0 --sustained-s 0 $*"
# PMC_SCRIPT=tools/c5_blocks_bench.py: another python command of this repository instead of bench.py (its arguments follow the tag)
if [ -n "$PMC_SCRIPT" ]; then BENCH="$PWD/$PMC_SCRIPT $*"; fi
GROUPS_=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"
 "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS"
)
# PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" (groups separated by ';') replaces the default set
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -r -a GROUPS_ <<< "$PMC_GROUPS"; fi
cd /tmp
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 $BENCH > /dev/null 2> "$OUT/g$i.log" || { echo "pmc pass $i ($grp) failed"; tail -3 "$OUT/g$i.log"; }
done
cd - > /dev/null
python3 - "$OUT" "${KERNELS:-scan_probe|scan_hits}" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, pat = sys.argv[1], re.compile(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if pat.search(n):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(acc):
        print(k, file=fh)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-50s n=%-5d mean=%.4g  max=%.4g" % (c, len(v), sum(v) / len(v), max(v)), file=fh)
print(open(os.path.join(out, "summary.txt")).read())
PY
find "$OUT" -name '*.csv' -size +2M -delete
