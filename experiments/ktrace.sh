#!/bin/bash
# kernel-trace stats of one experiment script: ktrace.sh OUTDIR script.py [args...]   (GPU box only)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -- python3 $GRAFT_REPO_ROOT/$@ > $GRAFT_REPO_ROOT/$out.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-100s calls %5s avg %10.1f us  total %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
