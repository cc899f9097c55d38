#!/bin/bash
# rocprofv3 --pmc over a native binary: pmc_bin.sh OUTDIR BINARY "CTR ..." ["CTR ..."]    (GPU box only)
out=$1; shift; bin=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$out/pass$i -- $GRAFT_REPO_ROOT/$bin > $GRAFT_REPO_ROOT/$out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $GRAFT_REPO_ROOT/$out/pass$i.log; exit 1; }
done
