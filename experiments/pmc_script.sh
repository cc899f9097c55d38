#!/bin/bash
# one rocprofv3 --pmc pass per counter group over any experiment script:
#   pmc_script.sh OUTDIR "script.py args" "CTR1 CTR2 ..." "CTR3 ..."      (GPU box only)
out=$1; shift
script=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$out/pass$i -- python3 $GRAFT_REPO_ROOT/experiments/$script > $GRAFT_REPO_ROOT/$out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $GRAFT_REPO_ROOT/$out/pass$i.log; exit 1; }
done
