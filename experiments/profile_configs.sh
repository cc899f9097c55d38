#!/bin/bash
# Per-config rocprofv3 kernel stats under profiles/: profile_configs.sh rNN [cfg ...]      (GPU box only; ~4 min for all)
#   profiles/rNN_<cfg>_kernel_stats.csv  = rocprofv3 --kernel-trace --stats of experiments/one_config.py <cfg> (pandrs:: rows + totals)
#   profiles/rNN_configs.log             = the same runs' own hipEvent phase times
tag=$1; shift
cfgs=${@:-north_star c3 c4_shard c4_one_gpu c5_shard c5_one_gpu join_indices}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
: > $root/profiles/${tag}_configs.log
for cfg in $cfgs; do
    rm -rf $out/$cfg
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$cfg -- python3 $root/experiments/one_config.py $cfg > $out/$cfg.json 2> $out/$cfg.err
    tail -1 $out/$cfg.json >> $root/profiles/${tag}_configs.log
    f=$(find $out/$cfg -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then
        python3 - "$f" > $root/profiles/${tag}_${cfg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    if "pandrs::" in r["Name"]:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
    fi
    echo "$cfg done: $(tail -1 $out/$cfg.json | cut -c1-200)"
    mkdir -p $root/gpurun_out/profiles_$tag; cp $root/profiles/${tag}_* $root/gpurun_out/profiles_$tag/ 2>/dev/null
done
