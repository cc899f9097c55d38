#!/bin/bash
# profiles/rNN_* of the library as it stands: refresh_profiles.sh rNN     (GPU box only; ~5 min)
#   1. kernel-trace stats of the bench's timed region          -> gpurun_out/prof_rNN/stats
#   2. two --pmc passes (FETCH_SIZE / WRITE_SIZE, kernel trace only) -> pmc_traffic.json
#   3. the bench line itself (reads the traffic file just made)
set -e
tag=$1; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
B="$root/bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B > $out/bench_under_rocprof.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $B > $out/write.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $root/profiles/${tag}_bench_kernel_stats.csv
cp $(find $out/fetch -name "*counter_collection.csv" | head -1) $root/profiles/${tag}_pmc_fetch_counter_collection.csv
cp $(find $out/write -name "*counter_collection.csv" | head -1) $root/profiles/${tag}_pmc_write_counter_collection.csv
tail -1 $out/bench_under_rocprof.json > $root/profiles/${tag}_bench_under_rocprof.json
cd $root
python3 experiments/pmc_traffic.py profiles/${tag}_pmc_fetch_counter_collection.csv profiles/${tag}_pmc_write_counter_collection.csv 12 100000000 1000000 4 > profiles/${tag}_pmc_traffic.json
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json > profiles/${tag}_bench.json
mkdir -p gpurun_out/profiles_$tag; cp profiles/${tag}_* gpurun_out/profiles_$tag/
python3 - <<PY
import json
d = json.load(open("profiles/${tag}_bench.json"))
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "dominant", d["roofline"].get("dominant_kernel", {}).get("frac"))
for k in ("north_star_sum", "c2_skew_80_20", "c2_sorted", "c3", "c4_shard", "c5_shard", "c5_one_gpu", "c1"): print(k, round(d[k]["ms"], 4))
PY
