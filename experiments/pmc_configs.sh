#!/bin/bash
# HBM traffic of ONE call per benchmark configuration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only,
# separate runs, as guides/MI355X_MICROARCH.md prescribes):  pmc_configs.sh rNN [cfg ...]        (GPU box only)
#   profiles/rNN_pmc_<cfg>.json = per-kernel bytes per launch (FETCH_SIZE x 2: the gfx950 correction), bytes per call of the
#   whole pipeline, the configuration's algorithmic bytes and their ratio.
tag=$1; shift
cfgs=${@:-north_star c3 c5_one_gpu}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for cfg in $cfgs; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf $out/${cfg}_$ctr
        timeout -k 10 280 rocprofv3 --pmc $ctr --output-format csv -d $out/${cfg}_$ctr -- python3 $root/experiments/one_config.py $cfg 3 > $out/${cfg}_$ctr.json 2> $out/${cfg}_$ctr.err || { echo "$cfg $ctr failed"; tail -3 $out/${cfg}_$ctr.err; }
    done
    f=$(find $out/${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    python3 $root/experiments/pmc_config_traffic.py $cfg "$f" "$w" 4 > $root/profiles/${tag}_pmc_${cfg}.json; mkdir -p $root/gpurun_out/profiles_$tag; cp $root/profiles/${tag}_pmc_${cfg}.json $root/gpurun_out/profiles_$tag/ && echo "$cfg: $(python3 -c "import json;d=json.load(open('$root/profiles/${tag}_pmc_${cfg}.json'));print('%.2f GB per call, %.2f x algorithmic' % (d['pipeline_hbm_bytes_per_call']/1e9, d['traffic_over_algorithmic']))")"
done
