#!/bin/bash
# Copies the summaries of a tools/collect_profiles.sh run (gpurun_out/<tag>/: bench lines, kernel-stat tables, counter and timestamp
# summaries -- not the raw rocprofv3 directories) into profiles/ as <tag>_<name>.   usage: tools/publish_profiles.sh <tag>
TAG=${1:-r04}; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/$TAG
for f in $O/*.json $O/*_kernel_stats.md $O/*.txt $O/*_raw.md $O/shard_sweep.md; do
  [ -f "$f" ] || continue
  # an empty file or a Python traceback is a collection step that failed: say so, keep what profiles/ holds
  if [ ! -s "$f" ] || grep -q "^Traceback (most recent call last)" "$f"; then echo "NOT PUBLISHED (empty or a traceback): $f" >&2; continue; fi
  b=$(basename $f)
  case $b in
    traffic.json) cp $f $R/profiles/${TAG}_traffic_raw.json;;
    traffic_profiles.json) ;;                       # = profiles/traffic.json itself (written by tools/traffic_to_json.py)
    *) cp $f $R/profiles/${TAG}_$b;;
  esac
done
[ -s $O/traffic_profiles.json ] && cp $O/traffic_profiles.json $R/profiles/traffic.json
ls $R/profiles | grep -c "^${TAG}_"
