#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_gru.py -m gpu -q -x 2>&1 | tail -2
REF="--hidden 128 --layers 4 --latent 128 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise"
python3 bench.py $REF > $O/bench_hidden128layers4latent128.json 2>> $O/bench.err
OS_GRU_STAGE=0 python3 bench.py $REF > $O/bench_hidden128layers4latent128_nostage.json 2>> $O/bench.err
python3 bench.py --hidden 128 --layers 4 --steps 3 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_hidden128layers4.json 2>> $O/bench.err
python3 bench.py --hidden 64 --layers 4 --steps 5 --warmup 1 --cpu-seconds 0 --parity-samples 2048 --no-second-noise > $O/bench_hidden64layers4.json 2>> $O/bench.err
for f in bench_hidden128layers4latent128 bench_hidden128layers4latent128_nostage bench_hidden128layers4 bench_hidden64layers4; do python3 - $O/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), {n: (round(v["ms_per_launch"], 4), v["kernel"]) for n, v in d["kernels"].items()}, d["parity"]["ok"])
PY
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_ref_shape2 --output-format csv -- python3 $R/bench.py $REF > /dev/null 2> $O/prof_ref2.log)
python3 - <<PY > $O/ref_shape_kernel_stats.md
import csv, glob
f = glob.glob("$O/prof_ref_shape2/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py $REF\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
head -6 $O/ref_shape_kernel_stats.md | cut -c1-150
