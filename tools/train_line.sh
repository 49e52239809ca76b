#!/bin/bash
# one --mode train bench line, condensed: ms per step and the per-kernel table.  usage: bash tools/train_line.sh [bench args]
python3 bench.py --mode train --steps 20 --warmup 3 --cpu-seconds 0 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:(v['kernel'], round(v['ms_per_launch'],4), v['launches_per_step']) for k,v in d['kernels'].items() if 'train' in k or 'gru_layer' in k})"
