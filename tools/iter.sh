#!/bin/bash
# one development iteration on a GPU box: GPU tests, a short bench line, SQ counters of the two hot kernels
O=gpurun_out/${1:-iter}; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 280 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 30 ${@:2} > $O/b.json 2> $O/b.log || tail -3 $O/b.log
python3 -c "
import json; j=json.load(open('$O/b.json')); r=j['roofline']; c=j['counters_per_batch_rank0']
print('value %.3e ms/batch %.4f probe ms %.4f other ms %.4f frac %.3f' % (j['value'], j['config']['ms_per_batch'], r['dominant_kernel']['ms'], r['other_kernels_ms'], r['frac']))"
bash tools/pmc_sq.sh ${1:-iter}_pmc ${@:2} | grep -E "INSTS|WAVE_CYCLES|WAIT_ANY|ACTIVE_INST|CONFLICT"
