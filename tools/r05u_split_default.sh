#!/bin/bash
# Round-5 experiment u: the split form inside overlapped steps (SF_STEP_SPLIT=1) against the fused kernel, now that the
# split form's matcher is pipelined; 200-step lines and the driver's 20-step command, alternating on one box.
set -o pipefail
out=gpurun_out/r05u; mkdir -p $out
for rep in 1 2 3; do
  for e in SF_STEP_SPLIT=0 SF_STEP_SPLIT=1; do
    for args in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
      echo "== $e $args rep $rep" | tee -a $out/summary.txt
      env $e timeout -k 10 300 python bench.py $args --no-extras --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
      python -c "
import json
d = json.loads(open('$out/b.json').read().strip().splitlines()[-1])
print('value %.3f M  ms_per_step %.4f  form %s' % (d['value'] / 1e6, d['ms_per_step'], d.get('verification_form')))" | tee -a $out/summary.txt
    done
  done
done
