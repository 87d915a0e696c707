#!/bin/bash
# default vs SF_FUSED=2 on the other shapes, interleaved on one box
C="--no-extras --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 300 python bench.py $C "$@" > gpurun_out/r03m/shape_$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/r03m/shape_$name.log; return 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r03m/shape_$name.log").read().strip().split("\n")[-1])
print("%-14s %6.2f M  %.3f ms" % ("$name", d["value"]/1e6, d["ms_per_step"]), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items() if x}, d["check"]["decisions_matching_ground_truth"], d["check"]["of"])
PY
}
mkdir -p gpurun_out/r03m
for v in fused split; do
  if [ $v = split ]; then export SF_FUSED=2; else unset SF_FUSED; fi
  run cfg3_$v --keyframes 20000 --features 1000 --iterations 2000 --steps 20 --warmup 2 &&
  run b512_$v --desc-bytes 64 --steps 100 --warmup 10 &&
  run k40_$v --keyframes 40000 --steps 20 --warmup 2 &&
  run k125_$v --keyframes 125000 --steps 6 --warmup 1 &&
  run pnp_$v --estimator pnp --steps 100 --warmup 10
done
