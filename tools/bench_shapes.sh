#!/bin/bash
C="--no-extras --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 400 python bench.py $C "$@" > gpurun_out/shape_$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/shape_$name.log; return 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/shape_$name.log").read().strip().split("\n")[-1])
print("$name", round(d["value"]), round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["kernel_ms_per_step"].items() if x}, d["check"]["decisions_matching_ground_truth"], d["check"]["of"])
PY
}
run cfg3 --keyframes 20000 --features 1000 --iterations 2000 --steps 20 --warmup 2 &&
run b512 --desc-bytes 64 &&
run cfg5 --desc-bytes 64 --netvlad-f16 &&
run k40 --keyframes 40000 --steps 30 --warmup 2 &&
run k125 --keyframes 125000 --steps 12 --warmup 1
