#!/bin/bash
# A step's kernels ONE AT A TIME (one lane, depth 1) under rocprofv3 --kernel-trace --stats: their alone durations, whose
# sum the overlapped step is held against.  usage: tools/alone_times.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SF_STEP_LANES=1 SF_STEP_DEPTH=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone -- python3 bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline "$@" > $out/alone_bench.json 2> $out/alone.err || { tail -5 $out/alone.err; exit 1; }
python tools/summarize_prof.py $tag $out/alone > /dev/null
python - $tag <<'PY'
import json, sys
d = json.load(open("profiles/%s_summary.json" % sys.argv[1]))
tot = 0.0
for k in d["kernels"][:24]:
    print("%-40s calls %5d  avg %10.1f us  median %10.1f  pct %5.1f" % (k["kernel"], k["calls"], k["avg_us"], k.get("median_us", 0), k["pct"]))
PY
tail -1 $out/alone_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one lane, depth 1: value %.3f M  ms_per_step %.4f' % (d['value']/1e6, d['ms_per_step']))"
cp profiles/${tag}_summary.json $out/
