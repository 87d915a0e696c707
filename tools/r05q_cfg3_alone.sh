#!/bin/bash
# Round-5 experiment q: the cfg3 step's kernels ONE AT A TIME (one lane, depth 1) under rocprofv3 --kernel-trace --stats:
# their alone durations, whose sum the overlapped step is held against.
set -o pipefail
out=gpurun_out/r05q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SF_STEP_LANES=1 SF_STEP_DEPTH=1 SF_FUSED=${SF_FUSED_FORM:-1} timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone${SF_FUSED_FORM:+_split} -- python3 bench.py --workload cfg3 --steps 12 --warmup 3 --no-extras --no-cpu-baseline > $out/alone_bench.json 2> $out/alone.err || { tail -5 $out/alone.err; exit 1; }
python tools/summarize_prof.py r05q_cfg3_alone${SF_FUSED_FORM:+_split} $out/alone${SF_FUSED_FORM:+_split} > /dev/null
python - <<'PY'
import json
d = json.load(open("profiles/r05q_cfg3_alone${SF_FUSED_FORM:+_split}_summary.json"))
for k in d["kernels"][:14]:
    print("%-40s calls %5d  avg %10.1f us  median %10.1f  pct %5.1f" % (k["kernel"], k["calls"], k["avg_us"], k.get("median_us", 0), k["pct"]))
PY
tail -1 $out/alone_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3f M  ms_per_step %.3f' % (d['value']/1e6, d['ms_per_step']))"
cp profiles/r05q_cfg3_alone${SF_FUSED_FORM:+_split}_summary.json $out/
