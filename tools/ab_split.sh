# A/B of verification-pipeline variants on the bench step (diagnostic):  bash tools/ab_split.sh <tag> NAME=ENV[,ENV...] ...
tag=$1; shift
mkdir -p gpurun_out/$tag
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $(echo "$envs" | tr ',' ' ') timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/$tag/$name.log 2> gpurun_out/$tag/$name.err
  python - "$tag" "$name" <<'PY'
import json,sys
tag,name=sys.argv[1:3]
for line in open('gpurun_out/%s/%s.log'%(tag,name)):
    if line.startswith('{'):
        j=json.loads(line); k=j['kernel_ms_per_step']
        print("%-14s value %.2f M  ms/step %.4f  fused/chain %.4f  match %.4f  filter %.4f  ok %s" % (name, j['value']/1e6, j['ms_per_step'], k.get('k_verify_fused',k.get('k_chain',0)), k.get('k_match_global',k.get('k_match_split',0)), k.get('k_nn_filter_f16',0), j['check']['decisions_matching_ground_truth']))
PY
done
