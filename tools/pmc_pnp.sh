# SQ counters of the PnP estimator's kernels in `bench.py --estimator pnp` (k_match_split + k_chain_pnp), full chain and
# truncated after its phases (SF_RANSAC_STOP: 1 gather, 2 + RANSAC loop, 3 + mask, 4 + refinement rounds); GPU box only
mkdir -p gpurun_out/pmc_pnp; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 0 1 2 3 4; do
  SF_RANSAC_STOP=$s timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_pnp/stop$s -- python3 bench.py --estimator pnp --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/pmc_pnp/stop$s.log 2>&1
  echo "== stop $s"; python tools/pmc_kernel.py gpurun_out/pmc_pnp/stop$s k_chain_pnp | grep -E "INSTS_VALU|INSTS_LDS|GRBM|WAIT_ANY|WAVE_CYCLES"
done
python tools/pmc_kernel.py gpurun_out/pmc_pnp/stop0 k_match_split | grep -E "INSTS_VALU|GRBM"
