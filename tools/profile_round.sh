#!/bin/bash
# One round of rocprofv3 evidence for `python bench.py` (run on the GPU box; outputs under gpurun_out/prof_<tag>):
# kernel stats, FETCH_SIZE / WRITE_SIZE passes and an SQ pass (each --pmc set in its own run).
# Usage: tools/profile_round.sh <tag> [bench args...]
set -e
tag=$1; shift
# one rocprofv3 invocation profiles ONE process: with --gpus N > 1 bench.py would start its ranks from a parent whose
# GPU the profiler's preloaded library has already initialised -- the program replacement this pool forbids
for a in "$@"; do
  if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "profile_round.sh: --gpus $a: profile one rank per rocprofv3 invocation" >&2; exit 2; fi
  prev=$a
done
out=gpurun_out/prof_$tag
mkdir -p $out
B="python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline $*"
# the stats pass times enough launches that its averages can be held against bench.py's own (HIP-event) figures
BS="python bench.py --steps 60 --warmup 20 --no-extras --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $BS > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $B > $out/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sqw -- $B > $out/sqw.log 2>&1
python tools/summarize_prof.py $tag $out/stats $out/fetch $out/write > $out/summary.txt
python tools/pmc_kernel.py $out/sq > $out/sq.txt
# (the verification's kernels: the fused form, or -- inside overlapped steps, the default since round 3's last session --
#  the split form's matching and chain kernels; a kernel that did not run leaves no file)
for kern in k_verify_fused k_match_split "k_chain<" k_chain_pnp "k_ba_pass<1, true, true>" "k_ba_pass<1, false, true>"; do python tools/sq_json.py $tag "$kern" $out/sq $out/sqw; done
cp profiles/${tag}_summary.json profiles/${tag}_kernel_stats.csv profiles/${tag}_sq_*.json $out/
grep "^{" $out/stats.log > $out/${tag}_bench_stdout.log
tail -30 $out/summary.txt; grep -E "k_verify_fused|k_match_split|k_chain|k_match_global_mf|k_ba_pass" $out/sq.txt
