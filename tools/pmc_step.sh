#!/bin/bash
# Per-kernel PMC evidence of the bench workload (run via gpurun): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections as
# MI355X_MICROARCH.md prescribes) and the SQ counters behind "MFMA utilisation vs gfx950 peak" (VERDICT r3 item 4), every pass with
# --kernel-trace only and the program directly behind `--`.
#   tools/pmc_step.sh TAG [bench args...]   ->  gpurun_out/pmc_step_TAG/{counters_by_kernel.json, summary.txt}
tag=${1:-fp32}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_step_$tag; rm -rf $out; mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --mode eager "$@" > $out/p$i.log 2>&1
done
python3 tools/pmc_step_post.py $out "$tag" "$*"
rm -rf $out/p[0-9]
