#!/bin/bash
# usage (GPU box): bash scripts/pmc_nb.sh <tag> [bench args...]
# PMC counters of the batched nonbonded kernel in separate rocprofv3 passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit
# one pass; --pmc never together with other trace domains) -> gpurun_out/pmc_<tag>/pmc_nonbonded.json, to be copied to profiles/.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-single --no-kernel-timing --nsteps-nc 200 "$@" > $out/g$i.log 2>&1
done
python3 scripts/pmc_collect.py $out "$@"
rm -rf $out/g*/
