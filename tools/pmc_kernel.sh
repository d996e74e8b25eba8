#!/bin/bash
# Counter-only rocprofv3 passes (no tracing options) over any command; per counter, mean and sum over the dispatches of the
# kernels whose name contains MATCH.  Usage: tools/pmc_kernel.sh OUTDIR MATCH -- <command ...>
# Writes gpurun_out/OUTDIR/summary.json: {counter: {n, mean, sum}} + derived figures (MFMA busy fraction, effective clock,
# LDS bank-conflict fraction, FETCH/WRITE bytes per dispatch with the gfx950 x2 correction on FETCH_SIZE).
out=$1; match=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for pmc in "${passes[@]}"; do
  timeout 600 rocprofv3 --pmc $pmc -d $O/p$i -o out --output-format csv -- "$@" > $O/p$i.log 2>&1 || echo "pass $i ($pmc) failed: $(tail -2 $O/p$i.log | head -1)"
  i=$((i+1))
done
# durations of the same kernels (separate, tracing-only run)
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o out --output-format csv -- "$@" > $O/kt.log 2>&1
python $R/tools/pmc_summary.py "$O" "$match" "$*"
