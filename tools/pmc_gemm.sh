#!/bin/bash
# PMC passes over one GEMM shape (counters only: no tracing options).  Usage: tools/pmc_gemm.sh OUTDIR M N K [env...]
# Prints, per counter, the mean over dispatches of the GEMM kernel.
out=$1; M=$2; N=$3; K=$4
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$out
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_SMEM"
)
i=0
for pmc in "${passes[@]}"; do
  d=$R/gpurun_out/$out/p$i
  timeout 300 rocprofv3 --pmc $pmc -d $d -o out --output-format csv -- python $R/tools/gemm_one.py $M $N $K 4 > $d.log 2>&1 || echo "pass $i failed: $(tail -2 $d.log | head -1)"
  i=$((i+1))
done
python - <<EOF
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
EOF
