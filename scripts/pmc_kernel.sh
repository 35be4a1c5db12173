#!/bin/bash
# Counter passes for ONE kernel (dev aid, run on the GPU box from the repo root):
#   scripts/pmc_kernel.sh <tag> <kernel-name-substring> -- <python script and args>
# rocprofv3 --pmc in separate passes (kernel-trace only, as MI355X_MICROARCH.md prescribes), per-kernel
# averages printed and kept under gpurun_out/pmc_<tag>/summary.txt
set -e
export TMPDIR=/tmp
tag=$1; pat=$2; shift 3
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o t -- python3 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed (see $out/p$i.log)"
done
python3 - "$out" "$pat" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
