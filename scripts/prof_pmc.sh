#!/bin/bash
# PMC passes over the single-kernel HEMM driver (one counter group per rocprofv3 run; no trace domains besides kernel-trace).
# usage: [PMC_DRIVER=scripts/dev_panel_only.py] scripts/prof_pmc.sh <out.txt> <driver args...>   (default driver: dev_gemm_only.py)
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"
 "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
# PMC_GROUPS="A B;C" overrides the default groups (semicolon separated)
if [ -n "${PMC_GROUPS:-}" ]; then IFS=';' read -r -a GROUPS_ <<< "$PMC_GROUPS"; fi
i=0
for g in "${GROUPS_[@]}"; do
  d=/tmp/pmc_$i; rm -rf $d
  rocprofv3 --kernel-trace --pmc $g -f csv -d $d -- python3 $REPO/${PMC_DRIVER:-scripts/dev_gemm_only.py} "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "## group: $g" >> "$OUT"
  python3 - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "gemm_f64_kernel" in r["Kernel_Name"] or "tail_reduce" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{c:28s} launches={len(v):3d} mean={sum(v)/len(v):.6g}  kernel={k}")
PY
  i=$((i+1))
done
tail -3 /tmp/pmc_0.log >> "$OUT"
