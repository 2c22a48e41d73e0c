#!/bin/bash
# Round-5 verdict item 2: attribute the 3M-vs-4M gap of the filter kernel with PMC passes of BOTH instantiations on ONE lease.
# The full-width filter HEMM of config 4 (N = 65536 complex, 2560 columns), CHASE_HIP_GEMM3M=1 / 0; per counter group one
# rocprofv3 run; the kernel's duration is taken from the SAME run's kernel trace, so clock = GRBM_GUI_ACTIVE / 8 XCDs / duration.
# usage: scripts/r06_pmc_3m_vs_4m.sh <out.txt> [N] [ncols] [reps]
OUT=$(realpath -m "$1"); N=${2:-65536}; NC=${3:-2560}; REPS=${4:-2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
GROUPS_=(
 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
 "FETCH_SIZE"
)
for M3 in 1 0; do
  echo "######## CHASE_HIP_GEMM3M=$M3  ($([ $M3 = 1 ] && echo 'three real products per complex product' || echo 'four products, the reference arithmetic'))" >> "$OUT"
  export CHASE_HIP_GEMM3M=$M3
  python3 $REPO/scripts/dev_gemm_only.py z $N $NC 3 2>&1 | tail -1 >> "$OUT"      # un-profiled rate first
  i=0
  for g in "${GROUPS_[@]}"; do
    d=/tmp/pmc34_${M3}_$i; rm -rf $d
    rocprofv3 --kernel-trace --pmc $g -f csv -d $d -- python3 $REPO/scripts/dev_gemm_only.py z $N $NC $REPS > /tmp/pmc34_${M3}_$i.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    k=$(find $d -name "*kernel_trace.csv" | head -1)
    echo "## group: $g" >> "$OUT"
    python3 - "$f" "$k" >> "$OUT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_f64_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(sys.argv[2])) if "gemm_f64_kernel" in r["Kernel_Name"]]
ms = sum(dur) / len(dur)
print(f"kernel duration in this run: launches={len(dur)} mean={ms:.3f} ms")
for c, v in sorted(acc.items()):
    m = sum(v) / len(v)
    extra = ""
    if c == "GRBM_GUI_ACTIVE": extra = f"   -> clock = {m / 8 / (ms * 1e-3) / 1e9:.3f} GHz"
    print(f"{c:28s} launches={len(v):3d} mean={m:.6g}{extra}")
PY
    i=$((i+1))
  done
done
