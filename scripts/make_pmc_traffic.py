"""profiles/pmc_traffic.json from the text prof_pmc.sh writes (one counter group per rocprofv3 pass over
scripts/dev_gemm_only.py z 65536 2560 2): fabric bytes of ONE full-width launch of the whole-tile three-multiplication filter
kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE is reported in KiB and at half
the bytes of 16-byte-per-lane reads: x 1024 x 2; WRITE_SIZE in KiB as it stands), plus the matrix-pipe figures of the same
launch.  usage: make_pmc_traffic.py <pmc txt> <out json> [source label]"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
vals = {}
for m in re.finditer(r"^(\S+)\s+launches=\s*(\d+) mean=(\S+)\s+kernel=(.*)$", txt, re.M):
    name, n, mean, kernel = m.group(1), int(m.group(2)), float(m.group(3)), m.group(4)
    if "gemm_f64_kernel" in kernel and "Lb1ELb0ELi1ELb0ELb1" in kernel.replace(" ", "") or "gemm_f64_kernel<true, false, 1, false, true" in kernel:
        vals[name] = mean
if not vals:                                     # kernel names are truncated to 60 characters: take the dominant gemm rows
    for m in re.finditer(r"^(\S+)\s+launches=\s*(\d+) mean=(\S+)\s+kernel=(.*gemm_f64_kernel.*)$", txt, re.M):
        vals.setdefault(m.group(1), float(m.group(3)))
N, n, F = 65536, 2560, 16
alg = F * (N * N + 2 * (N + N) * n)               # s [(N/r)(N/c) + ((N/r) + (N/c)) ncols (1 + [beta != 0])]
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
rec = {
    "workload": "cfg4",
    "kernel": "gemm_f64_kernel<true,false,1,false,true> (filter HEMM, whole-tile 3M instantiation), full-width launch N=65536 complex, ncols=2560, beta != 0",
    "source": sys.argv[3] if len(sys.argv) > 3 else sys.argv[1],
    "how": "scripts/prof_pmc.sh: rocprofv3 --kernel-trace --pmc <one group per run> over scripts/dev_gemm_only.py z 65536 2560 2; "
           "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane reads), WRITE_SIZE as reported",
    "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "algorithmic_bytes_per_launch": alg,
    "note": "memory-side (fabric) bytes INCLUDING Infinity-Cache hits (the counter sits on the L2's fabric side and cannot separate "
            "HBM reads); it moves by +-40 % from run to run with how closely the workgroups that share an H panel stay in step, "
            "and is not the bound: see mfma_busy_fraction",
}
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    rec["l2_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    rec["mfma_busy_cycles"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"]
    rec["grbm_gui_active"] = vals["GRBM_GUI_ACTIVE"]
    rec["mfma_busy_fraction"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0)
if "SQ_INSTS_MFMA" in vals:
    rec["mfma_insts"] = vals["SQ_INSTS_MFMA"]
json.dump(rec, open(sys.argv[2], "w"), indent=1)
print(json.dumps(rec, indent=1))
