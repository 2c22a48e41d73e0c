#!/bin/bash
# Regenerates tests/golden/driver_trace_*.txt: runs of the REFERENCE's own driver (chase::Solve, compiled as it stands from
# /root/reference/algorithm/algorithm.hpp — header-only, BLAS-free, no stand-ins) on the naive CPU kernel of
# tests/cpu_mock_kernel.hpp deriving from the reference's chase::ChaseBase<double>.  Build container only.
# Each file: iteration count, filtered-vector count, eigenpairs, and every virtual call with its scalar arguments.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
EXE=$(mktemp -d)/ref_driver_trace
g++ -std=c++17 -O2 -I"$REF" -o "$EXE" "$HERE/ref_driver_trace.cpp"
run() {   # name N nev nex deg opt perturb [seq]
    local name=$1; shift
    { echo "# reference driver run: ref_driver_trace $* (N nev nex deg opt perturb [seq]); Clement-type matrix of tests/chase_serial_solve.cpp:52-90"
      "$EXE" "$@"; } > "$HERE/driver_trace_$name.txt"
    echo "driver_trace_$name.txt: $(sed -n 2,3p "$HERE/driver_trace_$name.txt" | tr '\n' ' ')"
}
run clement256      256 24 16 16 1 1e-6     # the configuration of tests/chase_serial_solve.cpp (deg 16, opt on)
run clement256_fix  256 24 16 20 0 0        # no degree optimisation, unperturbed (analytic spectrum)
run clement512      512 50 14 10 1 1e-6     # few extra vectors, low degree: many iterations, many swaps
run clement1001    1001 60 40 20 1 1e-6
run clement1200    1200 80 60 20 1 1e-6     # shape of tests/noinput.cpp problem #0
run clement256_seq  256 24 16 16 1 1e-6 1   # two problems: random start, then the perturbed matrix in approximate mode ('A')

# function-level known answers: the reference's calc_degrees / locking / filter / lanczos (+ the pseudo-Hermitian routines) on
# the seeded scenarios of tests/driver_function_scenarios.hpp
EXE2=$(dirname "$EXE")/ref_driver_functions
g++ -std=c++17 -O2 -I"$REF" -o "$EXE2" "$HERE/ref_driver_functions.cpp"
{ echo "# reference driver routines (chase::Algorithm<double>) on tests/driver_function_scenarios.hpp: S scenario, I inputs, O outputs, C kernel calls"
  "$EXE2"; } > "$HERE/driver_functions.txt"
echo "driver_functions.txt: $(grep -c '^S' "$HERE/driver_functions.txt") scenarios"
