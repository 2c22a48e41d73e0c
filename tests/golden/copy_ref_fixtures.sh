#!/bin/bash
# Regenerates tests/golden/ref_fixtures/ from a ChASE checkout (data files of the reference's own tests; fp64 types only).
# usage: tests/golden/copy_ref_fixtures.sh /path/to/ChASE
set -e
REF=${1:-/root/reference}
OUT=$(dirname "$0")/ref_fixtures
mkdir -p "$OUT"
for t in double cdouble; do
  for c in 10 1e4 ill; do cp "$REF/tests/linalg/internal/QR_matrices/matrix_${t}_cond_${c}.bin" "$OUT/"; done
done
for f in cdouble_random_BSE cdouble_tiny_random_BSE; do
  cp "$REF/tests/linalg/internal/BSE_matrices/${f}.bin" "$REF/tests/linalg/internal/BSE_matrices/eigs_${f}.bin" \
     "$REF/tests/linalg/internal/BSE_matrices/SH_eigs_${f}.bin" "$OUT/"
done
for f in double_random_BSE double_tiny_random_BSE; do      # real pseudo-Hermitian fixtures [[A, B], [-B, -A]]
  cp "$REF/tests/linalg/internal/BSE_matrices/${f}.bin" "$REF/tests/linalg/internal/BSE_matrices/eigs_${f}.bin" "$OUT/"
done
