// impl_extras.hpp — what the C entry points, the bench and the tape (tape.hpp) need from an Impl beyond the ChaseBase surface.
// No HIP, no C ABI: includable by host-only test programs.
#pragma once
#include <cstddef>

namespace chase_amd {

// what the C entry points and the bench need beyond the ChaseBase surface, common to both Impls
struct HipImplExtras {
    virtual ~HipImplExtras() = default;
    virtual std::size_t locked() const = 0;
    virtual int last_qr_variant() const = 0;     // 0 = Householder, 1/2/3 = CholQR1 / CholQR2 / shifted CholQR2
    virtual double filter_ms() const = 0;        // HIP-event time between FilterPhaseStart/End, accumulated
    virtual std::size_t hemm_calls() const = 0;
    virtual std::size_t hemm_reused_vecs() const { return 0; }   // filter columns served from RR's cached H V (no GEMM)
    virtual std::size_t resd_rechecked() const { return 0; }     // residuals re-taken from a fresh four-product H v (on the tolerance)
    virtual void set_device_rng(bool) = 0;
    virtual void reset_counters() = 0;
    virtual void* device_V1() = 0;               // current (local) vector block, pending swaps applied
    virtual std::size_t local_rows() const = 0;
    // out[j] = || H v_j - lambda_j v_j ||_2 for the first ncols vectors the Impl holds, from a FRESH four-product H V
    // (never from products a previous step left behind): what the reference's solve tests recompute after a solve
    // (tests/chase_serial_solve.cpp:144-148,195-199, tests/chase_distributed_solve.cpp:209-284)
    virtual void recompute_residuals(std::size_t ncols, const double* lambda, double* out) = 0;
    // k >= 0: the next Resd re-takes the residuals of its first k columns from a fresh four-product H v whatever their
    // values (single-rank replay: as many as the recorded solve re-took on the tolerance, tape.hpp); -1: by value (default)
    virtual void set_forced_recheck(long) {}
    // v >= 0: the next QR takes variant v (0 Householder, 1/2/3 CholQR1 / CholQR2 / shifted CholQR2) whatever its data say -
    // a Cholesky factorisation that fails on the replayed rank's numbers is retried on a shifted Gram matrix instead of
    // falling back to Householder (single-rank replay follows the recording's QR variants, tape.hpp); -1: by data (default)
    virtual void set_forced_qr(int) {}
    virtual std::size_t forced_qr_retries() const { return 0; }
    // single-rank replay of the pseudo-Hermitian solve: the lone rank's projected matrices are partial sums - when Q^H S H Q does
    // not factorise, Rayleigh-Ritz runs the same operators on the identity instead of throwing; a Lanczos tridiagonal the host
    // eigensolver rejects yields zeros (the driver reads the tape's numbers either way)
    virtual void set_replay_tolerant(bool) {}
    virtual std::size_t replay_tolerated() const { return 0; }
    // phase marker of the profiler ranges (CHASE_HIP_ROCTX=1, roctx.hpp): nothing to implement
};

} // namespace chase_amd
