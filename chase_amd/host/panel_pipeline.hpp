// panel_pipeline.hpp — the ordering logic of the panel-pipelined distributed HEMM, separated from what it orders.
//
// One direction of the distributed product (linalg/internal/mpi/hemm.hpp:114-229: local GEMM, then an all-reduce inside the
// column or row group) on columns [c0, c0 + nc), cut on a FIXED panel grid so that the all-reduce of panel p can run on its
// group's communication stream while the GEMM of panel p + 1 runs on the compute stream, and so that the NEXT product - the
// other direction, whose input is this one's output - can start on panel p as soon as panel p's all-reduce has landed (the
// reference left this overlap as commented-out code, linalg/internal/nccl/hemm.hpp:97-288).
//
// Ops supplies the four actions; pChaseHip::hemm_ptr binds them to the C ABI (chase_hip_grid_event_wait, the MFMA GEMM,
// chase_hip_grid_allreduce, chase_hip_grid_event_record_on), tests/pipeline_harness.cpp binds them to a simulator of streams
// and events that checks every pair of conflicting accesses for a happens-before edge - on the CPU, for every combination of
// active / inactive groups and one / two communication streams (the 2 x 1 grid's race of round 4 is one of them).
//   pipe    the product is cut on the panel grid and ordered by the per-panel events: whenever EITHER direction's collective
//           is asynchronous - my input panels are the output of the previous product of the OTHER direction, whose
//           all-reduces may still be in flight, also when my own group has a single member;
//   active  THIS direction has a collective (its group has more than one member).  An inactive direction records nothing:
//           its output panel is ordered by the compute stream itself, and the slot keeps the other direction's last event,
//           which this product has already waited for.
#pragma once
#include <algorithm>
#include <cstddef>

namespace chase_amd {

template <class Ops>
inline void pipelined_product(Ops& ops, bool pipe, bool active, int group, std::size_t c0, std::size_t nc, std::size_t panel)
{
    std::size_t c = c0;
    while (c < c0 + nc) {
        const std::size_t fp = c / panel;                                   // fixed panel index
        const std::size_t cend = pipe ? std::min(c0 + nc, (fp + 1) * panel) : c0 + nc;
        const std::size_t w = cend - c;
        if (pipe) ops.event_wait((int)fp);                                  // the previous product's all-reduce of my input panel
        ops.product(c, w);
        if (active) {
            ops.allreduce(c, w, pipe);                                      // asynchronous when pipelined
            if (pipe) ops.event_record(group, (int)fp);
        }
        c = cend;
    }
}

} // namespace chase_amd
