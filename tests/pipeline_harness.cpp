// pipeline_harness.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// The event-ordering state machine of the panel-pipelined distributed HEMM (chase_amd/host/panel_pipeline.hpp, the code
// pChaseHip::hemm_ptr runs) on the CPU: the four actions are bound to a SIMULATOR of HIP streams and events that mirrors the
// grid's collective logic (chase_amd/csrc/grid.hip: a collective runs on its group's communication stream after an event
// recorded on the compute stream; per-panel slot events; chase_hip_grid_wait; one or two communication streams) and gives
// every operation a vector clock.  A filter-like sequence of alternating products (column -> row with the column group's
// all-reduce, row -> column with the row group's; shrinking widths, moving offsets, beta != 0) is then checked: every two
// operations that touch the same columns of the same block, at least one of them writing, must be ordered (happens-before)
// in issue order.  Output: "OK <ops>" or "HAZARD ..." per configuration.
//   usage: pipeline_harness <col_active 0|1> <row_active 0|1> <streams 1|2> <panel> <rule: fixed|r4bug> [pipelined 0|1]
// rule r4bug = round 4's first version of the rule (panelise only when the product's OWN group is active): the simulator must
// find the race of the one-column grid with it - that is what validates the checker.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../chase_amd/host/panel_pipeline.hpp"

using VC = std::array<long, 3>;                        // stream 0 = compute, 1 / 2 = communication streams
struct Sim {
    int nstreams = 2;
    bool active[2] = {true, true};                     // [ROW = 0], [COL = 1]
    long count[3] = {0, 0, 0};
    VC clock[3] = {{{0, 0, 0}}, {{0, 0, 0}}, {{0, 0, 0}}};   // what the NEXT op of a stream is already ordered after
    struct Ev { bool set = false; VC vc{}; };
    Ev ev_compute, ev_comm[2];
    std::vector<Ev> slots[2];
    bool pending[2] = {false, false};
    struct Acc { VC vc; int stream; long idx; std::string what; };
    // per buffer (0 = V / column-type, 1 = W / row-type) per column: last writer and the readers since
    std::vector<Acc> last_write[2];
    std::vector<std::vector<Acc>> readers[2];
    std::vector<std::string> hazards;
    long nops = 0;
    explicit Sim(std::size_t ncols) { for (int b = 0; b < 2; ++b) { last_write[b].assign(ncols, Acc{{{0, 0, 0}}, -1, 0, ""}); readers[b].assign(ncols, {}); } }
    int sidx(int group) const { return nstreams == 2 ? group : 0; }
    static VC vmax(VC a, const VC& b) { for (int i = 0; i < 3; ++i) a[i] = std::max(a[i], b[i]); return a; }
    static bool after(const VC& later, const Acc& a) { return a.stream < 0 || later[a.stream] >= a.idx; }
    Acc op(int stream, const std::string& what)
    {
        ++count[stream];
        clock[stream][stream] = count[stream];
        ++nops;
        return Acc{clock[stream], stream, count[stream], what};
    }
    Ev record(int stream) { Ev e; e.set = true; e.vc = clock[stream]; return e; }
    void wait(int stream, const Ev& e) { if (e.set) clock[stream] = vmax(clock[stream], e.vc); }
    void touch(const Acc& a, int buf, std::size_t c, std::size_t w, bool write)
    {
        for (std::size_t j = c; j < c + w; ++j) {
            const Acc& lw = last_write[buf][j];
            if (!after(a.vc, lw)) hazards.push_back(a.what + " reads/writes column " + std::to_string(j) + " of block " + std::to_string(buf) + " unordered after write by " + lw.what);
            if (write) {
                for (const Acc& r : readers[buf][j])
                    if (!after(a.vc, r)) hazards.push_back(a.what + " overwrites column " + std::to_string(j) + " of block " + std::to_string(buf) + " while " + r.what + " may still read it");
                last_write[buf][j] = a;
                readers[buf][j].clear();
            } else readers[buf][j].push_back(a);
        }
    }
    // ---- the grid's collective logic (grid.hip) -------------------------------------------------------------------------
    void grid_wait()
    {
        for (int i = 0; i < 2; ++i) {
            if (!pending[i]) continue;
            ev_comm[i] = record(1 + i);
            wait(0, ev_comm[i]);
            pending[i] = false;
        }
    }
    void allreduce(int group, int buf, std::size_t c, std::size_t w, bool async, const std::string& tag)
    {
        if (!active[group]) return;
        const int si = sidx(group);
        ev_compute = record(0);
        wait(1 + si, ev_compute);
        Acc a = op(1 + si, "allreduce " + tag);
        touch(a, buf, c, w, true);
        pending[si] = true;
        if (!async) grid_wait();
    }
    void event_record_on(int group, int slot)
    {
        const int si = sidx(group);
        if ((int)slots[si].size() <= slot) slots[si].resize(slot + 1);
        slots[si][slot] = record(1 + si);
    }
    void event_wait(int slot)
    {
        for (int si = 0; si < 2; ++si)
            if (slot < (int)slots[si].size()) wait(0, slots[si][slot]);
    }
};

struct Ops {
    Sim& s; bool bAc; int group; bool beta; int step;
    void event_wait(int slot) { s.event_wait(slot); }
    void product(std::size_t c, std::size_t w)
    {
        const int in = bAc ? 0 : 1, out = bAc ? 1 : 0;
        Sim::Acc a = s.op(0, "gemm step " + std::to_string(step) + (bAc ? " bAc" : " cAb") + " cols " + std::to_string(c) + "+" + std::to_string(w));
        s.touch(a, in, c, w, false);
        if (beta) s.touch(a, out, c, w, false);
        s.touch(a, out, c, w, true);
    }
    void allreduce(std::size_t c, std::size_t w, bool async) { s.allreduce(group, bAc ? 1 : 0, c, w, async, "step " + std::to_string(step) + " cols " + std::to_string(c) + "+" + std::to_string(w)); }
    void event_record(int grp, int slot) { s.event_record_on(grp, slot); }
};

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    const bool col_active = std::atoi(argv[1]) != 0, row_active = std::atoi(argv[2]) != 0;
    const int streams = std::atoi(argv[3]);
    const std::size_t panel = (std::size_t)std::atoi(argv[4]);
    const bool r4bug = std::strcmp(argv[5], "r4bug") == 0;
    const bool pipelined = argc > 6 ? std::atoi(argv[6]) != 0 : true;
    const std::size_t ncols = 1000;
    Sim s(ncols);
    s.nstreams = streams;
    s.active[0] = row_active; s.active[1] = col_active;
    // a filter: 2 calls of 12 steps; columns retire from the left by uneven amounts, beta != 0 after each call's first step
    int step = 0;
    for (int call = 0; call < 2; ++call) {
        std::size_t c0 = call ? 37 : 0, nc = ncols - c0;
        bool bAc = true;
        for (int i = 0; i < 12; ++i, ++step) {
            const int group = bAc ? 1 : 0;                        // CHASE_HIP_COL for the column -> row product
            const bool active = s.active[group], other = s.active[1 - group];
            const bool pipe = pipelined && (r4bug ? active : (active || other));
            Ops ops{s, bAc, group, i > 0, step};
            chase_amd::pipelined_product(ops, pipe, active, group, c0, nc, panel);
            bAc = !bAc;
            if (i % 2 == 1) { const std::size_t drop = 61 + 13 * (std::size_t)i; c0 += drop; nc -= drop; }
        }
        // FilterPhaseEnd: sync_comm, then something on the compute stream reads and rewrites both blocks entirely (QR, RR)
        s.grid_wait();
        Sim::Acc a = s.op(0, "QR after call " + std::to_string(call));
        for (int b = 0; b < 2; ++b) s.touch(a, b, 0, ncols, true);
    }
    if (s.hazards.empty()) std::printf("OK %ld\n", s.nops);
    else {
        std::printf("HAZARD %zu\n", s.hazards.size());
        for (std::size_t i = 0; i < s.hazards.size() && i < 5; ++i) std::printf("  %s\n", s.hazards[i].c_str());
    }
    return 0;
}
