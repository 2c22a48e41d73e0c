// roctx.hpp — profiler ranges around the ChaseBase virtuals of the Impls (CHASE_HIP_ROCTX=1).
//
// The reference brackets its GPU Impl's phases with NVTX ranges (Impl/chase_gpu/nvtx.hpp:36-78: ScopedNvtxRange "QR", "RR",
// "Resd", "Lanczos", ...); the MI355X equivalent is roctx.  With CHASE_HIP_ROCTX=1 every virtual of the four
// Impls pushes a range named after itself ("chase:QR", "chase:RR", "chase:Resd", "chase:Lanczos", "chase:Filter", ...) and, when
// the range ends, waits for the context's stream first - so that the kernels a phase launched lie inside its range and
// `rocprofv3 --kernel-trace --marker-trace` splits a solve's kernel table by phase (scripts/phase_table.py).  The library
// (librocprofiler-sdk-roctx, else libroctx64) is bound with dlopen on first use: no link-time dependency, nothing happens
// without the switch.
#pragma once
#include <dlfcn.h>
#include <cstdlib>
#include "../../include/chase_hip.h"

namespace chase_amd {

struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    static RoctxApi& get()
    {
        static RoctxApi api = [] {
            RoctxApi a;
            const char* e = std::getenv("CHASE_HIP_ROCTX");
            if (!e || std::atoi(e) == 0) return a;
            for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                    a.pop = (int (*)())dlsym(h, "roctxRangePop");
                    if (a.push && a.pop) { a.on = true; break; }
                }
            }
            return a;
        }();
        return api;
    }
};

inline bool roctx_enabled() { return RoctxApi::get().on; }
inline void roctx_push(const char* name) { auto& a = RoctxApi::get(); if (a.on) a.push(name); }
// the phase's device work is complete before its range closes
inline void roctx_pop(chase_hip_ctx* ctx) { auto& a = RoctxApi::get(); if (a.on) { if (ctx) chase_hip_ctx_sync(ctx); a.pop(); } }

struct PhaseRange {
    chase_hip_ctx* ctx;
    bool on;
    PhaseRange(chase_hip_ctx* c, const char* name) : ctx(c), on(roctx_enabled()) { if (on) roctx_push(name); }
    ~PhaseRange() { if (on) roctx_pop(ctx); }
    PhaseRange(const PhaseRange&) = delete;
    PhaseRange& operator=(const PhaseRange&) = delete;
};

} // namespace chase_amd
#define CHASE_PHASE(ctx, name) ::chase_amd::PhaseRange chase_phase_range_((ctx), "chase:" name)
