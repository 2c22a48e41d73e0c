// output_override.hpp — the one virtual of chase::ChaseBase<T> that exists only in a reference build configured with
// -DCHASE_OUTPUT: `virtual void Output(LogLevel, std::string, const char* category = "algorithm") = 0`
// (algorithm/interface.hpp:419-432).  The reference's Impls forward it to the logger singleton
// (Impl/chase_gpu/chase_gpu.hpp:465-469, Impl/pchase_gpu/pchase_gpu.hpp:618-622); so do these.
//
// The four Impl classes derive from WithOutput<BaseT> instead of BaseT: when BaseT declares Output (found by the detection idiom,
// the level type is read off the member's signature) the override is supplied, otherwise (this repository's mirror base, or a
// reference build without the option) WithOutput<BaseT> is an empty layer.  No preprocessor switch decides the class layout:
// the same Impl headers build in every configuration of the checkout.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#if defined(CHASE_OUTPUT) && defined(__has_include)
#if __has_include("algorithm/logger.hpp")
#include "algorithm/logger.hpp"
#define CHASE_HIP_HAVE_REF_LOGGER 1
#endif
#endif

namespace chase_amd {

template <class BaseT, class = void>
class WithOutput : public BaseT {
public:
    static constexpr bool overrides_output = false;
};

template <class BaseT>
class WithOutput<BaseT, std::void_t<decltype(&BaseT::Output)>> : public BaseT {
    template <class L> static L level_of(void (BaseT::*)(L, std::string, const char*));
public:
    static constexpr bool overrides_output = true;
    using LogLevel = decltype(level_of(&BaseT::Output));
    void Output(LogLevel level, std::string str, const char* category = "algorithm") override
    {
#ifdef CHASE_HIP_HAVE_REF_LOGGER
        chase::GetLogger().Log(level, category, str, this->get_rank());
#else
        // a base with Output but without the reference's logger in reach: the logger's rules (algorithm/logger.hpp:156-168)
        // on the same environment variables - level <= CHASE_LOG_LEVEL (default warn), rank == CHASE_LOG_RANK (default 0, -1 all)
        static const int lim = [] {
            const char* e = std::getenv("CHASE_LOG_LEVEL");
            if (!e) return 1;
            switch (e[0] | 0x20) { case 'e': return 0; case 'w': return 1; case 'd': return 3; case 't': return 4; default: return 2; }
        }();
        static const int only = [] { const char* e = std::getenv("CHASE_LOG_RANK"); return e ? std::atoi(e) : 0; }();
        (void)category;
        if (static_cast<int>(level) > lim || (only >= 0 && this->get_rank() != only)) return;
        std::fputs(str.c_str(), stdout);
        std::fflush(stdout);
#endif
    }
};

} // namespace chase_amd
