// Library-level entry points of the C ABI (include/neurite_amd.h).
#include <mutex>

#include "nrt_common.h"

// ---- self-cleaning counter slots ---------------------------------------------------------------------------------------------
// Kernels that coordinate their blocks through atomic counters (work lists of the persistent gather, "last block finishes" reductions)
// used to have the counters zeroed by a launch of their own in front -- 3-4 us of a 270 us step at batch 1, and hipMemsetAsync is not an
// option inside captured graphs (nrt_common.h).  Here the counters live in a device-resident ring of NRT_RING_SLOTS slots of
// NRT_RING_WORDS zero-initialised words; a launch takes the next slot (round robin, so launches in flight on different streams do not
// share one as long as fewer than NRT_RING_SLOTS overlap) and the LAST block to leave the kernel writes the zeros back.  A slot baked
// into a captured hipGraph stays valid: replays of one graph are ordered, and each leaves its slot clean.
__device__ unsigned nrt_ring_words[NRT_RING_SLOTS * NRT_RING_WORDS];

unsigned *nrt_ring_slot() {
    static unsigned *base[64];
    static unsigned next[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!base[dev]) {
        std::lock_guard<std::mutex> lock(mu);
        void *p = nullptr;
        if (!base[dev] && (hipGetSymbolAddress(&p, HIP_SYMBOL(nrt_ring_words)) != hipSuccess || !p)) return nullptr;
        if (!base[dev]) base[dev] = (unsigned *)p;
    }
    return base[dev] + (size_t)(__atomic_fetch_add(&next[dev], 1u, __ATOMIC_RELAXED) % NRT_RING_SLOTS) * NRT_RING_WORDS;
}

extern "C" const char *nrt_status_string(int status) {
    switch (status) {
        case NRT_OK: return "ok";
        case NRT_ERR_INVALID_ARG: return "invalid argument";
        case NRT_ERR_UNSUPPORTED: return "combination not supported by the HIP path";
        case NRT_ERR_LAUNCH: return "HIP kernel launch failed";
        case NRT_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}

extern "C" int nrt_abi_version(void) { return 1; }

// sha256 (16 hex digits) over the compile flags, the headers and every .hip file this library was built from; neurite_amd/build.py
// compares it -- read from the file, not through this function -- with the sources on disk to decide whether the library is stale
#ifndef NRT_BUILD_ID_STRING
#define NRT_BUILD_ID_STRING "unknown"
#endif
extern "C" const char *nrt_build_id(void) {
    static const char id[] = "NRT_BUILD_ID=" NRT_BUILD_ID_STRING;
    return id + 13;
}

extern "C" const char *nrt_target_arch(void) { return "gfx950"; }
