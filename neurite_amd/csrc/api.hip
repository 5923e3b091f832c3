// Library-level entry points of the C ABI (include/neurite_amd.h).
#include <mutex>
#include <unordered_map>

#include "nrt_common.h"

// ---- self-cleaning counter slots ---------------------------------------------------------------------------------------------
// Kernels that coordinate their blocks through atomic counters (work lists of the persistent gather, "last block finishes" reductions)
// used to have the counters zeroed by a launch of their own in front -- 3-4 us of a 270 us step at batch 1, and hipMemsetAsync is not an
// option inside captured graphs (nrt_common.h).  Here the counters live in a device-resident pool of NRT_RING_SLOTS slots of
// NRT_RING_WORDS zero-initialised words, and the LAST block to leave a kernel writes the zeros back.
//
// Who owns a slot (round 6; ADVICE r5: the round-robin ring of round 5 let two launches in flight on different streams, or a replayed
// graph and eager work, meet in one slot once more than 64 launches lay between them):
//   * eager launches: ONE slot per (device, stream), taken from the bottom of the pool the first time the stream is seen.  Launches on a
//     stream execute in order, so the slot is clean whenever the next kernel on that stream starts; no other stream ever sees it.
//   * launches recorded during stream capture: a slot of their own from the upper part of the pool -- a captured graph may be replayed
//     on any stream next to any eager work (replays of ONE graph are ordered by the runtime; replaying the same graph concurrently with
//     itself is as wrong for its output buffers as for its counters).  Captured slots are handed out in a ring of NRT_RING_SLOTS -
//     NRT_RING_STREAM_SLOTS entries: a process would have to keep more than that many captured launches ALIVE (7 168: graphs are rarely
//     destroyed and never reported to this library) before the oldest graph's slot meets a new one.
//   * the two users keep disjoint words (gather: NRT_RING_GATHER_OFF, weighted CCE: NRT_RING_CCE_OFF), so even kernels that did share a
//     slot could not read each other's tickets.
// nullptr when more than NRT_RING_STREAM_SLOTS streams have launched on a device: the callers fall back to their counter-free form
// (fused_wc.h: one block per item) or report NRT_ERR_WORKSPACE.  nrt_init() resolves the symbol outside any capture; nrt_counters_reset()
// re-zeroes the pool after a kernel was aborted mid-flight (a fault, a hipDeviceReset-less recovery): nothing else leaves a slot dirty.
__device__ unsigned nrt_ring_words[NRT_RING_SLOTS * NRT_RING_WORDS];

namespace {
struct RingDev {
    unsigned *base = nullptr;
    unsigned next_stream = 0;                       // slots [0, next_stream) belong to streams (at most NRT_RING_STREAM_SLOTS)
    unsigned next_graph = 0;                        // captured launches: slot NRT_RING_STREAM_SLOTS + next_graph % (the rest), a ring
    std::unordered_map<uintptr_t, unsigned> by_stream;
};
RingDev g_ring[64];
std::mutex g_ring_mu;

RingDev *ring_dev() {                               // (caller holds g_ring_mu)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    RingDev &r = g_ring[dev];
    if (!r.base) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(nrt_ring_words)) != hipSuccess || !p) return nullptr;
        r.base = (unsigned *)p;
    }
    return &r;
}
}  // namespace

unsigned *nrt_ring_slot(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_ring_mu);
    RingDev *r = ring_dev();
    if (!r) return nullptr;
    unsigned slot;
    if (capturing) {
        slot = NRT_RING_STREAM_SLOTS + (r->next_graph++ % (NRT_RING_SLOTS - NRT_RING_STREAM_SLOTS));
    } else {
        auto it = r->by_stream.find((uintptr_t)st);
        if (it != r->by_stream.end()) slot = it->second;
        else {
            if (r->next_stream >= NRT_RING_STREAM_SLOTS) return nullptr;
            slot = r->next_stream++;
            r->by_stream.emplace((uintptr_t)st, slot);
        }
    }
    return r->base + (size_t)slot * NRT_RING_WORDS;
}

extern "C" int nrt_init(void) {
    std::lock_guard<std::mutex> lock(g_ring_mu);
    return ring_dev() ? NRT_OK : NRT_ERR_LAUNCH;
}

extern "C" int nrt_counters_reset(void *stream) {
    unsigned *base;
    {
        std::lock_guard<std::mutex> lock(g_ring_mu);
        RingDev *r = ring_dev();
        if (!r) return NRT_ERR_LAUNCH;
        base = r->base;
    }
    return nrt_zero_async(base, (size_t)NRT_RING_SLOTS * NRT_RING_WORDS * 4, nrt_stream(stream)) == hipSuccess ? NRT_OK : NRT_ERR_LAUNCH;
}

// (tests: which slot a launch on `stream` would use right now, as an index into the pool; -1 = none)
extern "C" int nrt_counters_slot_index(void *stream) {
    unsigned *p = nrt_ring_slot(nrt_stream(stream));
    if (!p) return -1;
    std::lock_guard<std::mutex> lock(g_ring_mu);
    RingDev *r = ring_dev();
    return r ? (int)((p - r->base) / NRT_RING_WORDS) : -1;
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
