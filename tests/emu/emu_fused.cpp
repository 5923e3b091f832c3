// TEST INFRASTRUCTURE ONLY -- compiles neurite_amd/csrc/fused.hip (a patched copy, see build_emu.py) for the host on top of
// tests/emu/hip/hip_runtime.h.  The C entry points of the translation unit (nrt_warp_dice_soft_f32, ...) are then ordinary
// host functions operating on host memory.
#include <hip/hip_runtime.h>

// dynamic LDS of the kernels in this translation unit (`extern __shared__` arrays, patched to plain `extern`)
namespace {            // the kernels live in the translation unit's anonymous namespace, so do their dynamic-LDS arrays
alignas(16) unsigned char dd_smem[160 * 1024];
float fs[4096];
}  // namespace

#include "csrc/fused.hip"

// largest dynamic-LDS size requested by a launch since the previous call (identifies the kernel that ran)
extern "C" size_t emu_take_max_shmem(void) {
    const size_t v = emu_max_shmem;
    emu_max_shmem = 0;
    return v;
}
