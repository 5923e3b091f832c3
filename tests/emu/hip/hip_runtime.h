// TEST INFRASTRUCTURE ONLY -- a host stand-in for <hip/hip_runtime.h> that lets a kernel translation unit of
// neurite_amd/csrc be compiled for the CPU (clang++, -std=c++20) and run one thread block at a time on 256 OS threads:
// __syncthreads is a real barrier, the wave shuffles exchange through per-wave buffers with a per-wave barrier, LDS
// atomics are real atomics, `__shared__` arrays are function-static (one block is alive at any time), launches are
// synchronous.  It models the programming model (convergent waves of 64, block barriers), not timing and not the
// memory-ordering subtleties of the hardware.  Used by tests/test_emu_fused.py to exercise kernel logic when no GPU is
// available; parity on hardware is still the -m gpu tests' job.
#pragma once

#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct EmuIdx { unsigned x = 0, y = 0, z = 0; };

inline thread_local EmuIdx threadIdx;
inline EmuIdx blockIdx;
inline dim3 gridDim, blockDim;
inline std::barrier<> *emu_block_barrier = nullptr;
inline std::barrier<> *emu_wave_barrier[16] = {};
inline uint32_t emu_wave_buf[16][64];
inline size_t emu_max_shmem = 0;       // largest dynamic-LDS request since the last reset (tells the tests which kernel ran)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

typedef int hipError_t;
constexpr int hipSuccess = 0;
typedef void *hipStream_t;
inline hipError_t hipGetLastError() { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline int hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }

inline void __syncthreads() { emu_block_barrier->arrive_and_wait(); }

template <typename T>
inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    (void)width;
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    emu_wave_buf[w][l] = bits;
    emu_wave_barrier[w]->arrive_and_wait();
    const uint32_t r = emu_wave_buf[w][(unsigned)src & 63u];
    emu_wave_barrier[w]->arrive_and_wait();
    T out;
    std::memcpy(&out, &r, 4);
    return out;
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (int)((threadIdx.x & 63u) ^ (unsigned)mask), width); }

inline float __fmul_rn(float a, float b) { return a * b; }          // the emulator is built with -ffp-contract=off
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline int __mul24(int a, int b) { return (int)(((a << 8) >> 8) * ((b << 8) >> 8)); }
using std::max;
using std::min;

inline unsigned atomicCAS(unsigned *addr, unsigned expected, unsigned desired) {
    std::atomic_ref<unsigned> ref(*addr);
    unsigned e = expected;
    ref.compare_exchange_strong(e, desired);
    return e;                                                        // the old value, as the device intrinsic
}

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
struct __amdgpu_buffer_rsrc_t {
    const char *base;
    unsigned bytes;
};
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short, int num_records, int) {
    return __amdgpu_buffer_rsrc_t{(const char *)p, (unsigned)num_records};
}
typedef int emu_i4 __attribute__((ext_vector_type(4)));
inline emu_i4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    const unsigned long long off = (unsigned long long)voff + soff;
    emu_i4 v = {0, 0, 0, 0};
    if (off + 16ull <= r.bytes) std::memcpy(&v, r.base + off, 16);   // out-of-range reads return zero and touch nothing
    return v;
}

// one block at a time, its threads on OS threads
template <typename K, typename... Args>
inline void emu_launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    emu_max_shmem = std::max(emu_max_shmem, shmem);
    const unsigned nthr = block.x * block.y * block.z, nwave = (nthr + 63) / 64;
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                std::barrier<> bb((std::ptrdiff_t)nthr);
                emu_block_barrier = &bb;
                std::vector<std::unique_ptr<std::barrier<>>> wb;
                for (unsigned w = 0; w < nwave; ++w) {
                    wb.emplace_back(new std::barrier<>((std::ptrdiff_t)std::min(64u, nthr - 64 * w)));
                    emu_wave_barrier[w] = wb.back().get();
                }
                std::vector<std::thread> th;
                th.reserve(nthr);
                for (unsigned t = 0; t < nthr; ++t)
                    th.emplace_back([=]() {
                        threadIdx.x = t % block.x;
                        threadIdx.y = (t / block.x) % block.y;
                        threadIdx.z = t / (block.x * block.y);
                        kernel(args...);
                    });
                for (auto &x : th) x.join();
            }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)
