// Private helpers shared by the HIP translation units of libcrossloc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stddef.h>

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel and a plan may be lowered for
// any device of the node (one process driving several GPUs), from any thread: remember the largest size configured
// per (call site, device) in atomics.  Racing threads may both set the attribute - it is idempotent and cheap.
struct XlLdsLimit {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> bytes[kMaxDevices];
    XlLdsLimit() { for (auto &b : bytes) b.store(0, std::memory_order_relaxed); }
    // true when `lds` bytes still have to be configured on the current device (call done() after doing so)
    bool needs(size_t lds, int *device) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) { *device = -1; return true; }
        *device = d;
        return bytes[d].load(std::memory_order_acquire) < lds;
    }
    void done(size_t lds, int device) {
        if (device < 0) return;
        size_t cur = bytes[device].load(std::memory_order_relaxed);
        while (cur < lds && !bytes[device].compare_exchange_weak(cur, lds, std::memory_order_release)) {}
    }
};

// Sum over the 32 lanes of a half wave (lanes 0-31 / 32-63) as a fixed fp32 tree on the DPP path, one v_add_f32 per level, no
// LDS traffic: pairs, quads (quad_perm), half rows (row_half_mirror), rows (row_mirror), then lane 15 of rows 0 / 2 broadcast
// into rows 1 / 3 (row_bcast:15).  The total is valid in the UPPER 16 lanes of the half only (lanes 16-31 / 48-63).
// xl_wave_sum_top adds the lower half's total (lane 31, row_bcast:31): valid in lanes 48-63.  Used by the GroupNorm-statistics
// epilogues of the stem kernels, whose groups of 2 - 8 channels lie within one lane / one or both halves.
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ float xl_dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ float xl_half_wave_sum(float v)
{
    v += xl_dpp_f32<0xB1>(v);                            // quad_perm [1,0,3,2]
    v += xl_dpp_f32<0x4E>(v);                            // quad_perm [2,3,0,1]
    v += xl_dpp_f32<0x141>(v);                           // row_half_mirror
    v += xl_dpp_f32<0x140>(v);                           // row_mirror
    v += xl_dpp_f32<0x142, 0xA>(v);                      // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add 0)
    return v;
}
__device__ __forceinline__ float xl_wave_sum_top(float v)
{
    v = xl_half_wave_sum(v);
    v += xl_dpp_f32<0x143, 0xC>(v);                      // row_bcast:31 into rows 2 and 3
    return v;
}

// Maximum of non-negative floats over a launch, as float bits in *slot (non-negative floats order like their bits): a DPP tree over
// the wave, then lane 63 issues the atomicMax only when its value beats what the slot already holds (a plain load first: thousands of
// waves on one address otherwise serialise in L2 - measured +0.05 ms per GroupNorm-backward launch; after the first few arrivals
// almost every wave skips the atomic).  slot may be nullptr.
// PRECONDITIONS (every caller - gnb_apply, wino6_dy, pair_weight_kernel<0> - meets them): blockDim.x is a multiple of 64 and
// ALL 64 lanes of the wave reach this call with values >= 0 (no early return in front of it; an out-of-range thread passes 0): the
// DPP tree reads neighbours' registers and lane 63 commits.  A wave that arrives with lanes switched off falls back to one
// atomicMax per live lane - slower, never wrong.  NaN: fmaxf drops it in the tree; a wave of NaNs commits nothing, and the
// fallback's integer compare would order a NaN's bits ABOVE every finite value - so the fallback commits finite values only (a
// non-finite gradient leaves the slot at the finite maximum: the scale then saturates the operand to fp16 inf, which the
// training loop's own non-finite loss check sees).
__device__ __forceinline__ void xl_wave_max_commit(float m, unsigned *slot)
{
    if (slot == nullptr) return;
    if (__builtin_amdgcn_read_exec() != ~0ull) {                       // partial wave: no tree
        const unsigned b = __builtin_bit_cast(unsigned, m);
        if (m >= 0.f && b < 0x7f800000u && b > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, b);
        return;
    }
    m = fmaxf(m, xl_dpp_f32<0xB1>(m));
    m = fmaxf(m, xl_dpp_f32<0x4E>(m));
    m = fmaxf(m, xl_dpp_f32<0x141>(m));
    m = fmaxf(m, xl_dpp_f32<0x140>(m));
    m = fmaxf(m, xl_dpp_f32<0x142, 0xA>(m));
    m = fmaxf(m, xl_dpp_f32<0x143, 0xC>(m));                          // lane 63: the maximum over the wave (values are >= 0)
    if ((threadIdx.x & 63) == 63) {
        const unsigned b = __builtin_bit_cast(unsigned, m);
        if (b > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, b);
    }
}
