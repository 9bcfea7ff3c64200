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
