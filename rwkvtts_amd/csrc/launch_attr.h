// rwkvtts_amd/csrc/launch_attr.h -- hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, DEVICE), thread-safe.
// The attribute belongs to the device the calling thread has current (a process that runs a kernel on a second GPU must set it there
// too: ADVICE round 5), so each launcher keeps one bit per device ordinal in its own `static DynLdsOnce` object.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace rwkv7 {
struct DynLdsOnce {
    std::atomic<unsigned long long> done[4] = {};   // device ordinals 0..255
    hipError_t ensure(const void *kernel, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::atomic<unsigned long long> &word = done[(dev >> 6) & 3];
        const unsigned long long bit = 1ull << (dev & 63);
        if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);   // idempotent: a race sets it twice
        if (e == hipSuccess) word.fetch_or(bit, std::memory_order_release);
        return e;
    }
};
}  // namespace rwkv7
