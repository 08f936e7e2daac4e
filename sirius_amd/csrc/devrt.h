// devrt.h -- HIP runtime include, launch macro and error plumbing shared by the kernels.
//
// Product builds are hipcc / gfx950 only.  SRS_EMU is defined solely by tests/emu/ (the CPU
// logic-emulator build used because the dev container has no GPU); it is never part of
// libsirius_amd.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(SRS_EMU)
#include "hipemu.h"
#define SRS_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipemu::launch(kernel, dim3 grid, dim3 block, (size_t)(smem), __VA_ARGS__)
#define SRS_KERNEL_BOUNDS(t, w)
#else
#include <hip/hip_runtime.h>
#define SRS_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3 grid, dim3 block, (size_t)(smem), (hipStream_t)(stream), __VA_ARGS__)
#define SRS_KERNEL_BOUNDS(t, w) __launch_bounds__(t, w)
#endif

#include <string>

namespace srs {

// The ONE place host code learns that it runs on the CPU logic emulator (tests/emu): callers branch on these with `if constexpr`,
// not with the preprocessor (r04: capi.hip had seven SRS_EMU blocks).
namespace rt {
#if defined(SRS_EMU)
constexpr bool kEmulated = true;      // one global execution context: no per-shard worker threads, no hiprtc, no device properties
inline bool device_arch(int, std::string &name) { name = "gfx950 (hipemu)"; return true; }
#else
constexpr bool kEmulated = false;
inline bool device_arch(int dev, std::string &name) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    name = prop.gcnArchName;
    return true;
}
#endif
}  // namespace rt

// thread-local last-error text, surfaced through srs_last_error()
void set_error(const std::string &msg);
const char *get_error();

struct DeviceError {
    int rc;
};

#define SRS_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::srs::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                             std::to_string(__LINE__) + ")");                                     \
            throw ::srs::DeviceError{5};                                                          \
        }                                                                                         \
    } while (0)

// Grow-only device scratch arena (one per handle); avoids hipMalloc on the hot path.
struct Arena {
    void *base = nullptr;
    size_t cap = 0, used = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (base) SRS_HIP_CHECK(hipFree(base));
        base = nullptr;
        cap = 0;
        SRS_HIP_CHECK(hipMalloc(&base, bytes));
        cap = bytes;
    }
    void reset() { used = 0; }
    template <class T>
    T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > cap) {
            set_error("internal: arena overflow");
            throw DeviceError{5};
        }
        T *p = reinterpret_cast<T *>(static_cast<char *>(base) + used);
        used += bytes;
        return p;
    }
    static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = used = 0;
    }
};

// A grow-only scratch arena owned by a host THREAD (a `static thread_local`), not by a handle: it follows its thread when srs_init_thread
// re-binds the thread to another device (the memory of the old device is released; handle-owned arenas never move).
struct ThreadArena : Arena {
    int dev = -1;
    void reserve(size_t bytes) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (base && cur != dev) release();
        dev = cur;
        Arena::reserve(bytes);
    }
};

}  // namespace srs
