// jit.h -- run-time compilation of straight-line row programs with hiprtc (see jit.hip).
#pragma once
#include <string>

#include "devrt.h"

namespace srs {
namespace jit {

struct Kernel {
    void *module = nullptr;     // hipModule_t
    void *function = nullptr;   // hipFunction_t
    double compile_seconds = 0;
};

// hiprtc present and not disabled (SRS_NO_JIT)
bool enabled();
// Compiles `source` for gfx950 (it may #include "field.cuh" and "rowprog_dev.cuh": the library carries their text) and
// loads `entry`.  false + log on failure; never throws.
bool compile(const std::string &source, const char *entry, Kernel &out, std::string &log);
// The hiprtc half alone (no device, nothing loaded): does `source` compile against the embedded headers?
bool compile_only(const std::string &source, size_t *code_bytes, std::string &log);
void release(Kernel &k);
// grid x 128 threads, `smem_bytes` of dynamic LDS, one struct argument passed by value
bool launch(const Kernel &k, unsigned blocks, unsigned threads, unsigned smem_bytes, const void *arg_struct, hipStream_t st);

}  // namespace jit
}  // namespace srs
