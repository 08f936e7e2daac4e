// jit_emu.cpp -- TEST INFRASTRUCTURE: the run-time compilation interface (csrc/jit.h) for the CPU logic emulator.
// The product compiles the emitted translation unit with hiprtc and loads it with hipModuleLoadData (csrc/jit.hip); here the SAME
// text is compiled with g++ against the emulator headers into a shared object and entered through srs_jit_launch_emu, so the
// whole path -- plan_sweep / emit_*_source for an arbitrary circuit, the kernel bodies of rowprog_dev.cuh, the launch parameters --
// runs in the GPU-less container.  Off unless SRS_EMU_JIT=1 (a compile takes ~10 s).
#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "jit.h"
#include "tuning.h"

namespace srs {
namespace jit {

bool enabled() {
    return std::getenv("SRS_EMU_JIT") != nullptr && tuning::get_or(tuning::NO_JIT, std::getenv("SRS_NO_JIT") != nullptr ? 1 : 0) == 0;
}

static bool build(const std::string &source, const std::string &out_so, bool syntax_only, double &seconds, std::string &log) {
    static std::atomic<int> seq{0};
    const std::string base = "/tmp/srs_emu_jit_" + std::to_string((long)getpid()) + "_" + std::to_string(seq++);
    const std::string src = base + ".cpp", errf = base + ".err";
    // the emitted unit defines the kernel srs_jit_rowprog; the emulator enters it through this launcher (the product never sees it)
    static const char kLauncher[] =
        "\nextern \"C\" void srs_jit_launch_emu(unsigned blocks, unsigned threads, unsigned smem, const void *a) {\n"
        "    hipemu::launch(srs::rowprog::srs_jit_rowprog, dim3(blocks), dim3(threads), (size_t)smem, *static_cast<const srs::rowprog::DevArgs *>(a));\n}\n";
    const std::string unit = source + kLauncher;
    if (FILE *f = std::fopen(src.c_str(), "wb")) { std::fwrite(unit.data(), 1, unit.size(), f); std::fclose(f); }
    else { log = "cannot write " + src; return false; }
    const std::string cmd = std::string("g++ -std=c++20 -O1 -fPIC -DSRS_EMU -I" SRS_EMU_INC1 " -I" SRS_EMU_INC2
                                        " -pthread -Wno-unknown-pragmas -Wno-attributes ") +
                            (syntax_only ? "-fsyntax-only " : "-shared -o " + out_so + " ") + src + " 2> " + errf;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = std::system(cmd.c_str());
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) {
        log = "g++ failed: " + cmd + "\n";
        if (FILE *f = std::fopen(errf.c_str(), "rb")) {
            char buf[4096];
            size_t n = std::fread(buf, 1, sizeof buf - 1, f);
            buf[n] = 0;
            log += buf;
            std::fclose(f);
        }
    }
    if (!std::getenv("SRS_JIT_DUMP")) std::remove(src.c_str());
    std::remove(errf.c_str());
    return rc == 0;
}

bool compile_only(const std::string &source, size_t *code_bytes, std::string &log) {
    double s = 0;
    if (!build(source, "", true, s, log)) return false;
    if (code_bytes) *code_bytes = source.size();
    return true;
}

bool compile(const std::string &source, const char *entry, Kernel &out, std::string &log) {
    (void)entry;
    static std::atomic<int> seq{0};
    const std::string so = "/tmp/srs_emu_jit_" + std::to_string((long)getpid()) + "_" + std::to_string(seq++) + ".so";
    if (!build(source, so, false, out.compile_seconds, log)) return false;
    void *h = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    std::remove(so.c_str());                                 // the mapping stays
    if (!h) { log = std::string("dlopen: ") + dlerror(); return false; }
    void *fn = dlsym(h, "srs_jit_launch_emu");
    if (!fn) { log = "srs_jit_launch_emu not found in the compiled object"; dlclose(h); return false; }
    out.module = h;
    out.function = fn;
    return true;
}

void release(Kernel &k) {
    if (k.module) dlclose(k.module);
    k.module = k.function = nullptr;
}

bool launch(const Kernel &k, unsigned blocks, unsigned threads, unsigned smem_bytes, const void *arg_struct, hipStream_t) {
    if (!k.function) return false;
    reinterpret_cast<void (*)(unsigned, unsigned, unsigned, const void *)>(k.function)(blocks, threads, smem_bytes, arg_struct);
    return true;
}

}  // namespace jit
}  // namespace srs
