// Is run-time compilation of a row program viable on the GPU box?  Compiles a kernel that pulls in field.cuh (+ the
// FIPS multiplier) with hiprtc, reports the compile time, loads and runs it.   g++ ... -lhiprtc -lamdhip64
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
static std::string slurp(const std::string &p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); }
int main(int argc, char **argv) {
    std::string root = argc > 1 ? argv[1] : ".";
    std::string field = slurp(root + "/sirius_amd/csrc/field.cuh"), fips = slurp(root + "/sirius_amd/csrc/field_fips.inc");
    if (field.empty() || fips.empty()) { printf("sources not found under %s\n", root.c_str()); return 1; }
    std::string src = "#include \"field.cuh\"\nusing namespace srs;\n"
                      "extern \"C\" __global__ void k(const fe_t *a, fe_t *o, int n) {\n"
                      "  int i = blockIdx.x * blockDim.x + threadIdx.x; fe_t x = a[i];\n"
                      "  for (int j = 0; j < n; ++j) x = Fr::add(Fr::mul(x, x), a[i]);\n  o[i] = x; }\n";
    hiprtcProgram prog;
    const char *hdr_src[2] = {field.c_str(), fips.c_str()};
    const char *hdr_name[2] = {"field.cuh", "field_fips.inc"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "rowprog_jit.hip", 2, hdr_src, hdr_name) != HIPRTC_SUCCESS) { printf("create failed\n"); return 1; }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-I/opt/rocm/include"};
    auto t0 = std::chrono::steady_clock::now();
    hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); printf("log: %.2000s\n", log.c_str()); }
    printf("hiprtcCompileProgram rc=%d in %.2f s\n", (int)rc, dt);
    if (rc != HIPRTC_SUCCESS) return 1;
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
    hipModule_t mod; hipFunction_t fn;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "k") != hipSuccess) { printf("load failed\n"); return 1; }
    printf("code object %zu bytes, loaded\n", cs);
    return 0;
}
