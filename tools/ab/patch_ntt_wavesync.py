"""A/B build: the workgroup barrier between radix-4 groups 0 and 1 of an NTT tile becomes a WAVE-level ordering (both groups of a thread slot
touch the same 32-row block of the tile, and a wavefront owns that block in both: the dependency is intra-wavefront).
usage: python tools/ab/patch_ntt_wavesync.py sirius_amd/csrc/ntt.hip /tmp/ntt_ws.hip && python tools/build_variant.py ntt_ws ntt.hip @/tmp/ntt_ws.hip"""
import sys
s = open(sys.argv[1]).read()
old = """    lazy_stage2<NCOLS, 0>(tile, W, RBITS);
    __syncthreads();
    lazy_stage2<NCOLS, 1>(tile, W, RBITS);"""
new = """    lazy_stage2<NCOLS, 0>(tile, W, RBITS);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    lazy_stage2<NCOLS, 1>(tile, W, RBITS);"""
assert s.count(old) == 1
open(sys.argv[2], "w").write(s.replace(old, new))
