"""ctypes loader for libsirius_amd.so (the C-ABI of include/sirius_amd.h).

There is no CPU implementation behind this package: if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"`) importing a compute entry raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsirius_amd.so")

OK, ERR_TOO_LONG_INPUT, ERR_NOT_POW2, ERR_K_TOO_LARGE, ERR_INVALID, ERR_DEVICE, ERR_LAYOUT, ERR_EVAL_INDEX, ERR_IO, ERR_INVALID_DATA, ERR_UNSUPPORTED = range(11)
CURVE_BN256, CURVE_GRUMPKIN = 0, 1
FIELD_FR, FIELD_FQ = 0, 1
SPACE_HOST, SPACE_DEVICE = 0, 1
REPR_MONT, REPR_CANON = 0, 1

_lib = None


class SiriusAmdError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__(f"sirius_amd rc={rc}: {msg}")
        self.rc = rc


def _prototypes():
    vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
    return {
        "srs_init": (i32, [i32]),
        "srs_init_thread": (i32, [i32]),
        "srs_last_error": (C.c_char_p, []),
        "srs_version": (C.c_char_p, []),
        "srs_tuning_set": (i32, [C.c_char_p, C.c_int64]),
        "srs_tuning_get": (i32, [C.c_char_p, C.POINTER(C.c_int64)]),
        "srs_tuning_reset": (None, []),
        "srs_tuning_name": (C.c_char_p, [i32]),
        "srs_scalar_field_of": (i32, [i32]),
        "srs_layout_selftest": (i32, [i32, vp, vp]),
        "srs_layout_selftest_point": (i32, [i32, vp]),
        "srs_dev_alloc": (i32, [sz, C.POINTER(vp)]),
        "srs_dev_free": (None, [vp]),
        "srs_host_alloc": (i32, [sz, C.POINTER(vp)]),
        "srs_host_free": (None, [vp]),
        "srs_upload": (i32, [vp, vp, sz, vp]),
        "srs_download": (i32, [vp, vp, sz, vp]),
        "srs_commit_upload": (i32, [vp, vp, sz, vp, i32, vp, vp]),
        "srs_concat_len": (sz, [C.POINTER(sz), sz, sz]),
        "srs_concat_with_padding": (i32, [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, vp]),
        "srs_commit_upload_columns": (i32, [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, vp, i32, vp, vp]),
        "srs_ck_create_multi": (i32, [i32, vp, sz, i32, i32, C.POINTER(vp)]),
        "srs_ck_setup_synthetic_multi": (i32, [i32, sz, C.c_uint64, i32, C.POINTER(vp)]),
        "srs_ck_num_shards": (i32, [vp]),
        "srs_ck_msm_stats": (i32, [vp, C.POINTER(C.c_uint64)]),
        "srs_ck_has_wide_table": (i32, [vp]),
        "srs_ck_shard_stats": (i32, [vp, i32, C.POINTER(C.c_uint64)]),
        "srs_structure_kernel_kind": (i32, [vp, i32]),
        "srs_ck_create": (i32, [i32, vp, sz, i32, C.POINTER(vp)]),
        "srs_ck_create_sharded": (i32, [i32, vp, sz, i32, u32, u32, C.POINTER(vp)]),
        "srs_ck_setup_synthetic": (i32, [i32, sz, C.c_uint64, u32, u32, C.POINTER(vp)]),
        "srs_ck_setup_uniform_bytes": (i32, [C.c_char_p, sz, sz, sz, vp]),
        "srs_ck_setup": (i32, [i32, u32, C.c_char_p, sz, C.POINTER(vp)]),
        "srs_ck_get_bases": (i32, [vp, vp]),
        "srs_ck_local_len": (sz, [vp]),
        "srs_point_lincomb": (i32, [i32, vp, vp, vp, sz, i32, vp]),
        "srs_point_lincomb_async": (i32, [i32, vp, vp, vp, sz, i32, vp, C.POINTER(C.c_uint64)]),
        "srs_job_wait": (i32, [C.c_uint64]),
        "srs_fe_powers": (i32, [i32, vp, sz, vp]),
        "srs_jit_selfcheck": (i32, [C.POINTER(sz), C.c_char_p, sz]),
        "srs_profile_enable": (None, [i32]),
        "srs_profile_reset": (None, []),
        "srs_profile_sampling": (None, [C.c_uint]),
        "srs_profile_get": (i32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "srs_ck_load_file": (i32, [i32, C.c_char_p, sz, u32, u32, C.POINTER(vp)]),
        "srs_ck_save_file": (i32, [vp, C.c_char_p]),
        "srs_ck_count_off_curve": (i32, [vp, C.POINTER(sz)]),
        "srs_is_sat_gates": (i32, [vp, i32, vp, vp, sz, vp, i32, vp, C.POINTER(sz)]),
        "srs_ck_free": (None, [vp]),
        "srs_ck_len": (sz, [vp]),
        "srs_commit": (i32, [vp, vp, sz, i32, i32, vp, vp]),
        "srs_commit_batch": (i32, [vp, C.POINTER(vp), C.POINTER(sz), sz, i32, i32, vp, vp]),
        "srs_point_sum": (i32, [i32, vp, sz, vp]),
        "srs_point_mul": (i32, [i32, vp, i32, vp, vp]),
        "srs_structure_create": (i32, [i32, u32, sz, sz, sz, C.POINTER(vp), C.POINTER(vp), i32, vp, sz, sz, C.POINTER(vp)]),
        "srs_structure_create_lookup": (i32, [i32, u32, sz, sz, sz, C.POINTER(vp), C.POINTER(vp), i32, vp, sz, sz, sz, i32, vp, sz,
                                              C.POINTER(vp)]),
        "srs_structure_num_witness_columns": (sz, [vp]),
        "srs_lookup_coeff_1": (i32, [vp, vp, vp, i32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "srs_lookup_coeff_2": (i32, [i32, vp, vp, vp, vp, sz, i32, vp, vp, vp]),
        "srs_is_sat_log_derivative": (i32, [vp, vp, i32, vp, C.POINTER(sz)]),
        "srs_sparse_create": (i32, [i32, sz, vp, vp, vp, sz, C.POINTER(vp)]),
        "srs_sparse_free": (None, [vp]),
        "srs_sparse_matvec": (i32, [vp, vp, i32, vp, vp]),
        "srs_is_sat_permutation": (i32, [vp, vp, i32, vp, C.POINTER(sz)]),
        "srs_is_sat_witness_commit": (i32, [vp, C.POINTER(vp), C.POINTER(sz), sz, vp, vp, sz, vp, i32, vp, C.POINTER(sz), C.POINTER(i32)]),
        "srs_structure_set_shard": (i32, [vp, u32, u32]),
        "srs_structure_upload_shard_halo": (i32, [vp, vp, vp, sz, i32, vp]),
        "srs_poseidon_new": (i32, [i32, sz, sz, sz, sz, C.POINTER(vp)]),
        "srs_poseidon_free": (None, [vp]),
        "srs_poseidon_reset": (None, [vp]),
        "srs_poseidon_absorb_field": (i32, [vp, vp, sz]),
        "srs_poseidon_absorb_point": (i32, [vp, i32, vp]),
        "srs_poseidon_squeeze": (i32, [vp, sz, i32, vp]),
        "srs_poseidon_squeeze_device": (i32, [vp, sz, i32, vp, C.POINTER(C.c_double)]),
        "srs_batch_invert_assigned": (i32, [i32, vp, vp, vp, sz, i32, vp, vp]),
        "srs_structure_free": (None, [vp]),
        "srs_structure_num_cross_terms": (sz, [vp]),
        "srs_structure_num_challenges": (sz, [vp]),
        "srs_structure_program_source": (sz, [vp, i32, vp, sz, C.POINTER(C.c_uint64), C.POINTER(i32)]),
        "srs_cross_terms": (i32, [vp, vp, vp, vp, sz, i32, vp, C.POINTER(vp)]),
        "srs_commit_cross_terms": (i32, [vp, vp, vp, vp, vp, sz, i32, vp, C.POINTER(vp), vp]),
        "srs_eval_gates": (i32, [vp, i32, vp, vp, sz, i32, vp, vp]),
        "srs_fold_witness": (i32, [i32, vp, vp, vp, vp, sz, i32, vp]),
        "srs_fold_error": (i32, [i32, vp, vp, C.POINTER(vp), sz, vp, sz, i32, vp]),
        "srs_pg_context_new": (i32, [vp, sz, vp]),
        "srs_pg_beta_stroke": (i32, [vp, sz, vp, vp, vp]),
        "srs_pg_prove": (i32, [vp, vp, vp, sz, vp, C.POINTER(vp), C.POINTER(vp), sz, sz, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "srs_sangria_prove": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp, C.POINTER(vp), vp, vp, vp, vp, C.POINTER(C.c_uint64)]),
        "srs_sangria_prove_incoming": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, vp, vp, vp, C.POINTER(vp), vp, vp, vp, vp,
                                             C.POINTER(C.c_uint64)]),
        "srs_pg_compute_F": (i32, [vp, vp, sz, vp, vp, vp, sz, i32, i32, vp, vp]),
        "srs_pg_compute_G": (i32, [vp, vp, sz, C.POINTER(vp), C.POINTER(vp), sz, sz, i32, i32, vp, vp]),
        "srs_pg_compute_K_from_G": (i32, [vp, sz, vp, sz, u32, vp, vp]),
        "srs_pg_evaluate_e": (i32, [vp, vp, sz, vp, vp, sz, i32, i32, vp, vp]),
        "srs_pg_calculate_e": (i32, [vp, sz, vp, sz, vp, vp, u32, vp]),
        "srs_lagrange_eval": (i32, [vp, u32, vp]),
        "srs_poly_eval": (i32, [vp, sz, vp, vp]),
        "srs_fold_lincomb": (i32, [i32, vp, C.POINTER(vp), vp, sz, sz, i32, vp]),
        "srs_fold_lincomb_sharded": (i32, [i32, vp, C.POINTER(vp), vp, sz, sz, u32, u32, vp]),
        "srs_structure_fold_sharded": (i32, [vp, vp, vp, vp, sz, i32, vp]),
        "srs_ntt": (i32, [i32, vp, sz, i32, i32, i32, vp]),
        "srs_ntt_set_max_radix_bits": (i32, [i32]),
        "srs_ntt_batch": (i32, [i32, vp, sz, sz, sz, i32, i32, i32, vp]),
    }


def load(path=None):
    """Load the shared library (idempotent).  `path` is a test hook for tests/emu."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SRS_AMD_LIB") or LIB_PATH      # SRS_AMD_LIB: developer switch (A/B runs of library builds)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build the HIP extension first (__graft_entry__.build()); "
            "sirius_amd has no CPU fallback")
    if path is None:
        try:  # share torch's HIP runtime when torch is in the process (same libamdhip64 soname)
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(p)
    for name, (res, args) in _prototypes().items():
        fn = getattr(lib, name)      # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # test hook: SRS_TEST_TUNING="msm_sort=2,commit_chunks=3" applies srs_tuning_set after loading (the tests' subprocesses; the library
    # itself reads no tunable from the environment)
    for item in filter(None, os.environ.get("SRS_TEST_TUNING", "").split(",")):
        name, _, value = item.partition("=")
        if lib.srs_tuning_set(name.strip().encode(), int(value)) != 0:
            raise ValueError(f"SRS_TEST_TUNING: unknown tunable {name!r}")
    return lib


def lib():
    return load()


def check(rc):
    if rc != OK:
        raise SiriusAmdError(rc, (lib().srs_last_error() or b"").decode())
