"""The Rust shim (shim/) cannot be compiled in this image (no cargo / rustc): these checks keep it from drifting away from
include/sirius_amd.h mechanically.  CPU only."""
import os
import re
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys as G       # noqa: E402

SHIM = os.path.join(ROOT, "shim", "src")


def _rust_externs():
    """name -> number of parameters, from the extern "C" block of shim/src/sys.rs"""
    text = open(os.path.join(SHIM, "sys.rs")).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for name, args in re.findall(r"pub fn (srs_\w+)\((.*?)\)(?: -> [^;]+)?;", block):
        out[name] = 0 if not args.strip() else len(_split_top(args))
    return out


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def test_sys_rs_is_current():
    """shim/src/sys.rs == a fresh run of the generator on the header"""
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]).returncode == 0, \
        "shim/src/sys.rs is stale: run python tools/gen_rust_sys.py"


def test_extern_block_matches_header():
    """every function of the header is declared in the extern block with the same number of parameters and mapped types; the
    library exports each of them (the same names the ctypes loader binds)"""
    h = G.parse_header()
    rust = _rust_externs()
    names = [f[0] for f in h["functions"]]
    assert len(names) == len(set(names)) and len(names) >= 90
    assert set(rust) == set(names)
    for name, ret, args in h["functions"]:
        assert rust[name] == len(args), name
    from sirius_amd import _lib
    assert set(_lib._prototypes()) <= set(names), set(_lib._prototypes()) - set(names)
    # the type map on the patterns the header uses
    assert G.rust_type("const srs_fe *const *") == "*const *const srs_fe" and G.rust_type("srs_fe *const *") == "*const *mut srs_fe"
    assert G.rust_type("srs_ck **") == "*mut *mut srs_ck" and G.rust_type("const char *") == "*const c_char"
    assert G.rust_type("size_t") == "usize" and G.rust_type("void *") == "*mut c_void" and G.rust_type("uint32_t") == "u32"


def test_shim_modules_call_declared_functions_with_declared_arity():
    """every `srs_*(...)` call in the hand-written modules names a header function and passes as many arguments as it declares"""
    h = G.parse_header()
    arity = {f[0]: len(f[2]) for f in h["functions"]}
    consts = {k for k, _ in h["enums"]}
    seen = set()
    for fn in ("lib.rs", "commit.rs", "fft.rs", "sangria.rs", "protogalaxy.rs"):
        text = open(os.path.join(SHIM, fn)).read()
        text = re.sub(r"//[^\n]*", "", text)                        # comments (incl. the doc examples) are not code
        for m in re.finditer(r"\b(srs_[a-zA-Z0-9_]+)\s*\(", text):
            name = m.group(1)
            assert name in arity, (fn, name)
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            args = text[m.end(): i - 1]
            n = 0 if not args.strip() else len(_split_top(args))
            assert n == arity[name], (fn, name, n, arity[name])
            seen.add(name)
        for c in re.findall(r"\bSRS_[A-Z0-9_]+\b", text):
            assert c in consts, (fn, c)
    # the bodies INTEGRATION.md replaces are all reachable through the shim
    for must in ("srs_commit", "srs_commit_batch", "srs_commit_upload", "srs_commit_upload_columns", "srs_ck_create", "srs_ck_create_multi",
                 "srs_ntt", "srs_structure_create", "srs_commit_cross_terms", "srs_fold_witness", "srs_fold_error", "srs_point_lincomb",
                 "srs_pg_context_new", "srs_pg_compute_F", "srs_pg_compute_G", "srs_pg_compute_K_from_G", "srs_pg_evaluate_e",
                 "srs_fold_lincomb", "srs_layout_selftest", "srs_layout_selftest_point"):
        assert must in seen, must
