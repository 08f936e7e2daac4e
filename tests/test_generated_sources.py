"""The generated device sources under sirius_amd/csrc are what their generators emit today (no hand edits, no stale output).
rowprog_spec.inc is covered where it is used (tests/test_emu_jit.py drives the emitter it comes from)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,inc", [("gen_field_fips.py", "field_fips.inc"), ("gen_field29_chain.py", "field29_chain.inc")])
def test_generated_inc_is_current(tmp_path, gen, inc):
    out = tmp_path / inc
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", gen), str(out)], stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "sirius_amd", "csrc", inc)).read(), f"{inc} is not what tools/{gen} generates"
