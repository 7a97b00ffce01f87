"""The reference-side binding shown in INTEGRATION.md (integration/GpuMixtureScorer.hpp) must compile against
the reference's own headers.  Build container only: /root/reference does not exist on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/sietill"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "FeatureScorer.hpp")), reason="reference sources not present")
def test_reference_side_stub_compiles_against_reference_headers(tmp_path):
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "GpuMixtureScorer.hpp"\nint main() { return 0; }\n')
    cmd = ["g++", "--std=c++11", "-fsyntax-only", "-Wall", "-include", "emmintrin.h", "-I" + REF,
           "-I" + os.path.join(REF, "rapidjson", "include"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "integration"), str(tu)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
