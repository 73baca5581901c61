"""SURVEY §5 "race detection / sanitizers": the product's host code (scene loader, dump, PNG, checkpoints) and the oracle
under AddressSanitizer + UndefinedBehaviorSanitizer, CPU builds only (tools/sanitizers/Makefile; GPU sanitizers are not
available on this pool). Runs when `-fsanitize=address` links in this environment."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_links():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write("int main() { return 0; }\n")
        r = subprocess.run(["g++", "-fsanitize=address,undefined", src, "-o", os.path.join(d, "t")], capture_output=True)
        return r.returncode == 0 and subprocess.run([os.path.join(d, "t")]).returncode == 0


def test_host_code_and_oracle_are_clean_under_asan_and_ubsan(tmp_path):
    if not _asan_links():
        pytest.skip("-fsanitize=address does not link here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "sanitizers"), "run", "N=3000", f"OUT={tmp_path}"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "clean" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
