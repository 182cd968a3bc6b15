"""Repository contract: the product never touches the oracle or the reference checkout."""
import os
import re

from conftest import ROOT


def _py_files(top):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_package_never_imports_oracle_or_reads_reference():
    for path in _py_files("stereoscene_amd"):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), path
        assert "/root/reference" not in src, path


def test_bench_only_uses_oracle_in_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle", src)]
    a, b = src.index("def cpu_baseline"), src.index("def main")
    assert uses and all(a < u < b for u in uses)
    assert "/root/reference" not in src


def test_required_files_exist():
    for f in ("DESIGN.md", "INTEGRATION.md", "include/ssbev.h", "bench.py", "__graft_entry__.py", "oracle/path_ref.py",
              "oracle/make_golden.py", "tests/golden/vt_small.npz", "profiles"):
        assert os.path.exists(os.path.join(ROOT, f)), f
