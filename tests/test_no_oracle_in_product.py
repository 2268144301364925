"""The oracle is test infrastructure: nothing under hdrnet_amd/ (the product) may import,
load or execute anything under oracle/, nor fall back to a CPU implementation."""
import os
import re

from conftest import ROOT


def product_files():
    for d, _, files in os.walk(os.path.join(ROOT, "hdrnet_amd")):
        if "lib" in os.path.relpath(d, ROOT).split(os.sep)[1:2]:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle/|liboracle|libhdrnet_ref|oracle\.)")
    bad = []
    for path in product_files():
        for n, line in enumerate(open(path, errors="replace"), 1):
            if pat.search(line):
                bad.append(f"{os.path.relpath(path, ROOT)}:{n}: {line.strip()}")
    assert not bad, "\n".join(bad)


def test_product_has_no_compat_layers():
    pat = re.compile(r"(__HIP_PLATFORM_AMD__|__CUDACC__|cuda_runtime\.h|import\s+triton|hipify)")
    bad = []
    for path in product_files():
        for n, line in enumerate(open(path, errors="replace"), 1):
            if pat.search(line):
                bad.append(f"{os.path.relpath(path, ROOT)}:{n}: {line.strip()}")
    assert not bad, "\n".join(bad)
