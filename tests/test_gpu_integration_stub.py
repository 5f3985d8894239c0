"""The ctypes stub printed in INTEGRATION.md is executed verbatim (only the library path is
made absolute) and must reproduce the reference's matrices."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_runs(tmp_path, golden_dir):
    import phyloligo_amd  # noqa: F401  (loads torch first so the process holds one HIP runtime)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes.*?)```", text, re.S).group(1)
    code = code.replace('"libphyloligo_amd.so"', repr(os.path.join(ROOT, "phyloligo_amd", "libphyloligo_amd.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = np.load(os.path.join(golden_dir, "distances.npz"))
    fasta = tmp_path / "a.fa"
    with open(fasta, "wb") as fh:
        for i, s in enumerate(g["contigs"]):
            fh.write(b">r%d\n" % i + bytes(s) + b"\n")
    freq = ns["compute_frequencies_hip"](str(fasta), "1111", "both")
    assert np.array_equal(freq, g["freq_1111_both"])
    for metric in ("Eucl", "JSD", "BC"):
        got = ns["compute_distances_hip"](freq, metric)
        np.testing.assert_allclose(got, g["%s_1111_both" % metric], rtol=1e-6, atol=1e-12, equal_nan=True)
