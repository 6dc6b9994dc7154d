"""Route A as a test: the unmodified reference GeoFormer, imported from /root/reference,
running over geoformer_amd.dropin's spconv / PG_OP / pointnet2._ext / faiss modules and the reference's own wrapper
files, must reproduce the committed fixture of the reference's forward.  Needs the reference checkout (build
container only; skipped on the GPU box) and its own process (the reference parses sys.argv at import)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference checkout not present")
def test_reference_model_runs_on_the_dropin_modules_and_reproduces_its_golden():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "route_a_check.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok ") == 12 and "FAIL" not in r.stdout
