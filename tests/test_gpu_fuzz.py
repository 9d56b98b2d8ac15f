"""A bounded, seeded slice of the randomised op fuzzing (tests/fuzz_gpu.py) under `pytest -m gpu`: every op family draws random shapes
and degenerate clouds (lattice points, duplicates, planar) for a few seconds and is compared with the CPU oracle (index outputs and
gathers bit-exact) or an fp64 torch statement (MFMA ops, 2e-5 relative)."""
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["fps", "knn", "3nn", "knn_grid", "3nn_grid", "gather", "backward", "linear", "attention", "chain_sa", "chain_fp", "netvlad", "afa", "linear_lds", "train_glue", "sa_mid", "sa_tiny", "fpx32", "fpx16", "attention_f16", "fpx3"]


@pytest.mark.parametrize("family", NAMES)
def test_fuzz_family(family, capsys):
    from tests import fuzz_gpu
    fams = dict(fuzz_gpu.FAMILIES)
    assert sorted(fams) == sorted(NAMES)
    fuzz_gpu.reseed(2026 + NAMES.index(family))
    ok = fuzz_gpu.run(family, fams[family], budget=4.0)
    out = capsys.readouterr().out
    assert ok, out


def test_fuzz_training_gemm_on_lds_resident_weights():
    """tools/fuzz_tgemm_cm.py, a seeded slice: random (batch, M, N, K) with every operand transform, both layouts of A, bias, beta, statistics and
    the fused BatchNorm-backward sums through pa_tgemm_nn / pa_tgemm_nn_bnred on the LDS-resident-weights kernel, twice each, against the
    LDS-tiled kernel.  (The fixed shapes of test_gpu_train_ops.py did not contain the one that exposed the store-data hazard of round 5.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_tgemm_cm.py"), "60", "11"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "60 cases, 0 mismatches" in r.stdout, r.stdout[-2000:]
