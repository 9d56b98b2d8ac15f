"""A bounded, seeded slice of the randomised op fuzzing (tests/fuzz_gpu.py) under `pytest -m gpu`: every op family draws random shapes
and degenerate clouds (lattice points, duplicates, planar) for a few seconds and is compared with the CPU oracle (index outputs and
gathers bit-exact) or an fp64 torch statement (MFMA ops, 2e-5 relative)."""
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["fps", "knn", "3nn", "knn_grid", "3nn_grid", "gather", "backward", "linear", "attention", "chain_sa", "chain_fp", "netvlad", "afa", "linear_lds", "train_glue", "sa_mid", "fpx16", "attention_f16", "fpx3"]


@pytest.mark.parametrize("family", NAMES)
def test_fuzz_family(family, capsys):
    from tests import fuzz_gpu
    fams = dict(fuzz_gpu.FAMILIES)
    assert sorted(fams) == sorted(NAMES)
    fuzz_gpu.reseed(2026 + NAMES.index(family))
    ok = fuzz_gpu.run(family, fams[family], budget=4.0)
    out = capsys.readouterr().out
    assert ok, out
