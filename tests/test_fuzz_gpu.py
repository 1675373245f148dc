"""A seeded slice of the randomised kernel-parity sweep (tests/fuzz.py) inside the `-m gpu` suite: 200 cases = 2400 kernel
checks against the fp64 restatements, every run the same shapes.  The long sweep (tools/fuzz_parity.py, other seeds) keeps
its log under profiles/."""
import pytest

from tests import fuzz

pytestmark = pytest.mark.gpu

CHUNKS, PER_CHUNK = 20, 10


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_random_shapes_match_the_fp64_restatement(gpu, chunk):
    fails = fuzz.run_cases(gpu, PER_CHUNK, seed=7000 + chunk)
    assert not fails, f"{len(fails)} of {12 * PER_CHUNK} random kernel checks failed; first: {fails[0]}"
