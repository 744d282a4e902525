"""oracle/philox.py against the Random123 known-answer vectors (CPU)."""
import numpy as np

from oracle import philox as P


def test_philox4x32_10_known_answers():
    for counter, key, expect in P.KAT:
        got = P.philox4x32_10(np.array(counter, np.uint32), key)
        assert [int(x) for x in got] == list(expect)


def test_philox_is_vectorised_consistently():
    rng = np.random.default_rng(0)
    c = rng.integers(0, 2 ** 32, size=(64, 4), dtype=np.uint64).astype(np.uint32)
    whole = P.philox4x32_10(c, (123, 456))
    for i in (0, 17, 63):
        assert np.array_equal(whole[i], P.philox4x32_10(c[i], (123, 456)))


def test_unit_noise_is_standard_normal_and_addressed_by_row_and_offset():
    from scipy import stats as ss
    z = P.unit_noise(8, 40000, seed=11, offset=3)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert ss.kstest(z[0], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[0], z[1])[0, 1]) < 2e-2                      # rows: independent draws
    assert abs(np.corrcoef(z[0], P.unit_noise(1, 40000, 11, 4)[0])[0, 1]) < 2e-2   # offsets too
    # ragged length: the tail of the last block is dropped, nothing else moves
    assert np.array_equal(P.unit_noise(2, 39999, 11, 3), z[:2, :39999])
