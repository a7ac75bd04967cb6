"""threefry2x32 + jax.random key plumbing (bpl-next_amd/csrc/threefry.hpp) -- host code,
no GPU.  Pinned by the Random123 known-answer vectors and by values jax publishes in its
own test-suite / docs (jax 0.4.24, non-partitionable threefry)."""
import ctypes as C

import numpy as np

import dc_oracle_c as OC
from bpl import _ffi


def test_random123_known_answers():
    """Threefry-2x32-20 known-answer vectors (Random123 kat_vectors; the same three are
    asserted by jax's own tests/random_test.py::testThreefry2x32): counter, key -> output."""
    h = OC.harness()
    kats = [
        ((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
        ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
        ((0x243F6A88, 0x85A308D3), (0x13198A2E, 0x03707344), (0xC4923A9C, 0x483DF7A0)),
    ]
    out = (C.c_uint32 * 2)()
    for ctr, key, exp in kats:
        h.harness_threefry_block(C.c_uint32(key[0]), C.c_uint32(key[1]), C.c_uint32(ctr[0]),
                                 C.c_uint32(ctr[1]), out)
        assert (int(out[0]), int(out[1])) == exp
    # the same block through the product ABI: bits(key, 1) hashes the single counter 0
    # padded with 0 -> block(key, (0, 0)), first word
    assert int(_ffi.threefry_bits((0, 0), 1)[0]) == 0x6B200159
    assert int(_ffi.threefry_bits((0xFFFFFFFF, 0xFFFFFFFF), 1)[0]) != 0x1CB996FC  # (counter 0, not ~0)


def test_jax_published_values():
    # jax tests/random_test.py: random_bits(PRNGKey(1701), 32, (3,))
    assert _ffi.threefry_bits(_ffi.prng_key(1701), 3).tolist() == [56197195, 4200222568, 961309823]
    # jax docs: random.split(PRNGKey(42)) -> [[2465931498 3679230171] [255383827 267815257]]
    assert _ffi.threefry_split(_ffi.prng_key(42), 2) == [(2465931498, 3679230171), (255383827, 267815257)]
    # PRNGKey(0) -> split -> well known subkeys
    assert _ffi.threefry_split(_ffi.prng_key(0), 2) == [(4146024105, 967050713), (2718843009, 1272950319)]


def test_normal_and_uniform_streams():
    h = OC.harness()
    out = np.zeros(5)
    h.harness_normal(C.c_uint32(0), C.c_uint32(0), 5, out.ctypes.data_as(C.c_void_p))
    # jax.random.normal(PRNGKey(0), (5,)) (float32)
    ref = np.array([0.18784384, -1.2833426, -0.2710917, 1.2490594, 0.24447003])
    assert np.abs(out - ref).max() < 2e-7
    u = np.zeros(100000)
    h.harness_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_float, C.c_void_p]
    h.harness_uniform(1, 2, u.size, -2.0, 2.0, u.ctypes.data_as(C.c_void_p))
    assert u.min() >= -2.0 and u.max() < 2.0
    assert abs(u.mean()) < 0.02 and abs(u.std() - 4 / np.sqrt(12)) < 0.01
    n = np.zeros(200000)
    h.harness_normal(C.c_uint32(3), C.c_uint32(4), n.size, n.ctypes.data_as(C.c_void_p))
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    assert np.isfinite(n).all()


def test_split_is_prefix_free_and_deterministic():
    a = _ffi.threefry_split((7, 9), 8)
    b = _ffi.threefry_split((7, 9), 8)
    assert a == b and len(set(a)) == 8
