"""csrc/blake3.cpp against the published BLAKE3 test vectors (test_vectors.json of the BLAKE3 repository:
input byte i = i % 251, key = "whats the Elvish word for friend"; first 32 output bytes) and intern's long
path (vector.rs:19-35) built on it.  No GPU use."""
import ctypes as C

import vectorchord_bm25_amd as vb

KEY = b"whats the Elvish word for friend"
HASH = {
    0: "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
    1: "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
    1024: "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
    1025: "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
    2048: "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
    2049: "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030",
    3072: "b98cb0ff3623be03326b373de6b9095218513e64f1ee2edd2525c7ad1e5cffd2",
    31744: "62b6960e1a44bcc1eb1a611a8d6235b6b4b78f32e7abc4fb4c6cdcce94895c47",
}
KEYED = {
    0: "92b2b75604ed3c761f9d6f62392c8a9227ad0ea3f09573e783f1498a4ed60d26",
    1: "6d7878dfff2f485635d39013278ae14f1454b8c0a3a2d34bc1ab38228a80c95b",
}


def _hash(data, key=None):
    L = C.CDLL(vb.library_path())
    out = (C.c_uint8 * 32)()
    assert L.vbm25_blake3(key, data, C.c_size_t(len(data)), out) == 0
    return bytes(out)


def _input(n):
    return bytes(i % 251 for i in range(n))


def test_published_vectors():
    for n, want in HASH.items():
        assert _hash(_input(n)).hex() == want, n
    for n, want in KEYED.items():
        assert _hash(_input(n), KEY).hex() == want, n


def test_intern_long_lexeme_is_the_keyed_hash_prefix():
    for lex in (b"x" * 16, b"internationalization", b"nul\0inside", _input(3000)):
        h = bytearray(_hash(lex, KEY)[:16])
        if h[15] == 0:
            h[15] = 1
        assert vb.intern(lex, KEY) == bytes(h)
