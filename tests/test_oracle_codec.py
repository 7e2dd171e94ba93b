"""Oracle codec vs the reference's unit-test properties and hand-derived known answers.

Reference tests mirrored: crates/simd/src/bitpacking_u32_ordered.rs:239-259,
bitpacking_u32_unordered.rs:190-208, bytepacking_u32_ordered.rs:216-239,
bytepacking_u32_unordered.rs:202-223 (round trips for every width / length).
The reference has no fixed byte vectors; the known answers below are derived by hand
from the compress! macro (crates/simd/src/bitpacking.rs:15-54) -- SURVEY section 8(c)(iii).
"""
import numpy as np
import pytest

import orc


@pytest.mark.parametrize("bits", range(0, 33))
def test_bitpacking_ordered_roundtrip(bits):
    rng = np.random.default_rng(bits)
    for _ in range(8):
        hi = (1 << bits) if bits < 32 else (1 << 32)
        data = np.sort(rng.integers(0, hi, 128, dtype=np.uint64)).astype(np.uint32)
        min_doc = int(data[0])
        meta, payload = orc.compress_doc_ids(min_doc, data)
        assert meta >> 7 == 0 and (meta & 127) <= bits
        assert len(payload) == 16 * (meta & 127)
        if meta & 127 == 0:
            # bitwidth 0: decode is a no-op that leaves stale buffer contents
            out = orc.decompress_doc_ids(min_doc, meta, payload, stale=np.arange(128))
            assert np.array_equal(out, np.arange(128))
            continue
        out = orc.decompress_doc_ids(min_doc, meta, payload)
        assert np.array_equal(out, data)


@pytest.mark.parametrize("bits", range(0, 33))
def test_bitpacking_unordered_roundtrip(bits):
    rng = np.random.default_rng(100 + bits)
    hi = (1 << bits) if bits < 32 else (1 << 32)
    data = rng.integers(0, hi, 128, dtype=np.uint64).astype(np.uint32)
    meta, payload = orc.compress_tfs(data)
    assert meta >> 7 == 0 and (meta & 127) <= bits
    if meta & 127:
        assert np.array_equal(orc.decompress_tfs(meta, payload), data)


@pytest.mark.parametrize("width", range(0, 5))
def test_bytepacking_roundtrip_every_len(width):
    rng = np.random.default_rng(200 + width)
    hi = (1 << (8 * width)) if width < 4 else (1 << 32)
    base = np.sort(rng.integers(0, max(hi, 1), 128, dtype=np.uint64)).astype(np.uint32)
    for n in range(1, 128):
        data = base[:n]
        meta, payload = orc.compress_doc_ids(int(data[0]), data)
        assert meta >> 7 == 1 and 1 <= (meta & 127) <= max(width, 1)
        assert len(payload) == (meta & 127) * n
        assert np.array_equal(orc.decompress_doc_ids(int(data[0]), meta, payload), data)
        tf = rng.permutation(data)
        meta, payload = orc.compress_tfs(tf)
        assert meta >> 7 == 1 and len(payload) == (meta & 127) * n
        assert np.array_equal(orc.decompress_tfs(meta, payload), tf)


def test_kat_b1_layout():
    # b = 1: bit t of the 32-bit word of lane l (bytes 4l..4l+3) is value 4t + l
    tf = np.zeros(128, dtype=np.uint32)
    tf[[0, 5, 127]] = 1  # (l=0,t=0) (l=1,t=1) (l=3,t=31)
    meta, payload = orc.compress_tfs(tf)
    assert meta == 1 and len(payload) == 16
    words = payload.view("<u4")
    assert words.tolist() == [1, 2, 0, 1 << 31]


def test_kat_b3_straddle():
    # SURVEY appendix A worked example: b=3, value index 42 = lane 2, step 10 -> bits [30,33)
    tf = np.zeros(128, dtype=np.uint32)
    tf[10] = 0b101  # lane 2, step 2 -> bits [6,9) of word 0 of lane 2
    tf[42] = 0b111  # low 2 bits in word 0 bits 30-31, high bit in word 1 bit 0
    meta, payload = orc.compress_tfs(tf)
    assert meta == 3 and len(payload) == 48
    w = payload.view("<u4").reshape(3, 4)  # [word][lane]
    assert w[0, 2] == (0b101 << 6) | (0b11 << 30)
    assert w[1, 2] == 1
    assert w[:, [0, 1, 3]].sum() == 0 and w[2, 2] == 0
    assert np.array_equal(orc.decompress_tfs(meta, payload), tf)


def test_kat_delta_chain_with_min_doc():
    # doc ids = min + cumulative deltas in INDEX order (field 0 is always 0)
    ids = (1000 + np.cumsum(np.r_[0, np.full(127, 5)])).astype(np.uint32)
    meta, payload = orc.compress_doc_ids(1000, ids)
    assert meta == 3  # delta 5 needs 3 bits
    w = payload.view("<u4").reshape(3, 4)
    # lane 0 stream: fields [0,5,5,...]; lanes 1..3: all 5
    lane0 = sum(5 << (3 * t) for t in range(1, 32))
    lane1 = sum(5 << (3 * t) for t in range(0, 32))
    for wi in range(3):
        assert w[wi, 0] == (lane0 >> (32 * wi)) & 0xffffffff
        assert w[wi, 1] == w[wi, 2] == w[wi, 3] == (lane1 >> (32 * wi)) & 0xffffffff
    assert np.array_equal(orc.decompress_doc_ids(1000, meta, payload), ids)


def test_kat_b32_raw_absolute():
    # one delta >= 2^31 forces bitwidth 32: payload = raw absolute ids, no delta
    # (bitpacking_u32_ordered.rs:119-121)
    ids = np.r_[np.arange(64), (1 << 31) + 100 + 3 * np.arange(64)].astype(np.uint32)
    meta, payload = orc.compress_doc_ids(int(ids[0]), ids)
    assert meta == 32 and len(payload) == 512
    assert np.array_equal(payload.view("<u4"), ids)
    assert np.array_equal(orc.decompress_doc_ids(0, meta, payload), ids)


def test_kat_bytepacked_w3_n5():
    ids = np.array([7, 7 + 0x010203, 7 + 0x010203 + 1, 0x900000, 0x900000 + 0xffffff],
                   dtype=np.uint32)
    meta, payload = orc.compress_doc_ids(7, ids)
    assert meta == 0x83 and len(payload) == 15
    assert payload[:9].tolist() == [0, 0, 0, 3, 2, 1, 1, 0, 0]
    assert np.array_equal(orc.decompress_doc_ids(7, meta, payload), ids)
    # bytewidth 4: raw absolute, no delta (bytepacking_u32_ordered.rs:195,211)
    ids4 = np.array([5, 0x01000005 + 5], dtype=np.uint32)
    meta, payload = orc.compress_doc_ids(5, ids4)
    assert meta == 0x84 and np.array_equal(payload.view("<u4"), ids4)
    # tf bytewidth is at least 1 even for all-zero input (bytepacking_u32_unordered.rs:17-28)
    meta, payload = orc.compress_tfs(np.zeros(3, dtype=np.uint32))
    assert meta == 0x81 and len(payload) == 3
