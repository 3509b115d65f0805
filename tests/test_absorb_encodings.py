"""sponge/absorb.rs encodings (host side): the reference's own tests (absorb.rs:393-496) restated, plus the rules each
impl states.  No GPU needed; the sponge part is in tests/test_gpu_features.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_primitives_amd.sponge import absorb as ab  # noqa: E402

P = ab.P


def test_string_absorb_is_length_prefixed():  # absorb.rs:474-496 test_string_absort
    s1, s2, s3 = ab.Str("hello"), ab.Str("world"), ab.Str("helloworld")
    assert ab.collect_sponge_bytes(s1, s2) != s3.to_sponge_bytes()
    assert ab.collect_sponge_field_elements(s1, s2) != s3.to_sponge_field_elements()
    assert s1.to_sponge_bytes() == (5).to_bytes(8, "little") + b"hello"


def test_integer_rules():
    assert ab.U8(200).to_sponge_field_elements() == [200] and ab.U8(200).to_sponge_bytes() == b"\xc8"
    assert ab.U16(0x1234).to_sponge_bytes() == b"\x34\x12" and ab.U32(7).to_sponge_bytes() == b"\x07\0\0\0"
    assert ab.U128((1 << 128) - 1).to_sponge_field_elements() == [(1 << 128) - 1]
    assert ab.Usize(9).to_sponge_bytes() == (9).to_bytes(8, "little")
    assert ab.I16(-2).to_sponge_field_elements() == [P - 2] and ab.I16(-2).to_sponge_bytes() == b"\xfe\xff"
    assert ab.I64(5).to_sponge_field_elements() == [5]
    assert ab.Bool(True).to_sponge_field_elements() == [1] and ab.Bool(False).to_sponge_bytes() == b"\0"
    with pytest.raises(OverflowError):
        ab.U8(256)
    with pytest.raises(OverflowError):
        ab.I8(128)


def test_byte_slices_use_the_u8_batch_rule():  # absorb.rs:133-142
    b = bytes(range(70))
    el = ab.Bytes(b).to_sponge_field_elements()
    raw = (70).to_bytes(8, "little") + b
    assert el == [int.from_bytes(raw[i:i + 31], "little") for i in range(0, 78, 31)] and len(el) == 3
    assert ab.Bytes(b).to_sponge_bytes() == b
    assert ab.Bytes(b"\0").to_sponge_field_elements() != ab.Bytes(b"\0\0").to_sponge_field_elements()  # the length prefix
    assert ab.Str("abc").to_sponge_field_elements() == ab.Bytes(b"abc").to_sponge_field_elements()
    with pytest.raises(AssertionError):
        ab.Seq([ab.U8(1), ab.U8(2)])  # a slice of u8 is `Bytes`


def test_sequences_options_points_structs():
    seq = ab.Seq([ab.U16(1), ab.U16(2), ab.U16(3)])
    assert seq.to_sponge_field_elements() == [1, 2, 3] and seq.to_sponge_bytes() == b"\1\0\2\0\3\0"  # no length prefix (:41-80)
    assert ab.WithLength(seq).to_sponge_field_elements() == [3, 1, 2, 3]
    assert ab.WithLength(ab.Bytes(b"xy")).to_sponge_bytes() == (2).to_bytes(8, "little") + b"xy"
    assert ab.Opt(ab.U32(9)).to_sponge_field_elements() == [1, 9] and ab.Opt().to_sponge_field_elements() == [0]
    assert ab.Opt(ab.U32(9)).to_sponge_bytes() == b"\1\x09\0\0\0"
    pt = ab.TEAffine(5, P - 1)
    assert pt.to_sponge_field_elements() == [5, P - 1]
    assert pt.to_sponge_bytes() == (5).to_bytes(32, "little") + (P - 1).to_bytes(32, "little")
    assert ab.Fe(P + 3).to_sponge_field_elements() == [3]  # field_cast works on reduced elements
    st = ab.Struct(ab.U8(1), ab.U16(2), ab.Struct(ab.U8(7), ab.U16(8)), ab.Fe(6))
    assert st.to_sponge_field_elements() == [1, 2, 7, 8, 6]
    assert st.to_sponge_bytes() == b"\1" + b"\2\0" + b"\7" + b"\x08\0" + (6).to_bytes(32, "little")
