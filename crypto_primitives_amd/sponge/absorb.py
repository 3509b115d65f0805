"""The `Absorb` encodings of sponge/absorb.rs (host side): what `CryptographicSponge::absorb(&impl Absorb)` turns a value into
before the duplex sponge sees it.  Python has no `u16` or `Option<T>`, so a value is wrapped in a small typed object; each
wrapper restates ONE impl of the reference:

    U8, Bool                         absorb.rs:124-152      one element F::from(v); byte = v
    U16, U32, U64, U128, Usize       :169-186, 212-220      one element; bytes = little-endian of the type's width (usize as u64)
    I8 ... I128, Isize               :188-210, 222-230      element = +-F::from(|v|); bytes = little-endian two's complement
    Fe                               :154-167               a field element of the sponge's field (field_cast identity, :108-122);
                                                            bytes = serialize_compressed = 32 bytes little-endian canonical
    Bytes  (&[u8] / Vec<u8>)         :133-142 batch rule    elements = to_field_elements(u64_le(len) || bytes): 31-byte chunks;
                                                            bytes = the bytes themselves
    Str    (String)                  :232-241               elements as Bytes; bytes = usize_le(len) || utf-8
    Seq    (&[A] / Vec<A>, A != u8)  :41-80, 284-314        every item in turn, no length prefix (the default batch rule)
    WithLength (AbsorbWithLength)    :84-104                usize(len) first, then the sequence
    Opt    (Option<A>)               :316-331               Bool(is_some), then the item
    TEAffine                         :243-261               elements [x, y]; bytes = x || y, 32 bytes little-endian each
    Struct (#[derive(Absorb)])       macros/src/lib.rs      the fields in declaration order

`to_sponge_field_elements()` returns canonical python ints (the sponge's field is BLS12-381 Fr); `to_sponge_bytes()` bytes.
None of these byte / element layouts is pinned by a vector in the reference (its tests compare encodings with each other,
absorb.rs:393-496): they are restated from the source above and from ark-ff's published `ToConstraintField<[u8]>`.
"""
from .. import field

P = field.MODULUS
_CHUNK = (P.bit_length() - 1) // 8  # 31: bytes per element of ark-ff's `[u8]::to_field_elements`


class Absorbable:
    def to_sponge_bytes(self) -> bytes:
        raise NotImplementedError

    def to_sponge_field_elements(self):
        raise NotImplementedError


class _Unsigned(Absorbable):
    WIDTH = 1

    def __init__(self, v):
        v = int(v)
        if not 0 <= v < (1 << (8 * self.WIDTH)):
            raise OverflowError("%d does not fit %s" % (v, type(self).__name__))
        self.v = v

    def to_sponge_bytes(self):
        return self.v.to_bytes(self.WIDTH, "little")

    def to_sponge_field_elements(self):
        return [self.v % P]


class U8(_Unsigned):
    WIDTH = 1


class U16(_Unsigned):
    WIDTH = 2


class U32(_Unsigned):
    WIDTH = 4


class U64(_Unsigned):
    WIDTH = 8


class U128(_Unsigned):
    WIDTH = 16


class Usize(U64):  # :212-220: absorbed as u64
    pass


class Bool(Absorbable):
    def __init__(self, v):
        self.v = bool(v)

    def to_sponge_bytes(self):
        return bytes([int(self.v)])

    def to_sponge_field_elements(self):
        return [int(self.v)]


class _Signed(Absorbable):
    WIDTH = 1

    def __init__(self, v):
        v = int(v)
        if not -(1 << (8 * self.WIDTH - 1)) <= v < (1 << (8 * self.WIDTH - 1)):
            raise OverflowError("%d does not fit %s" % (v, type(self).__name__))
        self.v = v

    def to_sponge_bytes(self):
        return self.v.to_bytes(self.WIDTH, "little", signed=True)

    def to_sponge_field_elements(self):
        return [(-(abs(self.v) % P)) % P if self.v < 0 else self.v % P]  # :195-201


class I8(_Signed):
    WIDTH = 1


class I16(_Signed):
    WIDTH = 2


class I32(_Signed):
    WIDTH = 4


class I64(_Signed):
    WIDTH = 8


class I128(_Signed):
    WIDTH = 16


class Isize(I64):  # :222-230
    pass


class Fe(Absorbable):
    """a native field element: python int (canonical) or one wire-format element"""

    def __init__(self, v):
        self.v = int(v) % P if isinstance(v, int) else field.to_ints(v)[0]

    def to_sponge_bytes(self):
        return self.v.to_bytes(32, "little")

    def to_sponge_field_elements(self):
        return [self.v]


def bytes_to_field_elements(data: bytes):
    """ark-ff `ToConstraintField<F> for [u8]`: (MODULUS_BIT_SIZE - 1) / 8 = 31-byte little-endian chunks"""
    return [int.from_bytes(data[i:i + _CHUNK], "little") for i in range(0, len(data), _CHUNK)]


class Bytes(Absorbable):
    """&[u8] / Vec<u8>: the u8 batch rule (:133-142)"""

    def __init__(self, b):
        self.b = bytes(b)

    def to_sponge_bytes(self):
        return self.b

    def to_sponge_field_elements(self):
        return bytes_to_field_elements(len(self.b).to_bytes(8, "little") + self.b)


class Str(Absorbable):
    def __init__(self, s):
        self.b = s.encode("utf-8")

    def to_sponge_bytes(self):
        return Usize(len(self.b)).to_sponge_bytes() + self.b

    def to_sponge_field_elements(self):
        return Bytes(self.b).to_sponge_field_elements()


class Seq(Absorbable):
    """&[A] / Vec<A> for A other than u8: the default batch rule -- item after item"""

    def __init__(self, items):
        self.items = list(items)
        assert all(isinstance(i, Absorbable) for i in self.items)
        assert not any(type(i) is U8 for i in self.items), "a slice of u8 is `Bytes` (it has its own batch rule)"

    def to_sponge_bytes(self):
        return b"".join(i.to_sponge_bytes() for i in self.items)

    def to_sponge_field_elements(self):
        return [e for i in self.items for e in i.to_sponge_field_elements()]


class WithLength(Absorbable):
    """AbsorbWithLength::to_sponge_*_with_length (:84-104) of a Bytes / Seq"""

    def __init__(self, seq):
        assert isinstance(seq, (Bytes, Seq))
        self.seq = seq

    def _len(self):
        return len(self.seq.b) if isinstance(self.seq, Bytes) else len(self.seq.items)

    def to_sponge_bytes(self):
        return Usize(self._len()).to_sponge_bytes() + self.seq.to_sponge_bytes()

    def to_sponge_field_elements(self):
        return Usize(self._len()).to_sponge_field_elements() + self.seq.to_sponge_field_elements()


class Opt(Absorbable):
    def __init__(self, item=None):
        assert item is None or isinstance(item, Absorbable)
        self.item = item

    def to_sponge_bytes(self):
        return Bool(self.item is not None).to_sponge_bytes() + (self.item.to_sponge_bytes() if self.item is not None else b"")

    def to_sponge_field_elements(self):
        return Bool(self.item is not None).to_sponge_field_elements() + (self.item.to_sponge_field_elements() if self.item is not None else [])


class TEAffine(Absorbable):
    """a Jubjub affine point (x, y over the sponge's field)"""

    def __init__(self, x, y):
        self.x, self.y = Fe(x), Fe(y)

    def to_sponge_bytes(self):
        return self.x.to_sponge_bytes() + self.y.to_sponge_bytes()

    def to_sponge_field_elements(self):
        return [self.x.v, self.y.v]


class Struct(Absorbable):
    """#[derive(Absorb)]: the fields in declaration order (macros/src/lib.rs)"""

    def __init__(self, *fields):
        assert all(isinstance(f, Absorbable) for f in fields)
        self.fields = fields

    def to_sponge_bytes(self):
        return b"".join(f.to_sponge_bytes() for f in self.fields)

    def to_sponge_field_elements(self):
        return [e for f in self.fields for e in f.to_sponge_field_elements()]


def collect_sponge_bytes(*items) -> bytes:  # collect_sponge_bytes! (:358-370)
    return b"".join(i.to_sponge_bytes() for i in items)


def collect_sponge_field_elements(*items):  # collect_sponge_field_elements! (:372-382)
    return [e for i in items for e in i.to_sponge_field_elements()]
