"""The .obj scanner's %f (raytracer_amd/host/MeshLoader.h: objFastFloat + strtof fallback) must convert exactly like
the strtof behind the reference's sscanf("%f") (MeshLoader.h:323-371 of the reference): same bits, same stop position."""
import ctypes as C
import random
import struct

import pytest

from raytracer_amd import _native


@pytest.fixture(scope="module")
def scan():
    host = _native.host_lib(lazy=True)
    libc = C.CDLL("libc.so.6")
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]

    def both(text):
        b = text.encode()
        v, n, fast = C.c_float(), C.c_int32(), C.c_int32()
        host.rth_scan_float(b, C.byref(v), C.byref(n), C.byref(fast))
        buf = C.create_string_buffer(b)
        end = C.c_char_p()
        ref = libc.strtof(buf, C.byref(end))
        consumed = C.cast(end, C.c_void_p).value - C.addressof(buf)
        bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
        return bits(v.value), n.value, bits(ref), consumed, fast.value
    return both


CASES = ["0", "-0", "0.0", "-0.000", "1", "-1", "1.5", "-.5", "5.", ".", "-", "", "abc", "+1.5", "+.25", "1e", "1e+", "1e-", "1e5",
         "1E-5", "1.25e+2x", "12abc", "0x1p3", "0x10", "-0X1.8p1", "inf", "-inf", "infinity", "nan", "nan(12)", "1e-40", "1e-38",
         "1.17549435e-38", "1.1754942e-38", "3.4028235e38", "3.5e38", "1e39", "1e-46", "16777217", "16777216", "16777218",
         "16777219", "33554434", "33554438", "9007199254740993", "9007199254740992", "12345678901234567890", "0.1", "0.2", "0.3",
         "123456.789", "0.000001234", "00000012.5000", "1.00000005960464477539", "1.0000000596046448", "1.00000006", "1/2/3",
         "7//9", "2.5/1", "1e22", "1e23", "1e-22", "1e-23", "8388608.5", "8388609.5", "4194304.25", "0.50000002980232238769531",
         "1.5e0010", "1e00000000000000000005", "1e99999999999", "-1e-99999999999", "1.e3", ".e3", "1.2.3", "1..2"]


def test_dangling_exponent_follows_strtof(scan):
    """A malformed number with a dangling exponent: glibc's scanf("%f") consumes the 'e' and fails the conversion (target left
    stale), strtof converts the mantissa and stops in front of the 'e'.  The scanner follows strtof -- a deliberate, documented
    difference from the reference's sscanf on input that none of its assets contains (MeshLoader.h)."""
    for text, value, consumed in (("1e", 1.0, 1), ("1.5e+ 2", 1.5, 3), ("-2.25e-x", -2.25, 5), ("3E", 3.0, 1)):
        v, n, ref, rn, _ = scan(text)
        assert (v, n) == (ref, rn)                                    # the same as libc's strtof ...
        assert n == consumed and struct.unpack("<f", struct.pack("<I", v))[0] == value  # ... which converts the mantissa alone


def test_scan_float_cases(scan):
    for t in CASES:
        v, n, ref, rn, _ = scan(t)
        assert (v, n) == (ref, rn) or (n == 0 and rn == 0), (t, hex(v), n, hex(ref), rn)


def test_scan_float_random(scan):
    rng = random.Random(7)
    fast = 0
    total = 20000
    for _ in range(total):
        nd = rng.randint(1, 21)
        digits = "".join(rng.choice("0123456789") for _ in range(nd))
        k = rng.randint(0, nd)
        t = digits[:k] + ("." if rng.random() < 0.8 else "") + digits[k:]
        if rng.random() < 0.3:
            t += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randint(0, 45))
        if rng.random() < 0.5:
            t = "-" + t
        t += rng.choice(["", " ", "/", "//3", " 4.5", "\r"])
        v, n, ref, rn, f = scan(t)
        assert (v, n) == (ref, rn), (t, hex(v), n, hex(ref), rn)
        fast += f
    assert fast > total // 2  # the fast path is the common one


def test_scan_float_near_midpoints(scan):
    # decimals close to float rounding midpoints: the double-rounding trap the fast path must hand to strtof
    rng = random.Random(11)
    for _ in range(4000):
        f = struct.unpack("<f", struct.pack("<I", rng.randint(0x00800000, 0x7F000000)))[0]
        g = struct.unpack("<f", struct.pack("<I", struct.unpack("<I", struct.pack("<f", f))[0] + 1))[0]
        mid = (f + g) / 2  # exact in double
        for t in (repr(mid), "%.17g" % mid, "%.16g" % mid, "%.9g" % mid, "%.8g" % mid):
            v, n, ref, rn, _ = scan(t)
            assert (v, n) == (ref, rn), (t, hex(v), n, hex(ref), rn)
