"""CPU restatement of the reference's string <-> value casts.  TEST INFRASTRUCTURE ONLY: imported by tests/, smoke() and bench.py's
cpu_baseline leg as the checker, never by the product path.

Each function follows the reference function it cites (paths relative to /root/reference/native/spark-expr/src/conversion_funcs) and is
pinned on that file's own unit-test vectors (tests/golden/reference_kats.json "string_casts", tests/test_string_casts_cpu.py).  Strings are
handled as UTF-8 bytes, as the reference does.

A parser returns (value, error): value None = SQL NULL; error is None or the Spark error class the reference raises in that eval mode.
"""
from typing import Optional, Tuple

LEGACY, ANSI, TRY = "legacy", "ansi", "try"

CAST_INVALID = "CAST_INVALID_INPUT"
NUMERIC_OUT_OF_RANGE = "NUMERIC_VALUE_OUT_OF_RANGE"


# --------------------------------------------------------------------------- trimming (trim.rs:41-128)

def trim_all_range(b: bytes) -> Tuple[int, int]:
    """UTF8String.trimAll: bytes <= 0x20 and 0x7F off both ends (trim.rs:47-49, 108-128)."""
    s, e = 0, len(b)
    while s < e and (b[s] <= 0x20 or b[s] == 0x7F):
        s += 1
    while e > s and (b[e - 1] <= 0x20 or b[e - 1] == 0x7F):
        e -= 1
    return s, e


def trim_all(b: bytes) -> bytes:
    s, e = trim_all_range(b)
    return b[s:e]


def trim_java_string(b: bytes) -> bytes:
    """java.lang.String.trim: bytes <= 0x20 (trim.rs:58-61, 92-96)."""
    s, e = 0, len(b)
    while s < e and b[s] <= 0x20:
        s += 1
    while e > s and b[e - 1] <= 0x20:
        e -= 1
    return b[s:e]


# --------------------------------------------------------------------------- string -> boolean (string.rs:260-312)

def string_to_bool(b: bytes, mode: str):
    t = trim_all(b).lower()        # bytes.lower() folds ASCII only, like eq_ignore_ascii_case
    if t in (b"t", b"true", b"y", b"yes", b"1"):
        return True, None
    if t in (b"f", b"false", b"n", b"no", b"0"):
        return False, None
    return None, (CAST_INVALID if mode == ANSI else None)


# --------------------------------------------------------------------------- string -> integers (string.rs:853-1115)

def _parse_sign(b: bytes):
    """string.rs:1107-1115: None for empty input; a lone sign stays part of the digits."""
    if not b:
        return None
    if b[0:1] == b"-" and len(b) > 1:
        return True, b[1:]
    if b[0:1] == b"+" and len(b) > 1:
        return False, b[1:]
    return False, b


def _to_int_generic(b: bytes, mode: str, bits: int):
    """do_parse_string_to_int_{legacy,ansi,try} over i32 / i64 (string.rs:942-1060): the value is accumulated NEGATIVE, digit by digit, and every
    step is checked against min / 10 and the subtraction — so what overflows is decided exactly as the reference decides it."""
    err = CAST_INVALID if mode == ANSI else None
    sg = _parse_sign(trim_all(b))
    if sg is None:
        return None, err
    negative, digits = sg
    lo = -(1 << (bits - 1))
    # Rust integer division truncates toward zero: i32::MIN / 10 = -214748364
    stop = -((-lo) // 10)
    result = 0
    i = 0
    n = len(digits)
    while i < n:
        ch = digits[i]
        i += 1
        if ch == 0x2E:      # '.'
            if mode != LEGACY:
                return None, err
            break
        if not (0x30 <= ch <= 0x39):
            return None, err
        if result < stop:
            return None, err
        x = result * 10 - (ch - 0x30)
        if x < lo or x > 0:       # checked_sub failed, or the value left the non-positive range
            return None, err
        result = x
    else:
        i = n
    if mode == LEGACY:
        for ch in digits[i:]:     # the fraction: digits only, values ignored
            if not (0x30 <= ch <= 0x39):
                return None, None
    # finalize_int_result (string.rs:931-937)
    if negative:
        return result, None
    if result == lo:              # checked_neg fails
        return None, err
    return -result, None


def string_to_int(b: bytes, mode: str, bits: int):
    """cast_string_to_int (string.rs:853-928): Int8 / Int16 parse as i32 and are then range-checked (string.rs:1063-1103)."""
    if bits in (32, 64):
        return _to_int_generic(b, mode, bits)
    v, err = _to_int_generic(b, mode, 32)
    if err:
        return None, err
    if v is not None and -(1 << (bits - 1)) <= v <= (1 << (bits - 1)) - 1:
        return v, None
    return None, (CAST_INVALID if mode == ANSI else None)


# --------------------------------------------------------------------------- string -> decimal (string.rs:314-758)

_I128_MAX = (1 << 127) - 1


def _digits_to_i128(d: bytes) -> Optional[int]:
    """string.rs:536-546: None when the value leaves i128."""
    v = 0
    for ch in d:
        v = v * 10 + (ch - 0x30)
        if v > _I128_MAX:
            return None
    return v


def _normalize_fullwidth(b: bytes) -> bytes:
    """string.rs:472-495: U+FF10..U+FF19 (EF BC 90..99) become ASCII digits."""
    out = bytearray()
    i = 0
    while i < len(b):
        if i + 2 < len(b) and b[i] == 0xEF and b[i + 1] == 0xBC and 0x90 <= b[i + 2] <= 0x99:
            out.append(b[i + 2] - 0x60)
            i += 3
        else:
            out.append(b[i])
            i += 1
    return bytes(out)


def _rust_parse_i32(b: bytes) -> Optional[int]:
    """str::parse::<i32>: optional sign, at least one ASCII digit, no overflow."""
    if not b:
        return None
    neg = False
    if b[0:1] in (b"+", b"-"):
        neg = b[0:1] == b"-"
        b = b[1:]
    if not b or not all(0x30 <= c <= 0x39 for c in b):
        return None
    v = int(b)
    v = -v if neg else v
    return v if -(1 << 31) <= v <= (1 << 31) - 1 else None


def string_to_decimal(b: bytes, precision: int, scale: int, mode: str):
    """parse_string_to_decimal + cast_string_to_decimal128_impl (string.rs:335-396, 579-758).  ANSI raises both for the parser's errors and
    for its Ok(None) (empty, inf / nan, a scale adjustment beyond 38: CAST_INVALID_INPUT); LEGACY / TRY give NULL (string.rs:352-372)."""
    def fail(cls):
        return None, (cls if mode == ANSI else None)
    t = trim_java_string(b)
    if any(c >= 0x80 for c in t):
        t = _normalize_fullwidth(t)
    if not t:
        return fail(CAST_INVALID)
    if t.lower() in (b"inf", b"+inf", b"-inf", b"infinity", b"+infinity", b"-infinity", b"nan"):
        return fail(CAST_INVALID)
    # parse_decimal_str (string.rs:676-758)
    pos = 0
    negative = False
    if t[0:1] == b"-":
        negative, pos = True, 1
    elif t[0:1] == b"+":
        pos = 1
    start = pos
    dot = exp = None
    while pos < len(t):
        ch = t[pos]
        if 0x30 <= ch <= 0x39:
            pos += 1
        elif ch == 0x2E and dot is None:
            dot = pos
            pos += 1
        elif ch in (0x65, 0x45):
            exp = pos
            break
        else:
            return fail(CAST_INVALID)
    exponent = 0
    if exp is not None:
        exponent = _rust_parse_i32(t[exp + 1:])
        if exponent is None:
            return fail(CAST_INVALID)
    mend = exp if exp is not None else pos
    if dot is not None:
        ip, fp = t[start:dot], t[dot + 1:mend]
    else:
        ip, fp = t[start:mend], b""
    if not ip and not fp:
        return fail(CAST_INVALID)
    iv = _digits_to_i128(ip)
    fv = _digits_to_i128(fp)
    if iv is None or fv is None or len(fp) > 38:
        return fail(CAST_INVALID)
    mant = iv * 10 ** len(fp) + fv
    if mant > _I128_MAX:
        return fail(CAST_INVALID)
    if negative:
        mant = -mant
    final_scale = len(fp) - exponent
    # parse_string_to_decimal (string.rs:606-663)
    if mant == 0:
        if final_scale < -37:
            return fail(NUMERIC_OUT_OF_RANGE)
        return 0, None
    adj = scale - final_scale
    if adj >= 0:
        if adj > 38:
            return fail(CAST_INVALID)
        v = mant * 10 ** adj
        if abs(v) > _I128_MAX and v != -(1 << 127):
            return fail(NUMERIC_OUT_OF_RANGE)
    else:
        if -adj > 38:
            return 0, None
        d = 10 ** (-adj)
        q = abs(mant) // d
        r = abs(mant) % d
        if r >= d // 2:
            q += 1
        v = -q if mant < 0 else q
    if abs(v) >= 10 ** precision:           # is_validate_decimal_precision
        return fail(NUMERIC_OUT_OF_RANGE)
    return v, None


# --------------------------------------------------------------------------- string -> date (string.rs:1221-1247, 1896-2046)

def days_from_civil(y: int, m: int, d: int) -> int:
    """string.rs:1221-1228."""
    if m <= 2:
        y, m = y - 1, m + 9
    else:
        m -= 3
    era = y // 400
    yoe = y - era * 400
    doy = (153 * m + 2) // 5 + d - 1
    doe = yoe * 365 + yoe // 4 - yoe // 100 + doy
    return era * 146097 + doe - 719468


def _ymd_to_epoch_day(y: int, m: int, d: int) -> Optional[int]:
    dim = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
    if not 1 <= m <= 12:
        return None
    mx = dim[m - 1]
    if m == 2 and y % 4 == 0 and (y % 100 != 0 or y % 400 == 0):
        mx = 29
    if d < 1 or d > mx:
        return None
    return days_from_civil(y, m, d)


def _wrap_i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >> 31 else v


def string_to_date(b: bytes, mode: str):
    """date_parser (string.rs:1896-2046)."""
    err = "CAST_INVALID_INPUT" if mode == ANSI else None

    def resolve(y, m, d):
        days = _ymd_to_epoch_day(y, m, d)
        if days is None or not -(1 << 31) <= days <= (1 << 31) - 1:
            return None, err
        if not -262143 <= y <= 262142:
            return None, None          # the reference's chrono limit: NULL in every mode
        return days, None
    if not b:
        return None, err
    j, end = trim_all_range(b)
    if j == end:
        return None, err
    t = b[j:end]
    if len(t) == 10 and t[4] == 0x2D and t[7] == 0x2D and all(0x30 <= c <= 0x39 for c in t[0:4] + t[5:7] + t[8:10]):
        return resolve(int(t[0:4]), int(t[5:7]), int(t[8:10]))
    seg = [1, 1, 1]
    sign = 1
    cur = 0
    val = 0
    digits = 0
    if b[j] == 0x2D:
        sign = -1
        j += 1
    elif b[j] == 0x2B:
        j += 1

    def valid_digits(s, n):
        return (s == 0 and 4 <= n <= 7) or (s != 0 and 0 < n <= 2)
    while j < end and cur < 3 and not (b[j] == 0x20 or b[j] == 0x54):
        ch = b[j]
        if cur < 2 and ch == 0x2D:
            if not valid_digits(cur, digits):
                return None, err
            seg[cur] = val
            val = 0
            digits = 0
            cur += 1
        elif not (0x30 <= ch <= 0x39):
            return None, err
        else:
            val = _wrap_i32(val * 10 + (ch - 0x30))
            digits += 1
        j += 1
    if not valid_digits(cur, digits):
        return None, err
    if cur < 2 and j < end:
        return None, err
    seg[cur] = val
    return resolve(_wrap_i32(sign * seg[0]), seg[1], seg[2])


# --------------------------------------------------------------------------- values -> string

def int_to_string(v: int) -> str:
    """arrow-cast's integer formatting (the reference defers to DataFusion for Int -> Utf8, numeric.rs:35-47): plain decimal digits."""
    return str(int(v))


def bool_to_string(v: bool) -> str:
    """arrow-cast boolean -> Utf8 (boolean.rs:25-31): "true" / "false"."""
    return "true" if v else "false"


def decimal_to_string(unscaled: int, scale: int, mode: str) -> str:
    """LEGACY: decimal128_to_java_string (numeric.rs:660-704) = java.math.BigDecimal.toString; ANSI / TRY: arrow-cast's plain notation
    (numeric.rs:76-80)."""
    coeff = str(abs(unscaled))
    n = len(coeff)
    sign = "-" if unscaled < 0 else ""
    adj = -scale + (n - 1)
    plain = mode != LEGACY or (scale >= 0 and adj >= -6)
    if plain:
        if scale <= 0:
            return sign + coeff + "0" * (-scale) if mode != LEGACY else sign + coeff
        if n > scale:
            return sign + coeff[:n - scale] + "." + coeff[n - scale:]
        return sign + "0." + "0" * (scale - n) + coeff
    out = sign + (coeff[0] + "." + coeff[1:] if n > 1 else coeff) + "E"
    if adj > 0:
        out += "+"
    return out + str(adj)


def civil_from_days(z: int):
    """Inverse of days_from_civil (the proleptic Gregorian calendar chrono uses)."""
    z += 719468
    era = z // 146097
    doe = z - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - doe // 146096) // 365
    y = yoe + era * 400
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    mp = (5 * doy + 2) // 153
    d = doy - (153 * mp + 2) // 5 + 1
    m = mp + 3 if mp < 10 else mp - 9
    return (y + 1 if m <= 2 else y), m, d


def _chrono_year(y: int) -> str:
    """chrono's %Y: four digits for 0..=9999, otherwise a sign and at least four digits."""
    if 0 <= y <= 9999:
        return "%04d" % y
    return ("+" if y > 0 else "-") + "%04d" % abs(y)


def date_to_string(days: int) -> str:
    """arrow-cast Date32 -> Utf8 (temporal.rs:26-28): chrono's "%Y-%m-%d"."""
    y, m, d = civil_from_days(int(days))
    return "%s-%02d-%02d" % (_chrono_year(y), m, d)


def timestamp_to_string(micros: int, offset_seconds: int = 0) -> str:
    """arrow-cast Timestamp(µs) -> Utf8 with "%Y-%m-%d %H:%M:%S%.f" (cast.rs:71) in a fixed-offset zone, then the trailing zeroes of the
    fraction removed (utils.rs:88-113)."""
    local = int(micros) + offset_seconds * 1_000_000
    days, rem = divmod(local, 86_400_000_000)
    secs, us = divmod(rem, 1_000_000)
    y, m, d = civil_from_days(days)
    s = "%s-%02d-%02d %02d:%02d:%02d" % (_chrono_year(y), m, d, secs // 3600, secs // 60 % 60, secs % 60)
    if us:
        s += (".%06d" % us).rstrip("0")
    return s


# --------------------------------------------------------------------------- time zones (utils.rs:62-330, temporal.rs:37-78)
# The reference resolves zone names with chrono-tz; here Python's zoneinfo over the database the tests name in $TZDIR.

def _zone(tz: str):
    import datetime
    import zoneinfo
    if tz in ("", "UTC", "Z", "GMT", "Etc/UTC"):
        return datetime.timezone.utc
    if tz[0] in "+-":
        parts = [int(x) for x in tz[1:].split(":")] + [0, 0]
        secs = parts[0] * 3600 + parts[1] * 60 + parts[2]
        return datetime.timezone(datetime.timedelta(seconds=-secs if tz[0] == "-" else secs))
    return zoneinfo.ZoneInfo(tz)


_EPOCH_NAIVE = None


_FOLD_FROM = 16725225600          # 2500-01-01T00:00:00Z: behind every explicit transition of the database, and far inside datetime's range
_FOLD_PERIOD = 146097 * 86400     # the Gregorian calendar, weekdays included, repeats every 400 years — and with it a zone's final rule


def _fold(seconds: int) -> int:
    """a late instant (or wall-clock second) read 400-year periods earlier: the zone's last rule holds for ever (java.time ZoneRules, Spark's
    answers), Python's datetime ends with the year 9999"""
    if seconds < _FOLD_FROM:
        return seconds
    return seconds - ((seconds - _FOLD_FROM) // _FOLD_PERIOD + 1) * _FOLD_PERIOD


def utc_offset_at(tz: str, utc_seconds: int) -> int:
    """seconds east of UTC in force at the instant (tz.from_utc_datetime)"""
    import datetime
    z = _zone(tz)
    if isinstance(z, datetime.timezone):          # a fixed offset: no calendar needed (years beyond datetime's stay representable)
        return int(z.utcoffset(None).total_seconds())
    utc_seconds = max(-62135510400, _fold(utc_seconds))      # (before the year 1: the zone's local mean time, as in the year 1)
    t = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc) + datetime.timedelta(seconds=utc_seconds)
    return int(t.astimezone(z).utcoffset().total_seconds())


def utc_to_local_us(tz: str, us: int) -> int:
    return us + utc_offset_at(tz, us // 1_000_000) * 1_000_000


def local_to_utc_us(tz: str, local_us: int) -> int:
    """resolve_local_datetime (utils.rs:184-205): Single / Ambiguous → the (earlier) instant; a gap → the offset of the wall clock three hours
    before; that one in a gap as well → the wall clock read as UTC."""
    import datetime
    z = _zone(tz)
    utc = datetime.timezone.utc
    if isinstance(z, datetime.timezone):
        return local_us - int(z.utcoffset(None).total_seconds()) * 1_000_000

    def in_gap(naive):
        return naive.replace(tzinfo=z, fold=0).astimezone(utc).astimezone(z).replace(tzinfo=None) != naive
    L = max(-62135510400, _fold(local_us // 1_000_000))
    naive = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=L)
    if not in_gap(naive):
        off = int(naive.replace(tzinfo=z, fold=0).utcoffset().total_seconds())
    else:
        probe = naive - datetime.timedelta(hours=3)
        off = 0 if in_gap(probe) else int(probe.replace(tzinfo=z, fold=0).utcoffset().total_seconds())
    return local_us - off * 1_000_000


# --------------------------------------------------------------------------- floats (numeric.rs:137-221, 884-990)

def _shortest_digits(x, is32: bool):
    """(mantissa without trailing zeroes, exponent) of the shortest decimal that reads back as the same float — Python's repr() for doubles,
    numpy's unique formatting for floats: the digits the `ryu` crate and Rust's Display produce"""
    import decimal
    import numpy as np
    text = np.format_float_scientific(np.float32(x), unique=True) if is32 else repr(abs(float(x)))
    t = decimal.Decimal(text).as_tuple()
    m, e = int("".join(map(str, t.digits))), t.exponent
    while m and m % 10 == 0:
        m //= 10
        e += 1
    return abs(m), e


def float_to_string(x, is32: bool) -> str:
    """spark_cast_float64_to_utf8 / float32 (numeric.rs:137-221)"""
    import math
    import numpy as np
    x = float(np.float32(x)) if is32 else float(x)
    if x != x:
        return "NaN"
    if math.isinf(x):
        return "Infinity" if x > 0 else "-Infinity"
    sign = "-" if math.copysign(1.0, x) < 0 else ""
    a = abs(x)
    if a == 0:
        return sign + "0.0"
    if a == (float(np.float32(1.4e-45)) if is32 else 5e-324):
        return sign + ("1.4E-45" if is32 else "4.9E-324")
    m, e = _shortest_digits(a, is32)
    d = str(m)
    n = len(d)
    lo, hi = (float(np.float32(0.001)), float(np.float32(1e7))) if is32 else (0.001, 1e7)
    if lo <= a < hi:
        point = n + e
        if point <= 0:
            return sign + "0." + "0" * -point + d
        if point >= n:
            return sign + d + "0" * (point - n) + ".0"
        return sign + d[:point] + "." + d[point:]
    return sign + d[0] + "." + (d[1:] if n > 1 else "0") + "E" + str(e + n - 1)


def float_to_decimal(x, precision: int, scale: int):
    """float_to_decimal128 (numeric.rs:955-990): the shortest digits of the value AS A DOUBLE, HALF_UP at the target scale; (None, None) for NaN /
    infinity, (None, "overflow") beyond the precision"""
    import math
    x = float(x)
    if x != x or math.isinf(x):
        return None, None
    if x == 0:
        return 0, None
    m, e = _shortest_digits(x, False)
    shift = e + scale
    if shift >= 0:
        v = m * 10 ** shift if shift <= 38 else None
    elif -shift > 38:
        v = 0
    else:
        d = 10 ** -shift
        v = m // d + (1 if m % d >= d // 2 else 0)
    if v is None or v >= 10 ** precision:
        return None, "overflow"
    return (-v if x < 0 else v), None


def string_to_float(b: bytes, mode: str, is32: bool):
    """cast_string_to_float (string.rs:177-258): String.trim, the names of infinity / NaN, one trailing d / D / f / F, Rust's float grammar; the
    conversion is correctly rounded straight to the target's width (Python's float() for doubles; exact rational rounding for floats)."""
    import re
    import numpy as np
    err = CAST_INVALID if mode == ANSI else None
    t = trim_java_string(b)
    try:
        s = t.decode("ascii")
    except UnicodeDecodeError:
        return None, err
    low = s.lower()
    if low in ("inf", "+inf", "infinity", "+infinity"):
        return float("inf"), None
    if low in ("-inf", "-infinity"):
        return float("-inf"), None
    if low == "nan":
        return float("nan"), None
    if s[-1:] in ("d", "D", "f", "F"):
        s = s[:-1]
    m = re.fullmatch(r"([+-]?)(inf|infinity|nan|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)", s, re.IGNORECASE)
    if not m:
        return None, err
    if m.group(2).lower() in ("inf", "infinity", "nan"):
        return float(m.group(1) + m.group(2)), None
    if not is32:
        return float(s), None
    from fractions import Fraction
    q = Fraction(s)
    neg = s.startswith("-")
    q = abs(q)
    if q == 0:
        return -0.0 if neg else 0.0, None
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    e = max(e, -126)
    scaled = q / Fraction(2) ** (e - 23)
    mant = scaled.numerator // scaled.denominator
    rem = scaled - mant
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and mant & 1):
        mant += 1
    v = float("inf") if (e + (1 if mant == 1 << 24 else 0)) > 127 else float(Fraction(mant) * Fraction(2) ** (e - 23))
    return float(np.float32(-v if neg else v)), None


# --------------------------------------------------------------------------- string -> timestamp / timestamp_ntz (string.rs:798-852, 1125-1900)
# The reference matches fourteen regular expressions (Unicode \d), splits the value on [T -:.] and lets failed integer parses fall back to
# defaults; both are restated as they are.  Time-only values ("T12:34", "12:34:56") take TODAY's date in the zone: `now_us` says what today is.

import re as _re

_WS = set([9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000] + list(range(0x2000, 0x200B)))
_RX = {k: _re.compile(v) for k, v in {
    "year": r"-?\d{4,6}", "month": r"-?\d{4,7}-\d{2}", "day": r"-?\d{4,7}-\d{2}-\d{2}", "hour": r"-?\d{4,7}-\d{2}-\d{2}[T ]\d{1,2}",
    "minute": r"-?\d{4,7}-\d{2}-\d{2}[T ]\d{2}:\d{2}", "second": r"-?\d{4,7}-\d{2}-\d{2}[T ]\d{2}:\d{2}:\d{2}",
    "microsecond": r"-?\d{4,7}-\d{2}-\d{2}[T ]\d{2}:\d{2}:\d{2}\.\d+", "t_h": r"T\d{1,2}", "t_hm": r"T\d{1,2}:\d{1,2}", "t_hms": r"T\d{1,2}:\d{1,2}:\d{1,2}",
    "t_hmsu": r"T\d{1,2}:\d{1,2}:\d{1,2}\.\d+", "b_hm": r"\d{1,2}:\d{1,2}", "b_hms": r"\d{1,2}:\d{1,2}:\d{1,2}", "b_hmsu": r"\d{1,2}:\d{1,2}:\d{1,2}\.\d+"}.items()}
_DATE_KINDS = ["year", "month", "day", "hour", "minute", "second", "microsecond"]
_TIME_KINDS = ["t_h", "t_hm", "t_hms", "t_hmsu", "b_hm", "b_hms", "b_hmsu"]


def _trim_ws(s: str, left=True, right=True) -> str:
    a, b = 0, len(s)
    while left and a < b and ord(s[a]) in _WS:
        a += 1
    while right and b > a and ord(s[b - 1]) in _WS:
        b -= 1
    return s[a:b]


def _rust_int(s: str, signed: bool, bits: int = 32):
    """str::parse::<i32 / u32>: an optional sign ('-' for signed types only), ASCII digits, no overflow"""
    if not s:
        return None
    t = s
    neg = False
    if t[0] == "+" or (signed and t[0] == "-"):
        neg = t[0] == "-"
        t = t[1:]
    if not t or any(c < "0" or c > "9" for c in t):
        return None
    v = -int(t) if neg else int(t)
    lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
    return v if lo <= v <= hi else None


def _ts_info(value: str, kind: str):
    """parse_to_timestamp_info (string.rs:1125-1205) → (y, mo, d, h, mi, s, us) or None"""
    sign, part = (-1, value[1:]) if value.startswith("-") else (1, value)
    parts = _re.split(r"[T \-:.]", part)

    def nxt(i, signed, default):
        if i >= len(parts):
            return default
        v = _rust_int(parts[i], signed)
        return default if v is None else v
    y = _rust_int(parts[0], True)
    year = sign * (0 if y is None else y)
    if not -290309 <= year <= 294248:
        return None
    mo, d, h, mi, s = nxt(1, False, 1), nxt(2, False, 1), nxt(3, False, 0), nxt(4, False, 0), nxt(5, False, 0)
    us = 0
    if len(parts) > 6:
        ms = parts[6].encode()[:6]
        try:
            v = _rust_int(ms.decode(), False)
        except UnicodeDecodeError:
            v = None
        us = (0 if v is None else v) * 10 ** (6 - len(ms))
    info = [1, 1, 1, 0, 0, 0, 0]
    got = [year, mo, d, h, mi, s, us]
    for k in range(_DATE_KINDS.index(kind) + 1):
        info[k] = got[k]
    return info


def _local_candidates(tz: str, local_s: int):
    """offsets of the spans whose wall clock shows local second L (chrono-tz from_local_datetime): [] a gap, [o] single, [o1, o2] ambiguous"""
    import datetime
    z = _zone(tz)
    if isinstance(z, datetime.timezone):
        return [int(z.utcoffset(None).total_seconds())]
    # (years beyond datetime's: before the zone's first transition its local mean time holds — a single offset; late ones fold)
    local_s = max(-62135510400, _fold(local_s))
    naive = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=local_s)
    utc = datetime.timezone.utc
    out = []
    for fold in (0, 1):
        aware = naive.replace(tzinfo=z, fold=fold)
        if aware.astimezone(utc).astimezone(z).replace(tzinfo=None) == naive:
            o = int(aware.utcoffset().total_seconds())
            if o not in out:
                out.append(o)
    return out


def _ts_to_micros(info, tz: str):
    """parse_timestamp_to_micros (string.rs:1249-1348)"""
    y, mo, d, h, mi, s, us = info
    if not (h < 24 and mi < 60 and s < 60):
        return None
    days = _ymd_to_epoch_day(y, mo, d)
    if days is not None and -262143 <= y <= 262142:
        local = days * 86400 + h * 3600 + mi * 60 + s
        c = _local_candidates(tz, local)
        if c:
            off = c[0]
        else:
            c = _local_candidates(tz, local - 10800)
            if not c:
                return None
            off = c[0]
        return (local - off) * 1_000_000 + us
    if -262144 <= y <= 262143 or days is None:
        return None
    c = _local_candidates(tz, 0)
    off = c[0] if c else 0
    v = (days * 86400 + h * 3600 + mi * 60 + s - off) * 1_000_000 + us
    return v if -(1 << 63) <= v < (1 << 63) else None


def _parse_sign_offset(s: str):
    """string.rs:1493-1531"""
    if s == "":
        return 0
    if s[0] == "+":
        sign = 1
    elif s[0] == "-":
        sign = -1
    else:
        return None
    rest = s[1:]
    if not rest:
        return None
    if ":" in rest:
        hs, ms = rest.split(":", 1)
        if not ms:
            return None
        h, m = _rust_int(hs, True), _rust_int(ms, True)
        if h is None or m is None:
            return None
    else:
        if len(rest.encode()) in (1, 2):
            h, m = _rust_int(rest, True), 0
        elif len(rest.encode()) == 4 and rest.isascii():
            h, m = _rust_int(rest[:2], True), _rust_int(rest[2:], True)
        else:
            return None
        if h is None or m is None:
            return None
    if not (0 <= h <= 18 and 0 <= m <= 59):
        return None
    return sign * (h * 3600 + m * 60)


def _fixed_zone(secs: int) -> str:
    a = abs(secs)
    return "%s%02d:%02d" % ("-" if secs < 0 else "+", a // 3600, a % 3600 // 60)


def _extract_offset_suffix(value: str):
    """string.rs:1566-1641 → (prefix, zone name) or None"""
    if value.endswith("Z"):
        return value[:-1], "UTC"
    for prefix in (" UTC", "UTC", " GMT", "GMT", " UT", "UT"):
        pos = value.rfind(prefix)
        if pos >= 0:
            secs = _parse_sign_offset(value[pos + len(prefix):])
            if secs is not None:
                return value[:pos], _fixed_zone(secs)
    for abbr, secs in ((" EST", -18000), ("EST", -18000), (" MST", -25200), ("MST", -25200), (" HST", -36000), ("HST", -36000)):
        pos = value.rfind(abbr)
        if pos >= 0 and pos + len(abbr) == len(value):
            return value[:pos], _fixed_zone(secs)
    sp = value.rfind(" ")
    if sp >= 0:
        name = value[sp + 1:]
        if "/" in name:
            try:
                import zoneinfo
                zoneinfo.ZoneInfo(name)
                return value[:sp], name
            except Exception:      # noqa: BLE001 — not a zone of the database
                pass
    pos = max(value.rfind("+"), value.rfind("-"))
    if pos >= 0:
        secs = _parse_sign_offset(value[pos:])
        if secs is not None:
            return value[:pos], _fixed_zone(secs)
    return None


def _time_only(value: str, tz: str, now_us: int):
    """parse_str_to_time_only_timestamp (string.rs:1855-1893): today's date in the zone with the value's time of day"""
    t = value[1:] if value.startswith("T") else value
    cp = t.split(":", 2)
    hour = _rust_int(cp[0], False) or 0
    minute = (_rust_int(cp[1], False) or 0) if len(cp) > 1 else 0
    sec, ns = 0, 0
    if len(cp) > 2:
        sf = cp[2]
        dot = sf.find(".")
        sec = _rust_int(sf[:dot] if dot >= 0 else sf, False) or 0
        if dot >= 0:
            frac = sf[dot + 1:].encode()[:6].decode(errors="ignore")
            ns = (_rust_int(frac.ljust(6, "0"), False) or 0) * 1000
    if hour >= 24 or minute >= 60 or sec >= 60:
        return None
    local_now = utc_to_local_us(tz, now_us)
    day = local_now // 86_400_000_000
    local = day * 86400 + hour * 3600 + minute * 60 + sec
    c = _local_candidates(tz, local)
    if len(c) != 1:                       # DateTime::with_hour…: `single()` — a gap or an overlap gives None
        return None
    return (local - c[0]) * 1_000_000 + ns // 1000


def _leading_plus(value: str):
    if not value.startswith("+"):
        return value
    rest = value[1:]
    i = next((k for k, c in enumerate(rest) if not ("0" <= c <= "9")), None)
    if i is not None and i >= 1 and rest[i] == "-":
        return rest
    return None


def string_to_timestamp(b: bytes, mode: str, tz: str, spark4: bool, now_us: int = 0):
    """cast_string_to_timestamp → timestamp_parser (string.rs:798-829, 1406-1491, 1667-1731)"""
    err = CAST_INVALID if mode == ANSI else None
    value = _trim_ws(b.decode("utf-8"), left=False)        # the cast's trim_end
    trimmed = _trim_ws(value)
    if not trimmed:
        return None, None
    if spark4 and len(value) > len(_trim_ws(value, right=False)) and any(_RX[k].fullmatch(trimmed) for k in ("t_h", "t_hm", "t_hms", "t_hmsu")):
        return None, err
    value = _leading_plus(trimmed)
    if value is None:
        return None, None
    zone = tz
    if not any(r.fullmatch(value) for r in _RX.values()):
        sfx = _extract_offset_suffix(value)
        if sfx:
            value, zone = sfx
    for k in _DATE_KINDS + _TIME_KINDS:
        if _RX[k].fullmatch(value):
            if k in _DATE_KINDS:
                info = _ts_info(value, k)
                v = None if info is None else _ts_to_micros(info, zone)
            else:
                v = _time_only(value, zone, now_us)
            return (v, None) if v is not None else (None, err)
    return None, err


def string_to_timestamp_ntz(b: bytes, mode: str, allow_time_zone: bool = True):
    """cast_string_to_timestamp_ntz → timestamp_ntz_parser (string.rs:831-852, 1733-1853)"""
    err = CAST_INVALID if mode == ANSI else None
    value = _trim_ws(b.decode("utf-8"))
    if not value:
        return None, None
    value = _leading_plus(value)
    if value is None:
        return None, None
    if any(_RX[k].fullmatch(value) for k in _TIME_KINDS):
        return None, err
    if not any(_RX[k].fullmatch(value) for k in _DATE_KINDS):
        sfx = _extract_offset_suffix(value)
        if sfx:
            if not allow_time_zone:
                return None, err
            value = _trim_ws(sfx[0], left=False)
    for k in _DATE_KINDS:
        if _RX[k].fullmatch(value):
            info = _ts_info(value, k)
            if info is None:
                return None, None
            y, mo, d, h, mi, s, us = info
            days = _ymd_to_epoch_day(y, mo, d)
            if days is None or h >= 24 or mi >= 60 or s >= 60:
                return None, err
            v = (days * 86400 + h * 3600 + mi * 60 + s) * 1_000_000 + us
            return (v, None) if -(1 << 63) <= v < (1 << 63) else (None, err)
    return None, err
