"""Exact Python-``int`` restatement of the reference's decimal rules — a second, independently written oracle
used to cross-check comet_oracle.c on random inputs (tests/test_oracle_kat.py).  TEST INFRASTRUCTURE ONLY."""


def div_round_half_up(value: int, divisor: int) -> int:
    """wide_decimal_binary_expr.rs:121-144 (truncating quotient/remainder like Rust)."""
    q = abs(value) // abs(divisor)
    r = abs(value) - q * abs(divisor)
    neg = (value < 0) != (divisor < 0)
    quot = -q if neg else q
    if r * 2 >= abs(divisor):
        return quot - 1 if neg else quot + 1
    return quot


def _wrap256(x: int) -> int:
    x &= (1 << 256) - 1
    return x - (1 << 256) if x >> 255 else x


def wide_decimal(op: str, l: int, s1: int, r: int, s2: int, p_out: int, s_out: int):
    """WideDecimalBinaryExpr (wide_decimal_binary_expr.rs:179-300): returns the value or None on overflow."""
    if op == "multiply":
        raw = _wrap256(l * r)
        diff = s1 + s2 - s_out
    else:
        m = max(s1, s2)
        a, b = _wrap256(l * 10 ** (m - s1)), _wrap256(r * 10 ** (m - s2))
        raw = _wrap256(a + b if op == "add" else a - b)
        diff = m - s_out
    if diff > 0:
        res = div_round_half_up(raw, 10 ** diff)
    elif diff < 0:
        res = _wrap256(raw * 10 ** (-diff))
    else:
        res = raw
    bound = 10 ** p_out - 1
    return None if (res > bound or res < -bound) else res


def rescale_check(v: int, s_in: int, p_out: int, s_out: int):
    """decimal_rescale_check.rs:108-150."""
    delta = s_out - s_in
    f = 10 ** abs(delta)
    if delta > 0:
        r = v * f
        if not (-(1 << 127) <= r < (1 << 127)):
            return None
    elif delta < 0:
        half = f // 2
        sign = (v > 0) - (v < 0)
        t = v + sign * half
        r = abs(t) // f * (1 if t >= 0 else -1)   # truncating division
    else:
        r = v
    return None if abs(r) > 10 ** p_out - 1 else r


def avg_decimal(sum_: int, count: int, target_p: int, target_s: int, sum_s: int):
    """avg() — avg_decimal.rs:670-689."""
    value = sum_ * 10 ** max(0, target_s - sum_s)
    if not (-(1 << 127) <= value < (1 << 127)):
        return None
    q = abs(value) // count
    rem = abs(value) - q * count
    div = q if value >= 0 else -q
    rem = rem if value >= 0 else -rem
    half = (count + 1) // 2
    nv = div
    if value >= 0:
        if rem >= half:
            nv = div + 1
    elif rem <= -half:
        nv = div - 1
    b = 10 ** target_p - 1
    return nv if -b <= nv <= b else None
