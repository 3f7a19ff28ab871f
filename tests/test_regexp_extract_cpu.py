"""regexp_extract's matcher on the host (comet_regexp_extract_host: csrc/regex.cpp compile_regex_captures + the SAME source the generated
kernels run, csrc/device/regex_vm.hpp) against the reference's own test vectors (string_funcs/regexp_extract.rs:150-296) and against
Python's `re` — a backtracking engine, i.e. leftmost and preference-ordered like the crate's captures — over generated patterns."""
import random
import re

import pytest

from datafusion_comet_amd import native


def ext(pattern, idx, value):
    return native.regexp_extract_host(pattern, idx, value)[1]


def test_reference_vectors():
    # basic_group_extraction, idx_zero_returns_whole_match, default idx, unmatched_optional_group_returns_empty_string
    assert [ext(r"(\d+)-(\d+)", 1, v) for v in ("100-200", "foo-bar", "nodelim")] == ["100", "", ""]
    assert ext(r"\d+", 0, "abc123def456") == "123"
    assert ext(r"(\d+)-(\d+)", 1, "100-200") == "100"
    assert [ext(r"(\d)", 1, v) for v in ("a1b", "c2d")] == ["1", "2"]
    assert ext(r"(foo)(bar)?", 2, "foo") == ""
    assert [ext(r"(\d+)-(\d+)", 1, v) for v in ("100-200", "foo-bar")] == ["100", ""]


def test_reference_errors():
    # group_index_out_of_range_errors, negative_index_errors, invalid_regex_errors (regexp_extract_common.rs:85-92)
    with pytest.raises(Exception, match=r"Expects group index between 0 and 2, but got 3"):
        ext(r"(a)(b)", 3, "abc")
    with pytest.raises(Exception, match=r"group index between 0 and 1, but got -1"):
        ext(r"(a)", -1, "abc")
    with pytest.raises(Exception, match=r"unclosed group"):
        ext(r"(unclosed", 0, "abc")


def test_preference_order_and_lazy_forms():
    assert ext(r"(a|ab)(c|bcd)(d*)", 1, "abcd") == "a"
    assert ext(r"(a|ab)(c|bcd)(d*)", 3, "abcd") == ""
    assert ext(r"(a+?)(a*)", 2, "aaa") == "aa"
    assert ext(r"<(.+?)>", 1, "<a><b>") == "a"
    assert ext(r"<(.+)>", 1, "<a><b>") == "a><b"
    assert ext(r"(\w+)\s(\w+)", 2, "höhe über null") == "über"
    assert ext(r"(?i)(strasse|weg)", 1, "Zur STRASSE") == "STRASSE"
    assert ext(r"(?m)^(\w+)$", 1, "ab cd\nef\ngh ij") == "ef"
    assert ext(r"\b(\d{2})\b", 1, "123 45 6") == "45"
    assert ext(r"(?:(a)|b)*", 1, "ab") == "a"
    assert ext(r"x*", 0, "éx") == ""
    assert ext(r"(x*)$", 1, "éxx") == "xx"


def test_refusals_name_the_construct():
    for pat, why in ((r"(a*)*", "empty string"), (r"(?P<n>a)", "named groups"), (r"a{1,40}b{1,40}", "matcher instructions"), (r"\p{L}", "escape")):
        with pytest.raises(Exception, match=why):
            ext(pat, 0, "a")


ALPHA = ["a", "b", "c", "0", "1", "-", "_", " ", "é", "中", "ß", "٣"]


def gen(rng, depth, groups):
    """→ (pattern for both engines): the subset both read the same way"""
    r = rng.random()
    if depth <= 0 or r < 0.35:
        k = rng.random()
        if k < 0.5:
            return re.escape(rng.choice(ALPHA)) if rng.random() < 0.9 else rng.choice(ALPHA)
        if k < 0.6:
            return "."
        if k < 0.75:
            return rng.choice([r"\d", r"\w", r"\s", r"\D", r"\W"])
        if k < 0.9:
            return rng.choice(["[a-c]", "[^a]", "[0-9_]", "[é中]", "[^0-9é]", r"[\w-]"])
        return rng.choice([r"\b", "^", "$"])
    if r < 0.55:
        return "".join(gen(rng, depth - 1, groups) for _ in range(rng.randint(2, 3)))
    if r < 0.7:
        return "(?:" + "|".join(gen(rng, depth - 1, groups) for _ in range(rng.randint(2, 3))) + ")"
    if r < 0.85:
        groups[0] += 1
        return "(" + gen(rng, depth - 1, groups) + ")"
    inner = gen(rng, depth - 1, groups)
    if inner in (r"\b", "^", "$"):
        return inner
    q = rng.choice(["*", "+", "?", "{2}", "{1,2}", "{0,2}", "{2,}"]) + ("?" if rng.random() < 0.3 else "")
    if len(inner) > 1 and not (inner.startswith("(") and inner.endswith(")")) and not (inner.startswith("[") and inner.endswith("]") and inner.count("[") == 1) and not (
            len(inner) == 2 and inner[0] == "\\"):
        inner = "(?:" + inner + ")"
    return inner + q


def test_against_a_backtracking_engine():
    rng = random.Random(20260925)
    checked = refused = 0
    for _ in range(1500):
        groups = [0]
        pat = gen(rng, 3, groups)
        flags = rng.choice(["", "", "", "(?i)", "(?s)", "(?m)"])
        if flags == "(?i)" and re.search("[éß中٣]", pat):
            flags = ""
        if flags == "(?m)" and r"\b" in pat:
            flags = ""
        # Python reads $ as "at the end or before a final newline"; the crate as "at the end" (\Z there); under (?m) both mean line ends
        py = flags + (pat if flags == "(?m)" else pat.replace("$", r"\Z"))
        try:
            pyre = re.compile(py)
        except re.error:
            continue
        texts = ["".join(rng.choice(ALPHA + (["\n"] if flags in ("(?s)", "(?m)") else [])) for _ in range(rng.randint(0, 9))) for _ in range(6)]
        for idx in range(groups[0] + 1):
            try:
                got = [ext(flags + pat, idx, t) for t in texts]
            except Exception as e:  # noqa: BLE001
                assert "not supported" in str(e), (pat, str(e))
                refused += 1
                break
            for t, g in zip(texts, got):
                m = pyre.search(t)
                want = (m.group(idx) or "") if m else ""
                assert g == want, (flags + pat, idx, t, g, want)
                checked += 1
    assert checked > 8000 and refused < checked


def test_the_gpu_tests_patterns_on_the_host():
    """what tests/test_regexp_extract_gpu.py asks of the device, asked of the same matcher here — against the oracle's restatement"""
    import pyarrow as pa
    from datafusion_comet_amd import serde as S
    from oracle import oracle as O
    from tests import test_regexp_extract_gpu as G
    t = G._table(3000)
    vals = t.column(0).to_pylist()
    for pat, idx in G.PATTERNS:
        want = O.run_plan_to_arrow(S, S.project(S.scan([S.T_STRING, S.T_INT32]), [G._rx(pat, idx)]), t).column(0).to_pylist()
        got = [None if v is None else ext(pat, idx, v) for v in vals]
        assert got == want, (pat, idx)


# ---- split: the device's two passes (rx_split) on the host ----

def test_split_reference_vectors():
    # string_funcs/split.rs tests: test_split_regex, _limit_positive, _limit_zero, _limit_negative, _empty_string
    assert native.split_host(r"\d+", -1, "foo123bar456baz") == ["foo", "bar", "baz"]
    assert native.split_host(",", 3, "a,b,c,d,e") == ["a", "b", "c,d,e"]
    assert native.split_host(",", 0, "a,b,c,,") == ["a", "b", "c"]
    assert native.split_host(",", -1, "a,b,c,,") == ["a", "b", "c", "", ""]
    assert native.split_host(",", -1, "") == [""]


def test_split_empty_matches_follow_find_iter():
    assert native.split_host("", -1, "abc") == ["", "a", "b", "c", ""]
    assert native.split_host("", 0, "abc") == ["", "a", "b", "c"]
    assert native.split_host("x*", -1, "axxbéc") == ["", "a", "b", "é", "c", ""]
    assert native.split_host("", -1, "") == ["", ""]
    assert native.split_host(",", 0, ",,,") == [""]
    assert native.split_host(r"\b", -1, "ab cd") == ["", "ab", " ", "cd", ""]
    assert native.split_host(r"(?m)^", -1, "a\nb\n") == ["", "a\n", "b\n", ""]


def test_split_against_the_oracles_restatement():
    from oracle import oracle as O
    rng = random.Random(77)
    checked = 0
    for _ in range(500):
        groups = [0]
        pat = gen(rng, 2, groups)
        try:
            O.crate_pattern_to_python(pat)
        except re.error:
            continue
        for _ in range(5):
            text = "".join(rng.choice(ALPHA) for _ in range(rng.randint(0, 10)))
            limit = rng.choice([-1, -1, 0, 1, 2, 3])
            try:
                got = native.split_host(pat, limit, text)
            except Exception as e:  # noqa: BLE001
                assert "not supported" in str(e), (pat, str(e))
                break
            assert got == O.split_like_the_crate(pat, text, limit), (pat, limit, text)
            checked += 1
    assert checked > 1500


# ---- regexp_extract_all: the device's two passes (rx_find_all) on the host ----

def test_extract_all_reference_vectors():
    # string_funcs/regexp_extract_all.rs tests: basic_group_extraction, second_group, idx_zero_returns_whole_matches, no_match_returns_empty_array,
    # null_subject_returns_null's valid rows, unmatched_optional_group_returns_empty_string
    ea = native.extract_all_host
    assert [ea(r"(\d+)-(\d+)", 1, v) for v in ("100-200, 300-400", "foo-bar", "nodelim")] == [["100", "300"], [], []]
    assert ea(r"(\d+)-(\d+)", 2, "100-200, 300-400") == ["200", "400"]
    assert ea(r"\d+", 0, "abc123def456") == ["123", "456"]
    assert ea(r"(\d+)", 1, "abc") == []
    assert [ea(r"(\d)", 1, v) for v in ("1 2 3", "4 5")] == [["1", "2", "3"], ["4", "5"]]
    assert ea(r"(foo)(bar)?", 2, "foo foo") == ["", ""]
    with pytest.raises(Exception, match=r"in `regexp_extract_all` is invalid: Expects group index between 0 and 2, but got 3"):
        ea(r"(a)(b)", 3, "abc")


def test_extract_all_against_the_oracles_restatement():
    from oracle import oracle as O
    rng = random.Random(78)
    checked = 0
    for _ in range(500):
        groups = [0]
        pat = gen(rng, 2, groups)
        try:
            rx = O.crate_pattern_to_python(pat)
        except re.error:
            continue
        for _ in range(4):
            text = "".join(rng.choice(ALPHA) for _ in range(rng.randint(0, 10)))
            idx = rng.randint(0, groups[0])
            try:
                got = native.extract_all_host(pat, idx, text)
            except Exception as e:  # noqa: BLE001
                assert "not supported" in str(e), (pat, str(e))
                break
            assert got == [m.group(idx) or "" for m in O.find_iter_like_the_crate(rx, text)], (pat, idx, text)
            checked += 1
    assert checked > 1200


def test_random_patterns_neither_crash_nor_hang():
    """patterns are user text inside plan bytes: whatever they are, planning answers (a program or a refusal by name) and the matcher terminates"""
    import time
    rng = random.Random(99)
    atoms = list("ab01.*+?|()[]{}^$\\-,:") + ["\\d", "\\w", "\\s", "\\b", "\\B", "[^", "(?:", "(?i)", "(?m)", "{2,}", "{0,3}", "é", "\\x41", "\\u00e9", "\\p{L}", "[[:alpha:]]", "*?", "+?", "\\z", "\\A"]
    texts = ["", "a", "ab01 ab", "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "é中 a-b,c:d\n", "((((", "0" * 64]
    t0 = time.time()
    ran = refused = 0
    for _ in range(6000):
        pat = "".join(rng.choice(atoms) for _ in range(rng.randint(0, 10)))
        for fn in (lambda v: native.regexp_extract_host(pat, rng.randint(0, 2), v), lambda v: native.split_host(pat, rng.choice([-1, 0, 2]), v), lambda v: native.extract_all_host(pat, 0, v)):
            try:
                for v in texts:
                    fn(v)
                ran += 1
            except Exception as e:  # noqa: BLE001
                assert isinstance(e, native.CometNativeException), (pat, type(e))
                refused += 1
    assert ran > 1000 and refused > 1000 and time.time() - t0 < 120
