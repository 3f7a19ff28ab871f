"""csrc/device/strfn.hpp — the source the derived-column kernels and the fused kernels run — on the host (comet_strfn_host): the digests against hashlib at every
padding boundary, crc32 against zlib, reverse / repeat / replace / substring_index / instr / ascii against Python's str methods and the oracle's restatement."""
import hashlib
import random
import zlib

from datafusion_comet_amd import native

sf = native.strfn_host


def test_digests_at_every_padding_boundary():
    rng = random.Random(3)
    for n in list(range(0, 140)) + [255, 256, 257, 1000, 4097]:
        v = bytes(rng.randrange(256) for _ in range(n))
        assert sf(10, v) == hashlib.md5(v).hexdigest().encode(), n
        assert sf(11, v) == hashlib.sha1(v).hexdigest().encode(), n
        assert sf(12, v) == hashlib.sha224(v).hexdigest().encode(), n
        assert sf(13, v) == hashlib.sha256(v).hexdigest().encode(), n
        assert sf(14, v) == hashlib.sha384(v).hexdigest().encode(), n
        assert sf(15, v) == hashlib.sha512(v).hexdigest().encode(), n
        assert sf(20, v) == zlib.crc32(v), n
    # the published test vectors (RFC 1321 A.5, FIPS 180-4 examples)
    assert sf(10, b"abc") == b"900150983cd24fb0d6963f7d28e17f72" and sf(11, b"abc") == b"a9993e364706816aba3e25717850c26c9cd0d89d"
    assert sf(13, b"abc") == b"ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert sf(15, b"")[:32] == b"cf83e1357eefb8bdf1542850d66d8007"


def test_string_functions_against_python():
    from oracle import oracle as O
    from datafusion_comet_amd import serde as S
    import pyarrow as pa
    rng = random.Random(4)
    alpha = ["a", "b", "x", ".", ",", " ", "é", "日", "aa"]
    words = ["", "a", "abc", "aaa", "日本語", "a,b,,c", "héllo wörld", "xxyxx", "www.apache.org"] + ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 12))) for _ in range(300)]
    for w in words:
        b = w.encode()
        assert sf(1, b).decode() == w[::-1]
        assert sf(2, b, k=3).decode() == w * 3 and sf(2, b, k=0) == b""
        for f_, t_ in [("a", "Z"), ("", "-"), ("aa", "b"), ("本", "xx"), (",,", ","), ("x", ""), (".", "..")]:
            assert sf(3, b, f_.encode(), t_.encode()).decode() == w.replace(f_, t_), (w, f_, t_)
        assert sf(21, b, b"b") == (w.find("b") + 1) and sf(21, b, "é".encode()) == (w.find("é") + 1)
        assert sf(22, b) == (ord(w[0]) if w else 0)
    # substring_index against the oracle's restatement (split / rsplit without overlap), through the oracle's own evaluator
    t = pa.table({"s": pa.array(words)})
    for d, c in [(".", 2), (".", -2), (",", 1), (",", -1), ("a", 5), ("aa", 1), ("aa", 2), ("aa", -1), ("", 1), (".", 0), ("x", -2), ("日", 1), ("a", -3)]:
        e = S.scalar_func("substring_index", [S.col(0, S.T_STRING), S.lit(d, S.T_STRING), S.lit(c, S.T_INT64)], S.T_STRING)
        want = O.run_plan_to_arrow(S, S.project(S.scan([S.T_STRING]), [e]), t).column(0).to_pylist()
        assert [sf(4, w.encode(), d.encode(), k=c).decode() for w in words] == want, (d, c)
    # Spark's documented answers
    assert sf(4, b"www.apache.org", b".", k=2) == b"www.apache" and sf(4, b"www.apache.org", b".", k=-2) == b"apache.org"
    assert sf(3, b"ABCabc", b"abc", b"DEF") == b"ABCDEF" and sf(21, b"SparkSQL", b"SQL") == 6 and sf(22, b"222") == 50 and sf(1, b"Spark SQL") == b"LQS krapS"
