"""GPU parity for the exchange partitioner: Spark murmur3 (seed 42, chained over key columns)
+ pmod, through the C ABI on device buffers, against the reference's own KATs and the oracle."""
import ctypes
import json
import os

import numpy as np
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def _hash(type_id, values_np, valid=None, precision=0, aux=None, seed_hashes=None, n=None):
    import torch
    n = len(values_np) if n is None else n
    h = torch.full((n,), 42, dtype=torch.int32, device="cuda:0") if seed_hashes is None else seed_hashes
    v = _dev(values_np)
    vb = None
    if valid is not None:
        vb = _dev(np.packbits(np.array(valid, bool), bitorder="little"))
    ax = _dev(aux) if aux is not None else None
    rc = native.lib().comet_murmur3_column(type_id, precision, v.data_ptr(), vb.data_ptr() if vb is not None else None,
                                           ax.data_ptr() if ax is not None else None, n, h.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    return h


def _u32(h):
    return [int(x) & 0xFFFFFFFF for x in h.cpu().numpy()]


def test_reference_kats(built):
    k = K["murmur3"]
    assert _u32(_hash(S.INT8, np.array(k["i8"]["values"], np.int8))) == k["i8"]["expected"]
    assert _u32(_hash(S.INT32, np.array(k["i32"]["values"], np.int32))) == k["i32"]["expected"]
    assert _u32(_hash(S.INT64, np.array(k["i64"]["values"], np.int64))) == k["i64"]["expected"]
    assert _u32(_hash(S.FLOAT, np.array(k["f32"]["values"], np.float32))) == k["f32"]["expected"]
    assert _u32(_hash(S.DOUBLE, np.array(k["f64"]["values"], np.float64))) == k["f64"]["expected"]
    enc = [s.encode() for s in k["str"]["values"]]
    offs = np.zeros(len(enc) + 1, np.int32)
    offs[1:] = np.cumsum([len(e) for e in enc])
    data = np.frombuffer(b"".join(enc), np.uint8)
    assert _u32(_hash(S.STRING, offs, aux=data, n=len(enc))) == k["str"]["expected"]


def test_nulls_leave_seed_and_pmod_kat(built):
    import torch
    vals = np.array([0, 1, 0, -1, 0], np.int64)
    valid = [False, True, True, True, False]
    h = _u32(_hash(S.INT64, vals, valid))
    assert h[0] == 42 and h[4] == 42 and h[1] == 0x99f0149d and h[2] == 0x9c67b85d and h[3] == 0xc8008529
    k = K["pmod"]
    hh = torch.tensor(np.array(k["hashes"], np.uint32).view(np.int32), device="cuda:0")
    out = torch.zeros(len(k["hashes"]), dtype=torch.int32, device="cuda:0")
    assert native.lib().comet_pmod_partition(hh.data_ptr(), len(k["hashes"]), k["n"], out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert out.cpu().tolist() == k["expected"]


def test_multi_column_chain_matches_oracle(built):
    """Q3-style exchange key (int64, date32, decimal(12,2), decimal(38,6)) chained like create_murmur3_hashes."""
    from oracle import oracle as O
    n = 200_000
    rng = np.random.default_rng(17)
    k0 = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    k1 = rng.integers(8000, 11000, n, dtype=np.int64).astype(np.int32)
    d_small = O.i64_to_dec(rng.integers(-10**11, 10**11, n, dtype=np.int64))
    d_wide = O.ints_to_dec([int(x) * 10**20 for x in rng.integers(-10**15, 10**15, 2000)])
    valid1 = rng.random(n) < 0.9
    want = np.full(n, 42, np.uint32)
    O.C.o_murmur3_i64(O._p(k0), None, ctypes.c_int64(n), O._p(want))
    vb = valid1.astype(np.uint8)
    O.C.o_murmur3_i32(O._p(k1), O._p(vb), ctypes.c_int64(n), O._p(want))
    O.C.o_murmur3_decimal(O._p(d_small), 12, None, ctypes.c_int64(n), O._p(want))
    h = _hash(S.INT64, k0)
    h = _hash(S.DATE, k1, valid1, seed_hashes=h)
    h = _hash(S.DECIMAL, d_small, precision=12, seed_hashes=h, n=n)
    assert np.array_equal(h.cpu().numpy().view(np.uint32), want)
    want2 = np.full(2000, 42, np.uint32)
    O.C.o_murmur3_decimal(O._p(d_wide), 38, None, ctypes.c_int64(2000), O._p(want2))
    assert np.array_equal(_hash(S.DECIMAL, d_wide, precision=38, n=2000).cpu().numpy().view(np.uint32), want2)
    # partition ids over 8 GPUs
    import torch
    out = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    native.lib().comet_pmod_partition(h.data_ptr(), n, 8, out.data_ptr(), None)
    torch.cuda.synchronize()
    want_p = np.zeros(n, np.int32)
    O.C.o_pmod_array(O._p(want), ctypes.c_int64(n), 8, O._p(want_p))
    assert np.array_equal(out.cpu().numpy(), want_p)
    assert 0 <= out.min().item() and out.max().item() < 8


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 256, 1027, 100_003])
def test_utf8_uniform_length_check(built, n):
    """The per-chunk check that lets the fused kernels address fixed-length strings directly: exact for every n, for a deviation at
    any position (first / last / lane boundaries), and for offset buffers that are not 16-byte aligned."""
    import ctypes
    import torch
    lib = native.lib()
    lib.comet_launch_utf8_uniform.restype = ctypes.c_int
    lib.comet_launch_utf8_uniform.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L = 3
    base = np.arange(n + 1, dtype=np.int32) * L

    def check(offs, shift):
        buf = torch.zeros(n + 1 + 4, dtype=torch.int32, device="cuda")
        buf[shift:shift + n + 1] = torch.from_numpy(offs).cuda()
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        assert lib.comet_launch_utf8_uniform(buf.data_ptr() + 4 * shift, n, L, flag.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        return int(flag.item()) == 0

    for shift in (0, 1):
        assert check(base, shift)
        for pos in sorted({0, n - 1, n // 2, min(63, n - 1), min(64, n - 1), min(255, n - 1), max(0, n - 2)}):
            bad = base.copy()
            bad[pos + 1:] += 1            # value `pos` is one byte longer
            assert not check(bad, shift), (pos, shift)
