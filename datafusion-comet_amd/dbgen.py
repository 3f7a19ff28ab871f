"""The columns nineteen of the twenty-two TPC-H queries read, as tpch-dbgen generates them — the data behind the reference's own golden answers
(spark/src/test/resources/tpch-query-results/q*.sql.out, written by CometTPCHQuerySuite over tables that GenTPCHData.scala:33-34 produces with
https://github.com/databricks/tpch-dbgen).  dbgen is not part of the reference's tree (and there is no network here), so this restates its
published algorithm (TPC-H tools 2.x: rnd.c NextRand / UnifInt, build.c mk_order / mk_cust / mk_supp / mk_part / rpb_routine / mk_sparse, bm_utils.c
a_rnd / agg_str / permute, the seed table of driver.c, dists.dss); it is pinned by the golden files themselves: the generated SF1 tables give exactly those
answers (tests/test_tpch_golden_cpu.py), down to the last digit of every sum and the last character of every printed address.  Not restated: comment text
(dbgen cuts it out of a pool of generated sentences) — which leaves out Q2, Q10 and Q13, the queries that print or match it; Q16's predicate on s_comment
is the outcome of two draws of its own and is generated as a Boolean column.

What dbgen does, per stream: a Park–Miller generator (seed ← seed · 16807 mod 2^31 − 1), a draw is lo + ⌊seed / (2^31 − 1) · (hi − lo + 1)⌋;
every column has its own stream, and after every ROW each stream is advanced to a fixed number of draws (its `boundary`: 7 for the lineitem
columns of an order, 1 for the order's own), so row i's draws start at seed₀ · 16807^(boundary · i) — which is what makes the generation
data-parallel here (numpy; modular powers by squaring)."""
from __future__ import annotations

import numpy as np
import pyarrow as pa

M = 2147483647
A = 16807
# driver.c Seed[]: stream → initial value
SEED = {"O_ODATE": 1066728069, "L_QTY": 209208115, "L_DCNT": 554590007, "L_TAX": 721958466, "L_PKEY": 1808217256, "L_SDTE": 1769349045,
        "L_CDTE": 904914315, "L_RDTE": 373135028, "L_RFLG": 717419739, "C_MSEG": 1140279430, "O_CKEY": 851767375, "O_LCNT": 1434868289,
        "L_SMODE": 675466456, "O_PRIO": 591449447, "P_TYPE": 1841581359, "L_SHIP": 1371272478, "P_MFG": 1, "P_BRND": 46831694, "P_SIZE": 1193163244,
        "P_CNTR": 727633698, "C_NTRG": 1489529863, "S_NTRG": 110356601, "L_SKEY": 2095021727, "C_PHNE": 1521138112, "C_ABAL": 298370230, "PS_QTY": 1671059989, "PS_SCST": 1051288424,
        "P_NAME": 709314158, "S_ADDR": 706178559, "S_PHNE": 884434366, "S_ABAL": 962338209, "BBB_CMNT": 202794285, "BBB_TYPE": 753643799}
STARTDATE_DAY = 8035          # 1992-01-01 as days since 1970-01-01 (dbgen's STARTDATE 92001)
CURRENTDATE_OFFSET = 1263     # 1995-06-17 (CURRENTDATE 95168) as days since 1992-01-01
SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]      # dists.dss msegmnt, equal weights
PRIORITIES = [b"1-URGENT", b"2-HIGH", b"3-MEDIUM", b"4-NOT SPECIFIED", b"5-LOW"]     # dists.dss o_oprio
# dists.dss smode, equal weights.  The golden answers pin MAIL and SHIP to the 5th and 7th entry (Q12) and AIR to the 2nd (Q19); the other four
# names are placed from memory of the file and no query here depends on which is which
SHIPMODES = [b"REG AIR", b"AIR", b"RAIL", b"TRUCK", b"MAIL", b"FOB", b"SHIP"]
INSTRUCTIONS = [b"DELIVER IN PERSON", b"COLLECT COD", b"NONE", b"TAKE BACK RETURN"]      # dists.dss instruct (Q19 pins the first)
# dists.dss p_cntr: 40 equally weighted names, size syllable outside (Q19's golden answer pins the twelve it names)
CONTAINERS = [a + b" " + b for a in (b"SM", b"LG", b"MED", b"JUMBO", b"WRAP") for b in (b"CASE", b"BOX", b"BAG", b"JAR", b"PKG", b"PACK", b"CAN", b"DRUM")]
# dists.dss p_types: 150 equally weighted names, the three syllables nested in this order (Q14's golden answer pins PROMO to the last 25)
TYPE_SYLLABLES = ([b"STANDARD", b"SMALL", b"MEDIUM", b"LARGE", b"ECONOMY", b"PROMO"], [b"ANODIZED", b"BURNISHED", b"PLATED", b"POLISHED", b"BRUSHED"],
                  [b"TIN", b"NICKEL", b"BRASS", b"STEEL", b"COPPER"])
PART_TYPES = [a + b" " + b + b" " + c for a in TYPE_SYLLABLES[0] for b in TYPE_SYLLABLES[1] for c in TYPE_SYLLABLES[2]]


def _mulmod(a: np.ndarray, b) -> np.ndarray:
    return (a * np.uint64(b)) % np.uint64(M) if np.isscalar(b) else (a * b) % np.uint64(M)


def _stream_starts(seed0: int, n_rows: int, boundary: int) -> np.ndarray:
    """the stream's value BEFORE row i's first draw, i = 0 … n_rows − 1: seed0 · A^(boundary · i) mod M"""
    step = pow(A, boundary, M)
    i = np.arange(n_rows, dtype=np.uint64)
    out = np.full(n_rows, seed0 % M, np.uint64)
    f = step
    for b in range(max(1, int(n_rows - 1).bit_length())):
        out = np.where((i >> np.uint64(b)) & np.uint64(1), _mulmod(out, f), out)
        f = f * f % M
    return out


def _draw(state: np.ndarray, lo: int, hi: int):
    """one draw per row: → (values, advanced state); UnifInt's double arithmetic"""
    state = _mulmod(state, A)
    return lo + ((state.astype(np.float64) / float(M)) * float(hi - lo + 1)).astype(np.int64), state


def _dec(values: np.ndarray, p: int, s: int) -> pa.Array:
    v = values.astype(np.int64)
    raw = np.empty(2 * len(v), np.int64)
    raw[0::2] = v
    raw[1::2] = v >> 63
    return pa.Array.from_buffers(pa.decimal128(p, s), len(v), [None, pa.py_buffer(raw.tobytes())])


def _utf8_from_choices(idx: np.ndarray, choices) -> pa.Array:
    lens = np.array([len(c) for c in choices], np.int32)
    offs = np.zeros(len(idx) + 1, np.int32)
    offs[1:] = np.cumsum(lens[idx])
    width = max(lens)
    table = np.zeros((len(choices), width), np.uint8)
    for k, c in enumerate(choices):
        table[k, :len(c)] = np.frombuffer(c, np.uint8)
    mask = np.arange(width)[None, :] < lens[idx][:, None]
    data = table[idx][mask].tobytes()
    return pa.Array.from_buffers(pa.utf8(), len(idx), [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(data)])


NATIONS = [(b"ALGERIA", 0), (b"ARGENTINA", 1), (b"BRAZIL", 1), (b"CANADA", 1), (b"EGYPT", 4), (b"ETHIOPIA", 0), (b"FRANCE", 3), (b"GERMANY", 3), (b"INDIA", 2),
           (b"INDONESIA", 2), (b"IRAN", 4), (b"IRAQ", 4), (b"JAPAN", 2), (b"JORDAN", 4), (b"KENYA", 0), (b"MOROCCO", 0), (b"MOZAMBIQUE", 0), (b"PERU", 1),
           (b"CHINA", 2), (b"ROMANIA", 3), (b"SAUDI ARABIA", 4), (b"VIETNAM", 2), (b"RUSSIA", 3), (b"UNITED KINGDOM", 3), (b"UNITED STATES", 1)]      # dists.dss nations
REGIONS = [b"AFRICA", b"AMERICA", b"ASIA", b"EUROPE", b"MIDDLE EAST"]


def nation() -> pa.Table:
    return pa.table([pa.array(np.arange(25, dtype=np.int32)), pa.array([n.decode() for n, _ in NATIONS]), pa.array(np.array([r for _, r in NATIONS], np.int32))],
                    names=["n_nationkey", "n_name", "n_regionkey"])


def region() -> pa.Table:
    return pa.table([pa.array(np.arange(5, dtype=np.int32)), pa.array([r.decode() for r in REGIONS])], names=["r_regionkey", "r_name"])


def customer(sf: int = 1) -> pa.Table:
    """c_custkey, c_mktsegment, c_nationkey, c_name, c_phone, c_acctbal (mk_cust: one draw of C_MSEG — pick_str over five equal weights — and of C_NTRG per customer;
    the name is "Customer#" and the key in nine digits)"""
    n = 150_000 * sf
    seg, _ = _draw(_stream_starts(SEED["C_MSEG"], n, 1), 1, 5)
    nat, _ = _draw(_stream_starts(SEED["C_NTRG"], n, 1), 0, 24)
    names = pa.array(["Customer#%09d" % k for k in range(1, n + 1)])
    # gen_phone: country code 10 + nation, then three draws of C_PHNE (area code, exchange, number); the balance: one draw of C_ABAL, in cents
    ph = _stream_starts(SEED["C_PHNE"], n, 3)
    area, ph = _draw(ph, 100, 999)
    exch, ph = _draw(ph, 100, 999)
    num, ph = _draw(ph, 1000, 9999)
    phones = pa.array(["%02d-%03d-%03d-%04d" % t for t in zip((10 + nat).tolist(), area.tolist(), exch.tolist(), num.tolist())])
    bal, _ = _draw(_stream_starts(SEED["C_ABAL"], n, 1), -99999, 999999)
    return pa.table([pa.array(np.arange(1, n + 1, dtype=np.int64)), _utf8_from_choices(seg - 1, SEGMENTS), pa.array(nat.astype(np.int32)), names, phones, _dec(bal, 12, 2)],
                    names=["c_custkey", "c_mktsegment", "c_nationkey", "c_name", "c_phone", "c_acctbal"])


# dists.dss colors: 92 equally weighted words, in the file's (alphabetical) order; Q9 / Q20's golden answers depend on where green and forest sit
COLORS = ("almond antique aquamarine azure beige bisque black blanched blue blush brown burlywood burnished chartreuse chiffon chocolate coral cornflower "
          "cornsilk cream cyan dark deep dim dodger drab firebrick floral forest frosted gainsboro ghost goldenrod green grey honeydew hot indian ivory khaki "
          "lace lavender lawn lemon light lime linen magenta maroon medium metallic midnight mint misty moccasin navajo navy olive orange orchid pale papaya "
          "peach peru pink plum powder puff purple red rose rosy royal saddle salmon sandy seashell sienna sky slate smoke snow spring steel tan thistle tomato "
          "turquoise violet wheat white yellow").split()
ALPHA_NUM = b"0123456789abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ,"      # a_rnd's alphabet, six bits per character


def _v_str(seed0: int, n: int, avg: int, boundary: int) -> pa.Array:
    """V_STR(avg): a_rnd(⌊0.4·avg⌋, ⌊1.6·avg⌋) — one draw for the length, then one draw of [0, MAX_LONG] per five characters, six bits each from the low end"""
    lo, hi = int(avg * 0.4), int(avg * 1.6)
    st = _stream_starts(seed0, n, boundary)
    ln, st = _draw(st, lo, hi)
    chars = np.zeros((n, hi), np.uint8)
    alpha = np.frombuffer(ALPHA_NUM, np.uint8)
    for g in range((hi + 4) // 5):
        v, st = _draw(st, 0, M)
        v = -v                 # UnifInt computes this one range, MAX_LONG − 0 + 1, in 32 bits: −2^31 — the draw is negative, shifted arithmetically
        for k in range(5):
            if 5 * g + k < hi:
                chars[:, 5 * g + k] = alpha[(v >> (6 * k)) & 63]
    mask = np.arange(hi)[None, :] < ln[:, None]
    offs = np.zeros(n + 1, np.int32)
    offs[1:] = np.cumsum(ln)
    return pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(chars[mask].tobytes())])


def supplier(sf: int = 1) -> pa.Table:
    """s_suppkey, s_nationkey, s_name, s_address, s_phone, s_acctbal, s_complaints (mk_supp: the address is V_STR(25) over S_ADDR — nine draws per supplier at most —,
    one draw of S_NTRG, gen_phone over S_PHNE, one draw of S_ABAL; the comment is text this module does not restate, but whether it carries "Customer … Complaints"
    — all Q16 asks of it — is two draws of their own: BBB_CMNT ≤ 10 of 10 000 marks the supplier, BBB_TYPE < 50 of 0 … 100 makes it Complaints rather than Recommends)"""
    n = 10_000 * sf
    nat, _ = _draw(_stream_starts(SEED["S_NTRG"], n, 1), 0, 24)
    ph = _stream_starts(SEED["S_PHNE"], n, 3)
    area, ph = _draw(ph, 100, 999)
    exch, ph = _draw(ph, 100, 999)
    num, ph = _draw(ph, 1000, 9999)
    phones = pa.array(["%02d-%03d-%03d-%04d" % t for t in zip((10 + nat).tolist(), area.tolist(), exch.tolist(), num.tolist())])
    bal, _ = _draw(_stream_starts(SEED["S_ABAL"], n, 1), -99999, 999999)
    bad, _ = _draw(_stream_starts(SEED["BBB_CMNT"], n, 1), 1, 10000)
    kind, _ = _draw(_stream_starts(SEED["BBB_TYPE"], n, 1), 0, 100)
    return pa.table([pa.array(np.arange(1, n + 1, dtype=np.int64)), pa.array(nat.astype(np.int32)), pa.array(["Supplier#%09d" % k for k in range(1, n + 1)]),
                     _v_str(SEED["S_ADDR"], n, 25, 9), phones, _dec(bal, 12, 2), pa.array((bad <= 10) & (kind < 50))],
                    names=["s_suppkey", "s_nationkey", "s_name", "s_address", "s_phone", "s_acctbal", "s_complaints"])


def part_names(sf: int = 1) -> pa.Array:
    """p_name: agg_str over the colors — the identity permutation of the 92 words is shuffled by 92 draws of P_NAME (swap entry i with a draw of [i, 91]) and the
    name is the first five words; entry i is final after step i, so five steps per part give the name"""
    n, nc = 200_000 * sf, len(COLORS)
    perm = np.tile(np.arange(nc, dtype=np.int16), (n, 1))
    st = _stream_starts(SEED["P_NAME"], n, nc)
    rows = np.arange(n)
    for i in range(5):
        src, st = _draw(st, i, nc - 1)
        a, b = perm[rows, src].copy(), perm[rows, i].copy()
        perm[rows, src], perm[rows, i] = b, a
    words = np.array(COLORS, dtype=object)
    first = words[perm[:, :5]]
    return pa.array([" ".join(r) for r in first.tolist()])


def part(sf: int = 1) -> pa.Table:
    """p_partkey, p_type, p_brand, p_container, p_size (mk_part: one draw per part of P_TYPE / P_MFG and P_BRND / P_CNTR / P_SIZE)"""
    n = 200_000 * sf
    t, _ = _draw(_stream_starts(SEED["P_TYPE"], n, 1), 1, 150)
    mfg, _ = _draw(_stream_starts(SEED["P_MFG"], n, 1), 1, 5)
    brnd, _ = _draw(_stream_starts(SEED["P_BRND"], n, 1), 1, 5)
    cntr, _ = _draw(_stream_starts(SEED["P_CNTR"], n, 1), 1, 40)
    size, _ = _draw(_stream_starts(SEED["P_SIZE"], n, 1), 1, 50)
    brands = [b"Brand#%d%d" % (m, b) for m in range(1, 6) for b in range(1, 6)]
    return pa.table([pa.array(np.arange(1, n + 1, dtype=np.int64)), _utf8_from_choices(t - 1, PART_TYPES), _utf8_from_choices((mfg - 1) * 5 + brnd - 1, brands),
                     _utf8_from_choices(cntr - 1, CONTAINERS), pa.array(size.astype(np.int32))], names=["p_partkey", "p_type", "p_brand", "p_container", "p_size"])


def partsupp(sf: int = 1) -> pa.Table:
    """ps_partkey, ps_suppkey, ps_availqty, ps_supplycost (mk_part: four suppliers per part — PART_SUPP_BRIDGE — with a draw of PS_QTY and of PS_SCST each)"""
    n, nsupp = 200_000 * sf, 10_000 * sf
    pk = np.arange(1, n + 1, dtype=np.int64)
    q, c = _stream_starts(SEED["PS_QTY"], n, 4), _stream_starts(SEED["PS_SCST"], n, 4)
    cols = [[], [], [], []]
    for snum in range(4):
        qty, q = _draw(q, 1, 9999)
        cost, c = _draw(c, 100, 100000)
        for k, v in enumerate((pk, (pk + snum * (nsupp // 4 + (pk - 1) // nsupp)) % nsupp + 1, qty, cost)):
            cols[k].append(v)
    order = np.lexsort((np.repeat(np.arange(4), n), np.tile(pk, 4)))
    cat = [np.concatenate(v)[order] for v in cols]
    return pa.table([pa.array(cat[0]), pa.array(cat[1]), pa.array(cat[2].astype(np.int32)), _dec(cat[3], 12, 2)], names=["ps_partkey", "ps_suppkey", "ps_availqty", "ps_supplycost"])


def orders_and_lineitem(sf: int = 1):
    """→ (orders[o_orderkey, o_custkey, o_orderdate, o_shippriority, o_orderpriority, o_totalprice, o_orderstatus],
          lineitem[l_orderkey, l_quantity, l_extendedprice, l_discount, l_tax, l_returnflag, l_linestatus, l_shipdate, l_partkey, l_commitdate,
                   l_receiptdate, l_shipmode, l_shipinstruct, l_suppkey]) in dbgen's row order"""
    n = 1_500_000 * sf
    ncust = 150_000 * sf
    i = np.arange(1, n + 1, dtype=np.int64)
    okey = ((i >> 3) << 5) | (i & 7)                                     # mk_sparse: the low 3 bits kept, 2 zero bits above them
    ckey, _ = _draw(_stream_starts(SEED["O_CKEY"], n, 1), 1, ncust)
    # every third customer places no orders (CUST_MORTALITY): step to a neighbour, alternating +1, −1 (once is always enough)
    dead = ckey % 3 == 0
    ckey = np.where(dead, np.minimum(ckey + 1, ncust), ckey)
    still = ckey % 3 == 0                                                  # (only when the step was clipped at the last key)
    ckey = np.where(still, ckey - 1, ckey)
    odate, _ = _draw(_stream_starts(SEED["O_ODATE"], n, 1), 92001, 94406)
    odate_off = odate - 92001                                              # days since 1992-01-01
    lines, _ = _draw(_stream_starts(SEED["O_LCNT"], n, 1), 1, 7)
    prio, _ = _draw(_stream_starts(SEED["O_PRIO"], n, 1), 1, 5)
    total = np.zeros(n, np.int64)                                         # o_totalprice, in cents: filled from the lines below
    shipped = np.zeros(n, np.int64)                                       # lines with status 'F': the order's status is 'F' if all, 'O' if none, else 'P'
    st = {k: _stream_starts(SEED[k], n, 7) for k in ("L_QTY", "L_DCNT", "L_TAX", "L_PKEY", "L_SDTE", "L_CDTE", "L_RDTE", "L_RFLG", "L_SMODE", "L_SHIP", "L_SKEY")}
    cols = {k: [] for k in ("okey", "qty", "ep", "disc", "tax", "rflag", "lstat", "ship", "order", "lcnt", "pkey", "commit", "receipt", "smode", "instr", "skey")}
    for l in range(7):
        qty, st["L_QTY"] = _draw(st["L_QTY"], 1, 50)
        disc, st["L_DCNT"] = _draw(st["L_DCNT"], 0, 10)
        tax, st["L_TAX"] = _draw(st["L_TAX"], 0, 8)
        pkey, st["L_PKEY"] = _draw(st["L_PKEY"], 1, 200_000 * sf)
        sd, st["L_SDTE"] = _draw(st["L_SDTE"], 1, 121)
        rd, st["L_RDTE"] = _draw(st["L_RDTE"], 1, 30)
        cd, st["L_CDTE"] = _draw(st["L_CDTE"], 30, 90)
        smode, st["L_SMODE"] = _draw(st["L_SMODE"], 1, 7)
        instr, st["L_SHIP"] = _draw(st["L_SHIP"], 1, 4)
        snum, st["L_SKEY"] = _draw(st["L_SKEY"], 0, 3)
        ship = odate_off + sd
        receipt = ship + rd
        commit = odate_off + cd
        # the return flag draws from its stream only when the line has been received by CURRENTDATE
        received = receipt <= CURRENTDATE_OFFSET
        has = lines > l
        rf_draw, adv = _draw(st["L_RFLG"], 1, 2)
        st["L_RFLG"] = np.where(received & has, adv, st["L_RFLG"])
        rflag = np.where(received, np.where(rf_draw == 1, 0, 1), 2)      # 0 'R', 1 'A', 2 'N'
        lstat = np.where(ship <= CURRENTDATE_OFFSET, 0, 1)                # 0 'F', 1 'O'
        price = 90000 + (pkey // 10) % 20001 + (pkey % 1000) * 100       # rpb_routine, in cents
        nsupp = 10_000 * sf
        skey = (pkey + snum * (nsupp // 4 + (pkey - 1) // nsupp)) % nsupp + 1     # PART_SUPP_BRIDGE: one of the part's four suppliers
        # mk_order: totalprice += ((eprice · (100 − discount)) / 100) · (100 + tax) / 100 — integer division, left to right
        total += np.where(has, ((price * qty * (100 - disc)) // 100) * (100 + tax) // 100, 0)
        shipped += np.where(has & (lstat == 0), 1, 0)
        for k, v in (("okey", okey), ("qty", qty), ("ep", price * qty), ("disc", disc), ("tax", tax), ("rflag", rflag), ("lstat", lstat), ("ship", ship),
                     ("order", i), ("lcnt", np.full(n, l, np.int64)), ("pkey", pkey), ("commit", commit), ("receipt", receipt), ("smode", smode - 1), ("instr", instr - 1), ("skey", skey)):
            cols[k].append(v[has])
    orders = pa.table([pa.array(okey), pa.array(ckey), pa.array((odate_off + STARTDATE_DAY).astype(np.int32), pa.int32()).cast(pa.date32()),
                       pa.array(np.zeros(n, np.int32)), _utf8_from_choices(prio - 1, PRIORITIES), _dec(total, 12, 2),
                       _utf8_from_choices(np.where(shipped == lines, 0, np.where(shipped == 0, 1, 2)), [b"F", b"O", b"P"])],
                      names=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority", "o_orderpriority", "o_totalprice", "o_orderstatus"])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    order = np.lexsort((cat["lcnt"], cat["order"]))                        # dbgen's row order: by order, then line number
    c = {k: v[order] for k, v in cat.items()}
    lineitem = pa.table([pa.array(c["okey"]), _dec(c["qty"] * 100, 12, 2), _dec(c["ep"], 12, 2), _dec(c["disc"], 12, 2), _dec(c["tax"], 12, 2),
                         _utf8_from_choices(c["rflag"], [b"R", b"A", b"N"]), _utf8_from_choices(c["lstat"], [b"F", b"O"]),
                         pa.array((c["ship"] + STARTDATE_DAY).astype(np.int32), pa.int32()).cast(pa.date32()), pa.array(c["pkey"]),
                         pa.array((c["commit"] + STARTDATE_DAY).astype(np.int32), pa.int32()).cast(pa.date32()),
                         pa.array((c["receipt"] + STARTDATE_DAY).astype(np.int32), pa.int32()).cast(pa.date32()), _utf8_from_choices(c["smode"], SHIPMODES),
                         _utf8_from_choices(c["instr"], INSTRUCTIONS), pa.array(c["skey"])],
                        names=["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate", "l_partkey",
                               "l_commitdate", "l_receiptdate", "l_shipmode", "l_shipinstruct", "l_suppkey"])
    return orders, lineitem


def parse_golden(path: str):
    """rows of a `*.sql.out` file of the reference's TPC-H suite (tab-separated, after the `-- !query output` line)"""
    rows, on = [], False
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("-- !query output"):
            on = True
        elif on and line:
            rows.append(line.split("\t"))
    return rows
