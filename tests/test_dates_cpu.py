"""The calendar functions the generated kernels call (csrc/device/dates.hpp), run on the host through comet_date_fn_host, against Python's datetime and
the reference's own vectors (datetime_funcs/next_day.rs:165-190, make_date.rs:170-215; Spark's documented weekday / weekofyear / last_day / trunc answers)."""
import datetime as dt
import random

from datafusion_comet_amd import native

E = dt.date(1970, 1, 1)
D = lambda y, m, d: (dt.date(y, m, d) - E).days
f = native.date_fn_host


def test_reference_and_documented_vectors():
    assert f(1, D(2009, 7, 30)) - 1 == 3                 # Spark: weekday('2009-07-30') = 3 (isodow − 1, datetime.scala CometWeekDay)
    assert f(2, D(2008, 2, 20)) == 8                     # Spark: weekofyear('2008-02-20') = 8
    assert f(4, D(2009, 1, 12)) == D(2009, 1, 31)        # Spark: last_day('2009-01-12')
    assert f(3, D(2019, 8, 4), 3) == D(2019, 7, 29) and f(3, D(2019, 8, 4), 1) == D(2019, 7, 1) and f(3, D(2009, 2, 12), 2) == D(2009, 2, 1) and f(3, D(2015, 10, 27), 0) == D(2015, 1, 1)   # Spark: trunc(...)
    assert f(5, D(2024, 1, 1), 0) == D(2024, 1, 8) and f(5, D(2024, 1, 1), 1) == D(2024, 1, 2)      # next_day.rs test_next_date_for_day_of_week
    assert f(5, D(2015, 1, 14), 1) == D(2015, 1, 20)     # Spark: next_day('2015-01-14', 'TU')
    assert (f(6, 1970, 1, 1), f(6, 1970, 1, 2), f(6, 1969, 12, 31)) == (0, 1, -1)                    # make_date.rs test_make_date_valid
    assert f(6, 2000, 2, 29) is not None and f(6, 2004, 2, 29) is not None and f(6, 2023, 6, 15) is not None
    assert f(6, 2023, 0, 15) is None and f(6, 2023, 13, 1) is None and f(6, 2023, 2, 29) is None and f(6, 2023, 4, 31) is None and f(6, 1900, 2, 29) is None and f(6, 2023, 6, 0) is None
    assert f(6, 262143, 1, 1) is None and f(6, -262144, 1, 1) is None and f(6, 262142, 12, 31) == 95026236 and f(6, -262143, 1, 1) == -96465292      # chrono's years
    assert f(3, 95026237, 0) is None and f(4, -96465293) is None and f(4, -96465292) is not None


def test_against_pythons_calendar():
    rng = random.Random(5)
    lo, hi = D(1, 1, 1), D(9999, 12, 24)
    days = [rng.randint(lo, hi) for _ in range(20000)] + [rng.randint(D(1890, 1, 1), D(2110, 1, 1)) for _ in range(20000)] + [lo, hi, 0, -1, 1]
    for x in days:
        d = E + dt.timedelta(days=x)
        assert f(0, x, 0) == d.year and f(0, x, 1) == d.month and f(0, x, 2) == d.day and f(0, x, 3) == (d.month - 1) // 3 + 1
        assert f(0, x, 4) == (d.weekday() + 1) % 7 and f(0, x, 5) == d.timetuple().tm_yday
        assert f(1, x) == d.isoweekday() and f(2, x) == d.isocalendar()[1]
        assert f(3, x, 0) == D(d.year, 1, 1) and f(3, x, 1) == D(d.year, (d.month - 1) // 3 * 3 + 1, 1) and f(3, x, 2) == D(d.year, d.month, 1) and f(3, x, 3) == x - d.weekday()
        nm = dt.date(d.year + (d.month == 12), d.month % 12 + 1, 1) if d.year < 9999 or d.month < 12 else None
        if nm:
            assert f(4, x) == (nm - E).days - 1
        t = rng.randrange(7)
        assert f(5, x, t) == x + 7 - (d.weekday() - t) % 7
        assert f(6, d.year, d.month, d.day) == x


def test_timestamp_trunc_on_the_wall_clock():
    rng = random.Random(6)
    U = dt.datetime(1970, 1, 1)
    for _ in range(5000):
        us = rng.randint(-6 * 10**15, 6 * 10**15)
        t = U + dt.timedelta(microseconds=us)
        back = lambda x: (x - U) // dt.timedelta(microseconds=1)
        want = [back(t.replace(month=1, day=1, hour=0, minute=0, second=0, microsecond=0)), back(t.replace(month=(t.month - 1) // 3 * 3 + 1, day=1, hour=0, minute=0, second=0, microsecond=0)),
                back(t.replace(day=1, hour=0, minute=0, second=0, microsecond=0)), back((t - dt.timedelta(days=t.weekday())).replace(hour=0, minute=0, second=0, microsecond=0)),
                back(t.replace(hour=0, minute=0, second=0, microsecond=0)), back(t.replace(minute=0, second=0, microsecond=0)), back(t.replace(second=0, microsecond=0)), back(t.replace(microsecond=0)),
                back(t.replace(microsecond=t.microsecond // 1000 * 1000)), us]
        assert [f(7, us, u) for u in range(10)] == want, us
