#!/usr/bin/env python3
"""Transcribes the reference's own string → timestamp vectors (native/spark-expr/src/conversion_funcs/string.rs, mod tests: every
`timestamp_parser(...)` / `timestamp_ntz_parser(...)` assertion) into tests/golden/timestamp_kats.json — run HERE, where /root/reference exists; the
tests read the JSON.  Each entry: [function, value, eval mode, zone, is_spark4_plus | allow_time_zone, expected] with expected an integer (µs), null,
"err", "some" or "none"."""
import json
import re
import sys

SRC = "/root/reference/native/spark-expr/src/conversion_funcs/string.rs"


def rust_str(s):
    s = re.sub(r"\\u\{([0-9a-fA-F]+)\}", lambda m: chr(int(m.group(1), 16)), s)
    return s.replace("\\t", "\t").replace("\\n", "\n").replace("\\r", "\r").replace('\\"', '"').replace("\\\\", "\\")


def main():
    text = open(SRC).read()
    text = text[text.index("mod tests {"):]
    out = []
    # split into test functions so that `let name = 123i64;` constants resolve locally
    for fn in re.split(r"\n    #\[test\]", text):
        consts = {m.group(1): int(m.group(2).replace("_", "")) for m in re.finditer(r"let (\w+)(?:: i64)? = (-?[0-9_]+)(?:i64)?;", fn)}
        zones = {m.group(1): m.group(2) for m in re.finditer(r'let (\w+) = &timezone::Tz::from_str\("([^"]+)"\)', fn)}
        call = r'(timestamp_parser|timestamp_ntz_parser)\(\s*"((?:[^"\\]|\\.)*)",\s*EvalMode::(\w+),\s*(\w+),\s*(\w+),?\s*\)'
        for m in re.finditer(call + r"\s*\.unwrap\(\),\s*(Some\(\s*([-\w]+)\s*\)|None)", fn):
            f, val, mode, a4, a5, exp, num = m.groups()
            if f == "timestamp_parser" and a4 not in zones:
                continue
            if exp == "None":
                e = None
            else:
                n = num.replace("_", "")
                n = re.sub(r"i64$", "", n)
                if n == "i64::MAX":
                    e = 2**63 - 1
                elif re.fullmatch(r"-?\d+", n):
                    e = int(n)
                elif n in consts:
                    e = consts[n]
                else:
                    continue
            out.append([f, rust_str(val), mode.lower(), zones.get(a4, a4), a5 == "true", e])
        for m in re.finditer(call + r"\s*\.is_err\(\)", fn):
            f, val, mode, a4, a5 = m.groups()
            out.append([f, rust_str(val), mode.lower(), zones.get(a4, a4), a5 == "true", "err"])
        for m in re.finditer(call + r"\s*\.unwrap\(\)\s*\.is_(none|some)\(\)", fn):
            f, val, mode, a4, a5, what = m.groups()
            out.append([f, rust_str(val), mode.lower(), zones.get(a4, a4), a5 == "true", what])
    # test_cast_string_to_timestamp_ntz: parse(s, allow_tz) = timestamp_ntz_parser(s, Legacy, allow_tz, false)
    for m in re.finditer(r'parse\("((?:[^"\\]|\\.)*)", (true|false)\),\s*(Some\(([\d_]+)\)|None)\s*\)', text):
        out.append(["timestamp_ntz_parser", rust_str(m.group(1)), "legacy", "true" if m.group(2) == "true" else "false", False, None if m.group(3) == "None" else int(m.group(4).replace("_", ""))])
    # loops over literal lists
    for ws in [" T2:30", "\tT2:30", "\nT2:30", " T2", "\tT2", "\nT2"]:
        out.append(["timestamp_parser", ws, "legacy", "UTC", True, "none"])
        out.append(["timestamp_parser", ws, "ansi", "UTC", True, "err"])
        out.append(["timestamp_parser", ws, "legacy", "UTC", False, "some"])
    for ok in ["T2:30", "T2"]:
        out.append(["timestamp_parser", ok, "legacy", "UTC", True, "some"])
    # Some(i64::MAX) / Some(i64::MIN) are written without parentheses-friendly digits
    for m in re.finditer(r'timestamp_parser\("([^"]+)", EvalMode::Legacy, tz, true\)\.unwrap\(\),\s*Some\(i64::(MAX|MIN)\)', text):
        out.append(["timestamp_parser", m.group(1), "legacy", "UTC", True, 2**63 - 1 if m.group(2) == "MAX" else -2**63])
    json.dump({"source": "string.rs mod tests, transcribed by tools/extract_timestamp_kats.py", "vectors": out}, sys.stdout, ensure_ascii=False, indent=0)


if __name__ == "__main__":
    main()
