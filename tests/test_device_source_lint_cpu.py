"""A cross-lane operation must be executed by every lane it reads from.  As the right operand of `&&` / `||` or an arm of `?:` it is executed only by the lanes
that get that far, and then reads lanes that sit the instruction out — undefined, and compiler-dependent in practice: `same = same && __shfl_up(valid, 1) != 0`
was hoisted by ROCm 7.0's compiler and not by ROCm 7.2's, which lost join build rows (profiles/r6_jit_compiler.md).  This check keeps the pattern out of the
device sources: in every statement, no wave intrinsic after a short-circuit or conditional operator."""
import glob
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "datafusion-comet_amd", "csrc")
_WAVE = r"(?:__shfl(?:_up|_down|_xor)?|__ballot|__any|__all|shfl_xor_u64|__reduce_\w+|__builtin_amdgcn_(?:ds_bpermute|ds_permute|readlane|readfirstlane|mov_dpp|update_dpp)\w*)"
_BAD = re.compile(r"(&&|\|\||\?)[^;{}]*?\b" + _WAVE + r"\s*\(")


def _statements(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    pos = 0
    for m in re.finditer(r"[;{}]", text):
        yield text[pos:m.start()], text.count("\n", 0, pos) + 1
        pos = m.end()


def _inside_call_of_wave_op(stmt, op_pos):
    """the operator sits INSIDE the argument list of a wave intrinsic (`__ballot(a && b)`): every lane evaluates it, fine"""
    depth = 0
    for i in range(op_pos - 1, -1, -1):
        c = stmt[i]
        if c == ")":
            depth += 1
        elif c == "(":
            if depth == 0:
                return re.search(_WAVE + r"\s*$", stmt[:i]) is not None
            depth -= 1
    return False


def test_no_wave_intrinsic_behind_a_short_circuit_operator():
    bad = []
    files = sorted(glob.glob(os.path.join(_CSRC, "device", "*.hpp")) + glob.glob(os.path.join(_CSRC, "*.hip")))
    assert len(files) > 8
    for f in files:
        for stmt, line in _statements(open(f).read()):
            for m in _BAD.finditer(stmt):
                if _inside_call_of_wave_op(stmt, m.start()):
                    continue
                # `for (…; cond; …)` / `while (cond)` headers and template angle brackets are not what this is about; a real hit has the intrinsic to the
                # RIGHT of the operator within one expression
                bad.append(f"{os.path.basename(f)}:{line}: {' '.join(stmt.split())[:160]}")
    assert not bad, "\n".join(bad)


def test_the_check_sees_the_pattern_it_is_for():
    hit = "same = same && __shfl_up(valid ? 1 : 0, 1, kWave) != 0"
    assert any(not _inside_call_of_wave_op(hit, m.start()) for m in _BAD.finditer(hit))
    fine = "mine += (u32)__popcll(__ballot(valid && !same))"
    assert all(_inside_call_of_wave_op(fine, m.start()) for m in _BAD.finditer(fine))
    fine2 = "const int prev_valid = __shfl_up(valid ? 1 : 0, 1, kWave)"
    assert all(_inside_call_of_wave_op(fine2, m.start()) for m in _BAD.finditer(fine2))
