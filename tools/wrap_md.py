#!/usr/bin/env python3
"""Reflows a Markdown file's paragraphs and list items at 120 columns (tables, headings, code blocks and blank lines stay as they are).
Usage: wrap_md.py FILE [width]"""
import re
import sys
import textwrap


def reflow(text, width=120):
    out, block, in_code = [], [], False
    marker = re.compile(r"^(\s*)([-*]|\d+\.)\s+")

    def flush():
        if not block:
            return
        first = block[0]
        m = marker.match(first)
        if m:
            head, sub = m.group(0), " " * len(m.group(0))
            body = first[len(head):]
        else:
            ind = re.match(r"^\s*", first).group(0)
            head, sub, body = ind, ind, first[len(ind):]
        body = " ".join([body.strip()] + [x.strip() for x in block[1:]])
        w = textwrap.TextWrapper(width=width, initial_indent=head, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False)
        out.extend(w.wrap(body) or [first])
        block.clear()

    for line in text.split("\n"):
        fence = line.lstrip().startswith("```")
        special = in_code or fence or not line.strip() or line.lstrip().startswith("|") or line.startswith("#")
        if special:
            flush()
            out.append(line)
            if fence:
                in_code = not in_code
            continue
        if marker.match(line):
            flush()
        block.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path).read()
    dst = reflow(src, width)
    assert src.split() == dst.split(), "reflow changed the text"
    open(path, "w").write(dst)
