#!/usr/bin/env python3
"""Re-flows the paragraphs of a Markdown file to a maximum line width (default 118), leaving headings, tables, code fences and the
indentation of list items alone (list items are re-flowed with a hanging indent).   tools/wrap_md.py [-w N] file.md [...]  (in place)"""
import re
import sys
import textwrap


def wrap(text, width):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+)", first)
        lead = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
        body = " ".join(l.strip() for l in para)
        if m:
            body = body[len(m.group(1).strip()) + 1:].strip() if body.startswith(m.group(1).strip()) else body
        hang = " " * len(lead)
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False) or [""])
        para.clear()

    for line in text.split("\n"):
        s = line.rstrip()
        if s.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(s)
            continue
        if fence or s.startswith("|") or s.startswith("#") or not s.strip() or re.match(r"^\s*[-=]{3,}\s*$", s):
            flush()
            out.append(s)
            continue
        if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", s) and para:
            flush()
        para.append(s)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    args = sys.argv[1:]
    width = 118
    if args and args[0] == "-w":
        width, args = int(args[1]), args[2:]
    for path in args:
        src = open(path).read()
        open(path, "w").write(wrap(src, width).rstrip("\n") + "\n")
