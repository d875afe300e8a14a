#!/usr/bin/env python3
"""Reflow a markdown file to a column limit without changing its content: paragraphs and list items are re-wrapped (hanging indent for
items), code fences and headings are left alone, table rows are either left alone or (--tables-to-lists) turned into bullet items
"* cell 1 — cell 2 — ..." so that no line exceeds the limit.        python tools/reflow_md.py FILE [--width 118] [--tables-to-lists]"""
import re
import sys
import textwrap


def wrap(text, width, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def main():
    path = sys.argv[1]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 118
    t2l = "--tables-to-lists" in sys.argv
    out, para, fence = [], [], False
    item_re = re.compile(r"^(\s*)([*+-]|\d+\.|\([a-z0-9]+\))\s+")

    def flush():
        if not para:
            return
        m = item_re.match(para[0])
        if m:
            lead = para[0][:m.end()]
            out.append(wrap(para[0][m.end():] + " " + " ".join(para[1:]), width, lead, " " * len(lead)))
        else:
            ind = re.match(r"^\s*", para[0]).group(0)
            out.append(wrap(" ".join(para), width, ind, ind))
        para.clear()

    for line in open(path).read().split("\n"):
        if line.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
        elif fence or line.startswith("#"):
            flush()
            out.append(line)
        elif not line.strip():
            flush()
            out.append("")
        elif line.lstrip().startswith("|"):
            flush()
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            if not t2l or len(line) <= width:
                out.append(line)
            elif all(set(c) <= set("-: ") for c in cells):
                continue
            else:
                out.append(wrap(" — ".join(c for c in cells if c), width, "* ", "  "))
        elif item_re.match(line):
            flush()
            para.append(line)
        else:
            para.append(line)
    flush()
    open(path, "w").write("\n".join(out))
    long_ = [i + 1 for i, ln in enumerate(out) if len(ln) > width + 2]
    print(path, len(out), "lines;", len(long_), "lines over the limit", long_[:10])


if __name__ == "__main__":
    main()
