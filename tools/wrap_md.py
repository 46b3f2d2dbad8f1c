#!/usr/bin/env python3
"""Re-wraps the prose of a Markdown file at a readable width (default 110 columns): paragraphs and bullet items are
re-flowed, continuation lines of a bullet are indented under its text; headings, tables, code fences and blank lines
are left alone; tables whose rows exceed the width are turned into bullet lists (first cell bold).
Usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 110
lines = open(path).read().split("\n")
out, i = [], 0
bullet = re.compile(r"^(\s*)([*-]|\d+\.)\s+")


def flow(text, first, rest):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False).split("\n")


while i < len(lines):
    ln = lines[i]
    if ln.startswith("```"):
        out.append(ln)
        i += 1
        while i < len(lines) and not lines[i].startswith("```"):
            out.append(lines[i])
            i += 1
        if i < len(lines):
            out.append(lines[i])
            i += 1
        continue
    if ln.startswith("|"):
        tbl = []
        while i < len(lines) and lines[i].startswith("|"):
            tbl.append(lines[i])
            i += 1
        if max(len(t) for t in tbl) <= max(width, 140):
            out += tbl
        else:
            rows = [[c.strip() for c in t.strip().strip("|").split("|")] for t in tbl]
            head = rows[0]
            for r in rows[2:]:
                cells = [c for c in r[1:] if c]
                label = r[0] if r[0].startswith("**") else "**%s**" % r[0]
                if len(head) > 2:
                    body = "; ".join("%s: %s" % (h, c) if h else c for h, c in zip(head[1:], r[1:]) if c)
                else:
                    body = " ".join(cells)
                out += flow("%s — %s" % (label, body), "* ", "  ")
        continue
    if ln.strip() == "" or ln.startswith("#") or ln.startswith("<") or ln.startswith("    "):
        out.append(ln)
        i += 1
        continue
    m = bullet.match(ln)
    if m:
        indent = m.group(1)
        mark = m.group(2)
        text = [ln[m.end():]]
        i += 1
        while i < len(lines) and lines[i].strip() and not bullet.match(lines[i]) \
                and not lines[i].startswith(("#", "|", "```")) and lines[i].startswith(indent + " "):
            text.append(lines[i].strip())
            i += 1
        out += flow(" ".join(text), "%s%s " % (indent, mark), indent + " " * (len(mark) + 1))
        continue
    para = [ln.strip()]
    i += 1
    while i < len(lines) and lines[i].strip() and not bullet.match(lines[i]) \
            and not lines[i].startswith(("#", "|", "```", "    ")):
        para.append(lines[i].strip())
        i += 1
    out += flow(" ".join(para), "", "")
open(path, "w").write("\n".join(out))
