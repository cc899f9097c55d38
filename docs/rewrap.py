#!/usr/bin/env python3
"""Re-wraps a markdown file to a readable width: paragraphs and list items are wrapped (hanging indent kept), code
fences are left alone, and table rows with cells too long to read in a table are turned into definition-style bullets.
usage: rewrap.py IN OUT [width] [longest table cell kept in a table]"""
import re, sys, textwrap

src, dst = sys.argv[1], sys.argv[2]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 110
CELL = int(sys.argv[4]) if len(sys.argv) > 4 else 160
lines = open(src).read().split("\n")
out, i, in_code = [], 0, False

def wrap(text, first, rest):
    return textwrap.fill(" ".join(text.split()), width=W, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False).split("\n")

def flush_table(rows):
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    body = [c for c in cells if not all(re.fullmatch(r":?-+:?", x or "-") for x in c)]
    if max(len(x) for c in body for x in c) <= CELL:
        return rows
    head, res = body[0], []
    for c in body[1:]:
        res += wrap("**" + c[0] + "**", "* ", "  ")
        for h, x in zip(head[1:], c[1:]):
            if x:
                res += wrap("*" + h + ":* " + x, "  - ", "    ")
    return res + [""]

while i < len(lines):
    ln = lines[i]
    if ln.lstrip().startswith("```"):
        in_code = not in_code; out.append(ln); i += 1; continue
    if in_code or not ln.strip() or ln.startswith("#"):
        out.append(ln); i += 1; continue
    if ln.lstrip().startswith("|"):
        rows = []
        while i < len(lines) and lines[i].lstrip().startswith("|"):
            rows.append(lines[i]); i += 1
        out += flush_table(rows); continue
    m = re.match(r"^(\s*)([*\-+]|\d+\.)\s+", ln)
    indent = (m.group(0), " " * len(m.group(0))) if m else (re.match(r"^\s*", ln).group(0),) * 2
    text = ln[len(m.group(0)):] if m else ln.strip()
    i += 1
    # continuation lines of the same paragraph / item
    while i < len(lines) and lines[i].strip() and not lines[i].lstrip().startswith(("|", "```", "#")) \
            and not re.match(r"^\s*([*\-+]|\d+\.)\s+", lines[i]):
        text += " " + lines[i].strip(); i += 1
    out += wrap(text, indent[0], indent[1])
open(dst, "w").write("\n".join(out))
