#!/usr/bin/env python3
"""Keep the design documents readable in a terminal and in a diff: wrap paragraph and list lines at WIDTH characters and turn tables whose rows are longer than
LIMIT into bullet lists (one bullet per row, `header: cell` pairs), leaving code fences, headings and short tables alone.

    python tools/reflow_md.py DESIGN.md docs/*.md        (in place)"""
import re
import sys
import textwrap

WIDTH, LIMIT = 150, 200


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, esc, tick = [], "", False, False
    for ch in row:      # split on | outside `code` and not escaped
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick and not esc:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        esc = ch == "\\"
    out.append(cur.strip())
    return out


def wrap(text, first, rest):
    return textwrap.fill(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def table_to_bullets(rows):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        lead = c[0] if c and c[0] else "-"
        parts = []
        for k in range(1, len(c)):
            if c[k]:
                h = head[k] if k < len(head) and head[k] else ""
                parts.append(("%s: %s" % (h, c[k])) if h else c[k])
        text = ("**%s**" % lead.strip("*") if lead != "-" else "") + (" — " if parts and lead != "-" else "") + "; ".join(parts)
        out.append(wrap(text, "- ", "  "))
    return out


def reflow(src):
    lines = src.split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l); i += 1; continue
        if fence or l.startswith("#") or not l.strip():
            out.append(l); i += 1; continue
        if l.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            if max(len(r) for r in rows) > LIMIT and len(rows) >= 3 and re.match(r"^\s*\|?[\s:|-]+\|?\s*$", rows[1]):
                out.extend(table_to_bullets(rows))
                out.append("")
            else:
                out.extend(rows)
            i = j; continue
        if len(l) <= LIMIT:
            out.append(l); i += 1; continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", l)
        if m:
            first = l[:m.end()]
            out.append(wrap(l[m.end():], first, " " * len(first)))
        else:
            ind = re.match(r"^\s*", l).group(0)
            out.append(wrap(l.strip(), ind, ind))
        i += 1
    return "\n".join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        with open(path) as f:
            src = f.read()
        res = reflow(src)
        with open(path, "w") as f:
            f.write(res)
        print(path, len(src), "->", len(res), "bytes, longest line", max(len(x) for x in res.split("\n")))
