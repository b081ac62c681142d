#!/usr/bin/env python3
"""tools/wrap_md.py FILE [width=140] -- keep a Markdown file within `width` columns: paragraphs and list items are re-wrapped (code fences,
headings and tables that fit are left alone); a table with a row that does not fit is rewritten as a list, one item per row:
`* **first cell** — header2: cell2; header3: cell3`.  Idempotent."""
import re
import sys
import textwrap


def cells(row):
    return [c.strip() for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]


def wrap_block(text, width, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def main(path, width=140):
    lines = open(path).read().split("\n")
    out, i, n = [], 0, len(lines)
    bullet = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")
    while i < n:
        ln = lines[i]
        if ln.startswith("```"):
            j = i + 1
            while j < n and not lines[j].startswith("```"):
                j += 1
            out += lines[i:j + 1]; i = j + 1; continue
        if ln.startswith("|") and i + 1 < n and re.match(r"^\|\s*:?-+", lines[i + 1]):
            j = i
            while j < n and lines[j].startswith("|"):
                j += 1
            tbl = lines[i:j]
            if max(len(r) for r in tbl) <= width:
                out += tbl
            else:
                head = cells(tbl[0])
                for r in tbl[2:]:
                    c = cells(r)
                    parts = []
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            parts.append(("%s: %s" % (h, v)) if len(head) > 2 else v)
                    first = c[0] if c[0].startswith("**") else "**%s**" % c[0]
                    out.append(wrap_block("%s — %s" % (first, "; ".join(parts)), width, "* ", "  "))
            i = j; continue
        if not ln.strip() or ln.startswith("#") or ln.startswith("|"):
            out.append(ln); i += 1; continue
        m = bullet.match(ln)
        if m:
            ind = m.group(0)
            body = [ln[len(ind):]]
            j = i + 1
            while j < n and lines[j].strip() and not bullet.match(lines[j]) and not lines[j].startswith(("#", "|", "```")) and lines[j].startswith(" "):
                body.append(lines[j]); j += 1
            out.append(wrap_block(" ".join(body), width, ind, " " * len(ind)))
            i = j; continue
        j = i
        para = []
        while j < n and lines[j].strip() and not bullet.match(lines[j]) and not lines[j].startswith(("#", "|", "```")):
            para.append(lines[j]); j += 1
        out.append(wrap_block(" ".join(para), width))
        i = j
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140)
