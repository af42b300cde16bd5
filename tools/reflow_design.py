"""One-off formatter for DESIGN.md (VERDICT r4 hygiene: "94 KB of 1-2 KB lines"): the two wide tables (kernel families, parity pins) become headed
paragraphs, every paragraph / list item is wrapped at 150 columns; code blocks and the small tables stay as they are.
    python tools/reflow_design.py DESIGN.md"""
import re, sys, textwrap

W = 150
path = sys.argv[1]
lines = open(path).read().split("\n")
out, i, in_code = [], 0, False


def wrap(text, first="", rest=""):
    return textwrap.wrap(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    parts, cur, depth = [], "", 0
    body = row.strip()
    body = body[1:] if body.startswith("|") else body
    body = body[:-1] if body.endswith("|") else body
    tick = False
    for ch in body:
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick:
            parts.append(cur.strip()); cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


while i < len(lines):
    l = lines[i]
    if l.strip().startswith("```"):
        in_code = not in_code
        out.append(l); i += 1; continue
    if in_code:
        out.append(l); i += 1; continue
    if l.lstrip().startswith("|"):
        j = i
        while j < len(lines) and lines[j].lstrip().startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) <= 260:
            out.extend(block)
        else:
            hdr = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                out.append("")
                out.extend(wrap("**" + c[0].replace("**", "") + "**"))
                out.append("")
                for name, val in zip(hdr[1:], c[1:]):
                    if val and val != "—":
                        out.extend(wrap(f"*{name}:* {val}", "- ", "  "))
            out.append("")
        i = j; continue
    if not l.strip() or l.startswith("#"):
        out.append(l); i += 1; continue
    m = re.match(r"^(\s*)([*\-] |\d+\. )?(.*)$", l)
    indent, bullet, text = m.group(1), m.group(2) or "", m.group(3)
    # the paragraph / list item continues over the following lines until a blank line, a heading, a table row, a code fence or the next list item
    j = i + 1
    while j < len(lines):
        n = lines[j]
        if not n.strip() or n.startswith("#") or n.lstrip().startswith("|") or n.strip().startswith("```") or re.match(r"^\s*([*\-] |\d+\. )", n):
            break
        text += " " + n.strip()
        j += 1
    out.extend(wrap(text, indent + bullet, indent + " " * len(bullet)))
    i = j
open(path, "w").write("\n".join(out))
print(len(out), "lines, longest", max(len(x) for x in out))
