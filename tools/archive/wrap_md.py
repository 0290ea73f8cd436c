"""Wrap the long PROSE lines of a Markdown file (paragraphs and list items) at a given width; tables, headings, fenced and indented code are left
alone.  Continuation lines keep the item's indentation, and never start with something Markdown would read as a marker.
    python tools/wrap_md.py DESIGN.md [width]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
marker = re.compile(r"^(\s*)([*+-] |\d+\. )?")
bad_start = re.compile(r"^([*+-] |\d+\. |#|\||>|=== |--- )")
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence
        out.append(line); continue
    if fence or len(line) <= width + 2 or line.lstrip().startswith(("|", "#")) or line.startswith("    "):
        out.append(line); continue
    m = marker.match(line)
    indent, bullet = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(bullet):]
    cont = indent + " " * len(bullet)
    parts = textwrap.wrap(body, width=width - len(cont), break_long_words=False, break_on_hyphens=False)
    # a continuation line must not look like a list item / heading / table row
    k = 1
    while k < len(parts):
        if bad_start.match(parts[k]) and " " in parts[k - 1]:
            head, last = parts[k - 1].rsplit(" ", 1)
            parts[k - 1], parts[k] = head, last + " " + parts[k]
        k += 1
    out.append(indent + bullet + parts[0])
    out.extend(cont + p for p in parts[1:])
open(path, "w").write("\n".join(out))
