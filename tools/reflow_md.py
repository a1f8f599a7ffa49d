"""Re-wraps a markdown file to at most 118 columns: paragraphs and list items are filled, fenced code is left alone, a table
with a line over 120 columns becomes a nested list (row = item, other cells = sub-items named by their column header).  What keeps
DESIGN.md and docs/design/*.md readable (tests/test_bench_host.py checks the width).   usage: python tools/reflow_md.py FILE > OUT"""
import re, sys, textwrap
W = 118

def wrap(text, indent="", first=None):
    first = indent if first is None else first
    return textwrap.fill(" ".join(text.split()), width=W, initial_indent=first, subsequent_indent=indent,
                         break_long_words=False, break_on_hyphens=False)

def cells(line):
    line = line.strip()
    if line.startswith("|"): line = line[1:]
    if line.endswith("|"): line = line[:-1]
    # split on unescaped pipes
    out, cur, i = [], "", 0
    while i < len(line):
        if line[i] == "\\" and i + 1 < len(line) and line[i+1] == "|":
            cur += "|"; i += 2; continue
        if line[i] == "|":
            out.append(cur.strip()); cur = ""
        else:
            cur += line[i]
        i += 1
    out.append(cur.strip())
    return out

def table(lines):
    if max(len(l) for l in lines) <= 120:
        return lines
    head = cells(lines[0]); rows = [cells(l) for l in lines[2:]]
    out = []
    for r in rows:
        out.append(wrap(f"**{r[0]}**", indent="  ", first="* "))
        for h, c in zip(head[1:], r[1:]):
            if c:
                out.append(wrap(f"*{h}:* {c}" if h else c, indent="    ", first="  - "))
    return out

LIST = re.compile(r"^(\s*)([*\-+]|\d+\.|\([a-z0-9]+\))\s+")

def reflow(src):
    lines = src.split("\n")
    out, i, n = [], 0, len(lines)
    while i < n:
        l = lines[i]
        if l.startswith("```"):
            out.append(l); i += 1
            while i < n and not lines[i].startswith("```"):
                out.append(lines[i]); i += 1
            if i < n: out.append(lines[i]); i += 1
            continue
        if l.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|?[\s:|-]+\|[\s:|-]*$", lines[i+1]):
            j = i
            while j < n and lines[j].lstrip().startswith("|"): j += 1
            out += table(lines[i:j]); i = j; continue
        if not l.strip() or l.startswith("#") or l.startswith("{"):
            out.append(l if len(l) <= 120 or not l.startswith("#") else l); i += 1; continue
        # paragraph or list item: gather continuation lines
        m = LIST.match(l)
        if m:
            indent = " " * (len(m.group(1)) + len(m.group(2)) + 1)
            first = m.group(1) + m.group(2) + " "
            body = l[m.end():]
        else:
            lead = len(l) - len(l.lstrip())
            indent = first = " " * lead
            body = l.strip()
        j = i + 1
        while j < n and lines[j].strip() and not lines[j].startswith("#") and not lines[j].startswith("```") \
                and not LIST.match(lines[j]) and not lines[j].lstrip().startswith("|"):
            body += " " + lines[j].strip(); j += 1
        out.append(wrap(body, indent=indent, first=first)); i = j
    return "\n".join(out)

if __name__ == "__main__":
    sys.stdout.write(reflow(open(sys.argv[1]).read()))
