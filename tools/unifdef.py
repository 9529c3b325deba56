#!/usr/bin/env python3
"""A small unifdef: resolves the preprocessor conditionals of a source file whose macros are given fixed values and
drops the branches that can no longer be compiled, together with the `#ifndef X / #define X v / #endif` blocks that
defined them.  Used once per round to retire the experiment switches DESIGN.md lists as settled.

    tools/unifdef.py file.hip RTX_BURN=0 RTX_PRUNE=1 ... > out.hip

Conditions that mention any macro that is NOT given stay as they are (with the known ones substituted when --subst).
Uses of a retired macro in ordinary code are reported on stderr (they are edited by hand)."""
import re
import sys


def main():
    path = sys.argv[1]
    known = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        known[k] = int(v)
    ident = re.compile(r"[A-Za-z_][A-Za-z_0-9]*")

    def evaluate(expr):
        """value of the condition, or None when it mentions something unknown"""
        expr = re.sub(r"//.*", "", expr).strip()
        names = set(ident.findall(expr)) - {"defined"}
        if not names or not names <= set(known):
            return None
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", "1", expr)
        e = ident.sub(lambda m: str(known[m.group(0)]), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", " !=")
        return bool(eval(e))

    lines = open(path).read().split("\n")
    out = []
    # stack entries: [kind, taken_already, emitting, resolved]  (resolved: the directive lines themselves are dropped)
    stack = []
    i = 0
    emitting = lambda: all(s[2] for s in stack)
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", ln)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("ifndef", "ifdef"):
            name = ident.search(rest).group(0)
            if name in known and d == "ifndef":
                # the defining block of a retired macro: dropped whole
                depth = 1
                while depth:
                    i += 1
                    mm = re.match(r"\s*#\s*(ifndef|ifdef|if|endif)\b", lines[i])
                    if mm:
                        depth += -1 if mm.group(1) == "endif" else 1
                i += 1
                continue
            stack.append([d, False, True, False])
            if emitting():
                out.append(ln)
        elif d == "if":
            v = evaluate(rest)
            if v is None:
                stack.append([d, False, True, False])
                if emitting():
                    out.append(ln)
            else:
                stack.append([d, v, v, True])
        elif d == "elif":
            s = stack[-1]
            if s[3]:
                v = evaluate(rest)
                if v is None:
                    raise SystemExit("%s:%d: #elif with unknown macros after a resolved #if" % (path, i + 1))
                s[2] = (not s[1]) and v
                s[1] = s[1] or v
            elif all(t[2] for t in stack[:-1]):
                out.append(ln)
        elif d == "else":
            s = stack[-1]
            if s[3]:
                s[2] = not s[1]
                s[1] = True
            elif all(t[2] for t in stack[:-1]):
                out.append(ln)
        else:
            s = stack.pop()
            if not s[3] and emitting():
                out.append(ln)
        i += 1
    text = "\n".join(out)
    for n, ln in enumerate(out):
        code = re.sub(r"//.*", "", ln)
        for k in known:
            if re.search(r"\b%s\b" % k, code):
                sys.stderr.write("use of %s at output line %d: %s\n" % (k, n + 1, ln.strip()[:140]))
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
