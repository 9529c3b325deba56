#!/usr/bin/env python3
"""Copies the binding between the two BINDING markers of oracle/ref_binding.cpp -- the code that is compiled against the real
reference and tested (tests/test_ref_binding.py) -- into INTEGRATION.md, between its <!-- BINDING:BEGIN/END --> markers."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "oracle", "ref_binding.cpp")).read()
a = src.index("// ---- BINDING (begin)")
a = src.index("\n", a) + 1
b = src.index("// ---- BINDING (end)")
body = src[a:b].rstrip() + "\n"
p = os.path.join(ROOT, "INTEGRATION.md")
md = open(p).read()
i = md.index("<!-- BINDING:BEGIN -->") + len("<!-- BINDING:BEGIN -->")
j = md.index("<!-- BINDING:END -->")
md = md[:i] + "\n```cpp\n" + body + "```\n" + md[j:]
open(p, "w").write(md)
print("INTEGRATION.md: %d lines of binding" % body.count("\n"))
