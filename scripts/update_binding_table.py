#!/usr/bin/env python3
"""Rewrites the generated Go binding table of INTEGRATION.md (between the go-binding-table markers) from go/arrowhip/*.go and
include/arrowhip.h — tests/test_go_shim_static.py fails when it is stale."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import go_static as G  # noqa: E402

chk = G.check()
if chk.errors:
    sys.exit("\n".join(chk.errors))
path = os.path.join(ROOT, "INTEGRATION.md")
text = open(path).read()
block = "<!-- go-binding-table:begin -->\n" + G.binding_table(chk) + "<!-- go-binding-table:end -->"
if "<!-- go-binding-table:begin -->" in text:
    text = re.sub(r"<!-- go-binding-table:begin -->\n.*?<!-- go-binding-table:end -->", lambda _m: block, text, flags=re.S)
else:
    text = text.rstrip("\n") + "\n\n" + block + "\n"
open(path, "w").write(text)
print(f"{len(chk.bound())} of {len(chk.protos)} entry points bound; table written")
