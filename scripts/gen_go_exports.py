"""Writes tests/golden/go_reference_exports.json: the exported identifiers of the arrow-go packages that go/arrowhip imports —
package-level names, and per type its exported fields, methods, embedded types and (for interfaces) method set — read from the Go
sources under /root/reference with the tokenizer of tests/go_static.py.  tests/test_go_shim_static.py checks every `pkg.Name` the shim
writes against it (the image has no Go toolchain; /root/reference is not on the GPU box, so the list is a committed fixture).
    python scripts/gen_go_exports.py            # rewrite the fixture
    python scripts/gen_go_exports.py --check    # exit 1 if the committed fixture differs from what the reference gives now
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import go_static as G  # noqa: E402

REF = "/root/reference"
MODULE = "github.com/apache/arrow-go/v18/"
PACKAGES = ["arrow", "arrow/compute", "arrow/compute/exec", "arrow/scalar", "arrow/bitutil", "arrow/array", "arrow/memory"]
OUT = os.path.join(ROOT, "tests", "golden", "go_reference_exports.json")


def exported(name):
    return bool(name) and name[0].isupper()


def interfaces_of(src):
    """{name: (method names, embedded interfaces)} of every `type Name interface { … }` of a source text"""
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?m)^(?:type\s+)?\t?([A-Za-z_][A-Za-z_0-9]*)(?:\[[^\]]*\])?\s+interface\s*\{", src):
        i, depth = m.end(), 1
        stmts, cur = [], ""
        while i < len(src) and depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if depth == 1 and ch in ";\n":
                stmts.append(cur.strip()); cur = ""
            else:
                cur += ch
            i += 1
        stmts.append(cur.strip())
        names, embeds = [], []
        for st in filter(None, stmts):
            mm = re.match(r"([A-Za-z_][A-Za-z_0-9]*)\s*\(", st)
            if mm:
                names.append(mm.group(1))
            elif re.fullmatch(r"[A-Za-z_][A-Za-z_0-9.]*", st):
                embeds.append(st.split(".")[-1])
        out[m.group(1)] = (sorted(set(n for n in names if exported(n))), sorted(set(e for e in embeds if exported(e))))
    return out


def struct_embeds_of(src):
    """{struct name: [embedded type names]} — a line of a struct body that is a type expression alone (`array`, `*memory.Buffer`,
    `floatArray[float64]`)"""
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?m)^(?:type\s+)?\t?([A-Za-z_][A-Za-z_0-9]*)(?:\[[^\]]*\])?\s+struct\s*\{", src):
        i, depth, body = m.end(), 1, ""
        while i < len(src) and depth:
            ch = src[i]
            depth += ch == "{"
            depth -= ch == "}"
            if depth:
                body += ch
            i += 1
        emb = []
        for line in body.split("\n"):
            line = re.sub(r"`[^`]*`", "", line).strip()
            mm = re.fullmatch(r"\*?([A-Za-z_][A-Za-z_0-9.]*)(\[[^\]]*\])?", line)
            if mm:
                emb.append(mm.group(1).split(".")[-1])
        out[m.group(1)] = emb
    return out


def scan(rel):
    d = os.path.join(REF, rel)
    pkg = G.Package()
    ifaces, sembeds, aliases = {}, {}, {}
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".go") and not fn.endswith("_test.go"):
            src = open(os.path.join(d, fn)).read()
            G._scan_file(pkg, fn, G.tokenize(src))
            ifaces.update(interfaces_of(src))
            for am in re.finditer(r"(?m)^(?:type\s+|\t)([A-Za-z_][A-Za-z_0-9]*)\s*=\s*([A-Za-z_][A-Za-z_0-9.]*)\s*$", re.sub(r"//[^\n]*", "", src)):
                aliases[am.group(1)] = am.group(2)
            sembeds.update(struct_embeds_of(src))
    symbols = sorted({q for kind, q, _, _ in pkg.decls if "." not in q and exported(q)})
    types = {}
    for name, fields in pkg.structs.items():   # unexported types too: exported ones embed them (array.Float64 embeds array.array)
        if True:
            types[name] = {"fields": {f: (ty or "").replace(" ", "") for f, ty in sorted(fields.items()) if exported(f)}, "methods": [],
                           "embeds": sorted(set(sembeds.get(name, [])))}
    for name, text in pkg.named_types.items():
        im = ifaces.get(name)
        types[name] = {"fields": {}, "methods": im[0] if im else [], "embeds": im[1] if im else sorted(set(sembeds.get(name, [])))}   # (generic structs land here)
        if re.fullmatch(r"[A-Za-z_][A-Za-z_0-9.]*", text.strip()):   # `type ExecResult = ArraySpan`, `type Type int`
            types[name]["underlying"] = text.strip()
            if aliases.get(name) == text.strip():
                types[name]["alias"] = True     # `type X = Y`: X and Y are the same type
    for f in pkg.funcs:
        if f.recv_type and exported(f.name):
            base = f.recv_type.split("[")[0]
            if base in types:
                types[base]["methods"] = sorted(set(types[base]["methods"]) | {f.name})
    # call shapes: package-level functions and methods → [parameter count, variadic] (a name defined more than once — per-platform
    # files — keeps a shape only if all definitions agree)
    funcs, clash = {}, set()
    for f in pkg.funcs:
        if not exported(f.name):
            continue
        key = f.name if not f.recv_type else f.recv_type.split("[")[0] + "." + f.name
        shape = [f.nparams, f.variadic]
        if key in funcs and funcs[key] != shape:
            clash.add(key)
        funcs[key] = shape
    for k in clash:
        del funcs[k]
    return {"symbols": symbols, "types": types, "funcs": dict(sorted(funcs.items()))}


def main():
    data = {MODULE + rel: scan(rel) for rel in PACKAGES}
    text = json.dumps(data, indent=0, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(OUT, {k.rsplit("/", 1)[-1]: len(v["symbols"]) for k, v in data.items()})


if __name__ == "__main__":
    main()
