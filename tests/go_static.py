"""A static checker for go/arrowhip/*.go that needs no Go toolchain (TEST INFRASTRUCTURE; the image has none, and the package had
shipped with a type declared twice).  It is not a Go compiler.  It checks what a binding over a C header gets wrong in practice:

  * lexical sanity: every (), [] and {} closes, per file;
  * one declaration per name in the package: top-level types, funcs, vars, consts, and methods per receiver type;
  * every import is used, every file carries the package clause and the build tag;
  * every `C.ah_*(...)` call against include/arrowhip.h: the function exists, the argument count matches, and every argument whose
    Go type the checker can derive (C.T(...) conversions, (*C.T)(p) casts, unsafe.Pointer, &local, parameters, locals declared
    with `var` / `:=`, struct fields, results of package functions) matches the C parameter the way cgo maps it
    (void* ↔ unsafe.Pointer, T* ↔ *C.T, scalar T ↔ C.T, untyped constants and nil as Go allows);
  * every `C.AH_*` name is an enumerator or macro of the header;
  * every method or field selected on a value whose type is a struct of the package exists;
  * the bound / unbound table of the header's entry points.

Arguments whose type cannot be derived are reported as `unverified` (with the expression), never silently passed."""
from __future__ import annotations

import json
import os
import re
import subprocess
from dataclasses import dataclass, field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO_DIR = os.path.join(ROOT, "go", "arrowhip")
HEADER = os.path.join(ROOT, "include", "arrowhip.h")
GO_EXPORTS = os.path.join(ROOT, "tests", "golden", "go_reference_exports.json")

# ---------------------------------------------------------------------------------------------------------------- C header


@dataclass
class CProto:
    name: str
    ret: str
    params: list      # canonical C types: "int", "int64_t", "void*", "uint8_t*", "ah_ctx**", ...


def _canon_ctype(decl: str) -> str:
    """'const uint8_t* lvalid' → 'uint8_t*';  'const void* const* values' → 'void**';  'int64_t n' → 'int64_t'"""
    d = decl.strip()
    d = re.sub(r"\bconst\b", " ", d)
    stars = d.count("*")
    d = d.replace("*", " ")
    words = d.split()
    if len(words) >= 2 and words[0] in ("unsigned", "signed", "long", "struct"):
        base = " ".join(words[:-1]) if len(words) > 2 else " ".join(words)
    else:
        base = words[0]
    return base + "*" * stars


def parse_header(path: str = HEADER):
    """→ ({name: CProto}, {enumerators and object-like macros})"""
    text = subprocess.check_output(["gcc", "-E", "-P", "-dD", "-x", "c", path], text=True)
    macros = set(re.findall(r"^#define\s+(AH_[A-Z0-9_]+)\b", text, flags=re.M))
    body = re.sub(r"^#.*$", "", text, flags=re.M)
    consts = set(macros)
    for m in re.finditer(r"\benum\b[^{;]*\{([^}]*)\}", body):
        for item in m.group(1).split(","):
            nm = item.split("=")[0].strip()
            if nm:
                consts.add(nm)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ah_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "(" in ret:
            continue
        plist = [] if params in ("", "void") else [_canon_ctype(p) for p in params.split(",")]
        protos[name] = CProto(name, _canon_ctype(ret + " x"), plist)
    return protos, consts


# ---------------------------------------------------------------------------------------------------------------- Go lexer


@dataclass
class Tok:
    kind: str   # ident number string op nl
    text: str
    line: int


_OPS3 = ("<<=", ">>=", "&^=", "...")
_OPS2 = (":=", "==", "!=", "<=", ">=", "&&", "||", "<-", "++", "--", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<", ">>", "&^")


def tokenize(src: str):
    toks, i, line, n = [], 0, 1, len(src)
    while i < n:
        c = src[i]
        if c == "\n":
            toks.append(Tok("nl", "\n", line)); line += 1; i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            if j < 0:
                raise SyntaxError(f"line {line}: unterminated block comment")
            line += src.count("\n", i, j); i = j + 2
        elif c == "`":
            j = src.find("`", i + 1)
            if j < 0:
                raise SyntaxError(f"line {line}: unterminated raw string")
            toks.append(Tok("string", src[i:j + 1], line)); line += src.count("\n", i, j); i = j + 1
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                if src[j] == "\n":
                    raise SyntaxError(f"line {line}: newline in string literal")
                j += 2 if src[j] == "\\" else 1
            toks.append(Tok("string", src[i:j + 1], line)); i = j + 1
        elif c.isalpha() or c == "_":
            j = i
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            toks.append(Tok("ident", src[i:j], line)); i = j
        elif c.isdigit():
            j = i
            while j < n and (src[j].isalnum() or src[j] in "._"):
                j += 1
            toks.append(Tok("number", src[i:j], line)); i = j
        else:
            for ops in (_OPS3, _OPS2):
                hit = next((o for o in ops if src.startswith(o, i)), None)
                if hit:
                    break
            t = hit or c
            toks.append(Tok("op", t, line)); i += len(t)
    return toks


def check_balanced(toks, fname):
    errs, stack = [], []
    pair = {")": "(", "]": "[", "}": "{"}
    for t in toks:
        if t.kind != "op":
            continue
        if t.text in "([{":
            stack.append(t)
        elif t.text in ")]}":
            if not stack or stack[-1].text != pair[t.text]:
                errs.append(f"{fname}:{t.line}: unbalanced '{t.text}'")
                return errs
            stack.pop()
    for t in stack:
        errs.append(f"{fname}:{t.line}: '{t.text}' never closed")
    return errs


# ---------------------------------------------------------------------------------------------------------------- Go structure


@dataclass
class Func:
    name: str
    recv_name: str | None
    recv_type: str | None      # without '*'
    params: dict
    results: list              # result types, in order
    named_results: dict
    body: list
    file: str
    line: int
    nparams: int = 0           # parameters as written (unnamed ones included)
    variadic: bool = False


@dataclass
class Package:
    files: dict = field(default_factory=dict)        # fname → tokens (no nl)
    decls: list = field(default_factory=list)        # (kind, qualified name, file, line)
    structs: dict = field(default_factory=dict)      # type → {field: type}
    named_types: dict = field(default_factory=dict)  # type → underlying type string (non-struct)
    funcs: list = field(default_factory=list)
    vars: dict = field(default_factory=dict)         # package-level var/const → type or None
    imports: dict = field(default_factory=dict)      # fname → {local name: path}
    errors: list = field(default_factory=list)


def _match(toks, i):
    """index of the token closing the bracket opened at toks[i]"""
    open_, close = toks[i].text, {"(": ")", "[": "]", "{": "}"}[toks[i].text]
    depth = 0
    for j in range(i, len(toks)):
        if toks[j].kind == "op":
            if toks[j].text == open_:
                depth += 1
            elif toks[j].text == close:
                depth -= 1
                if depth == 0:
                    return j
    raise SyntaxError(f"line {toks[i].line}: no closing {close}")


def _split_commas(toks):
    out, cur, depth = [], [], 0
    for t in toks:
        if t.kind == "op" and t.text in "([{":
            depth += 1
        elif t.kind == "op" and t.text in ")]}":
            depth -= 1
        if t.kind == "op" and t.text == "," and depth == 0:
            out.append(cur); cur = []
        else:
            cur.append(t)
    if cur:
        out.append(cur)
    return out


def _join(toks):
    s = ""
    for t in toks:
        if t.kind == "ident" and s and (s[-1].isalnum() or s[-1] == "_"):
            s += " "
        s += t.text
    return s


def _parse_param_list(toks):
    """Go parameter / result list → ordered [(name or None, type string)]"""
    groups = _split_commas(toks)
    if not groups:
        return []
    # a group is either "name Type", "name" (type follows in a later group) or "Type" (unnamed list)
    def is_named(g):
        return len(g) >= 2 and g[0].kind == "ident" and not (g[1].kind == "op" and g[1].text == ".")
    if not any(is_named(g) for g in groups):
        return [(None, _join(g)) for g in groups]
    out, pending = [], []
    for g in groups:
        if is_named(g):
            typ = _join(g[1:])
            for p in pending:
                out.append((p, typ))
            pending = []
            out.append((g[0].text, typ))
        else:
            pending.append(g[0].text)
    return out


def load_package(go_dir: str = GO_DIR) -> Package:
    pkg = Package()
    for fname in sorted(os.listdir(go_dir)):
        if not fname.endswith(".go"):
            continue
        src = open(os.path.join(go_dir, fname)).read()
        try:
            raw = tokenize(src)
        except SyntaxError as e:
            pkg.errors.append(f"{fname}: {e}")
            continue
        pkg.errors += check_balanced(raw, fname)
        if not re.search(r"^//go:build hip\s*$", src, flags=re.M):
            pkg.errors.append(f"{fname}: no '//go:build hip' line")
        if not re.search(r"^package arrowhip\s*$", src, flags=re.M):
            pkg.errors.append(f"{fname}: no 'package arrowhip' clause")
        if pkg.errors and any(e.startswith(fname) and ("unbalanced" in e or "never closed" in e) for e in pkg.errors):
            continue
        _scan_file(pkg, fname, raw)
    _duplicates(pkg)
    return pkg


def _scan_file(pkg: Package, fname: str, raw):
    toks = [t for t in raw if t.kind != "nl"]
    pkg.files[fname] = toks
    imports = {}
    i, n = 0, len(raw)
    depth = 0
    # top level scan over the raw stream (newlines matter inside grouped declarations)
    while i < n:
        t = raw[i]
        if t.kind == "op" and t.text in "([{":
            i = _match(raw, i) + 1
            continue
        if t.kind != "ident":
            i += 1
            continue
        if t.text == "import":
            j = i + 1
            while raw[j].kind == "nl":
                j += 1
            specs = []
            if raw[j].kind == "op" and raw[j].text == "(":
                k = _match(raw, j)
                line_toks = []
                for u in raw[j + 1:k + 1]:
                    if u.kind == "nl" or u is raw[k]:
                        if line_toks:
                            specs.append(line_toks)
                        line_toks = []
                    else:
                        line_toks.append(u)
                i = k + 1
            else:
                k = j
                while raw[k].kind != "nl":
                    k += 1
                specs.append(raw[j:k]); i = k
            for sp in specs:
                path = sp[-1].text.strip('"`')
                local = sp[0].text if len(sp) == 2 else path.rsplit("/", 1)[-1]
                if re.fullmatch(r"v\d+", local):
                    local = path.rsplit("/", 2)[-2]
                imports[local] = (path, sp[-1].line)
            continue
        if t.text == "func":
            i = _scan_func(pkg, fname, raw, i)
            continue
        if t.text == "type":
            i = _scan_type(pkg, fname, raw, i)
            continue
        if t.text in ("var", "const"):
            i = _scan_var(pkg, fname, raw, i, t.text)
            continue
        i += 1
    pkg.imports[fname] = imports
    used = {toks[k].text for k in range(len(toks) - 1) if toks[k].kind == "ident" and toks[k + 1].kind == "op" and toks[k + 1].text == "."}
    for local, (path, line) in imports.items():
        if local not in used and local != "_":
            pkg.errors.append(f"{fname}:{line}: \"{path}\" imported and not used")
    del depth


def _skip_nl(raw, i):
    while i < len(raw) and raw[i].kind == "nl":
        i += 1
    return i


def _scan_func(pkg, fname, raw, i):
    line = raw[i].line
    j = i + 1
    recv_name = recv_type = None
    if raw[j].kind == "op" and raw[j].text == "(":
        k = _match(raw, j)
        inner = [t for t in raw[j + 1:k] if t.kind != "nl"]
        idents = [t.text for t in inner if t.kind == "ident"]
        recv_type = idents[-1]
        recv_name = idents[0] if len(idents) > 1 else None
        j = k + 1
    name = raw[j].text
    j += 1
    if raw[j].kind == "op" and raw[j].text == "[":           # type parameters
        j = _match(raw, j) + 1
    k = _match(raw, j)
    params = _parse_param_list([t for t in raw[j + 1:k] if t.kind != "nl"])
    j = k + 1
    results, named = [], {}
    if raw[j].kind == "op" and raw[j].text == "(":
        k = _match(raw, j)
        for nm, ty in _parse_param_list([t for t in raw[j + 1:k] if t.kind != "nl"]):
            results.append(ty)
            if nm:
                named[nm] = ty
        j = k + 1
    else:
        rt = []
        while not (raw[j].kind == "op" and raw[j].text == "{") and raw[j].kind != "nl":
            rt.append(raw[j]); j += 1
        if rt:
            results.append(_join(rt))
    body = []
    if raw[j].kind == "op" and raw[j].text == "{":
        k = _match(raw, j)
        body = raw[j + 1:k]
        j = k + 1
    pkg.funcs.append(Func(name, recv_name, recv_type, {nm: ty for nm, ty in params if nm}, results, named, body, fname, line,
                          len(params), bool(params) and (params[-1][1] or "").startswith("...")))
    pkg.decls.append(("method" if recv_type else "func", f"{recv_type}.{name}" if recv_type else name, fname, line))
    return j


def _scan_type(pkg, fname, raw, i):
    j = _skip_nl(raw, i + 1)
    specs = []
    if raw[j].kind == "op" and raw[j].text == "(":
        k = _match(raw, j)
        u = j + 1
        while u < k:
            u = _skip_nl(raw, u)
            if u >= k:
                break
            v = u
            while v < k and raw[v].kind != "nl":
                if raw[v].kind == "op" and raw[v].text in "([{":
                    v = _match(raw, v)
                v += 1
            specs.append(raw[u:v]); u = v
        end = k + 1
    else:
        v = j
        while v < len(raw) and raw[v].kind != "nl":
            if raw[v].kind == "op" and raw[v].text in "([{":
                v = _match(raw, v)
            v += 1
        specs.append(raw[j:v]); end = v
    for sp in specs:
        name = sp[0].text
        pkg.decls.append(("type", name, fname, sp[0].line))
        rest = [t for t in sp[1:] if not (t.kind == "op" and t.text == "=")]
        if rest and rest[0].kind == "ident" and rest[0].text == "struct":
            b = next(x for x, t in enumerate(rest) if t.kind == "op" and t.text == "{")
            e = _match(rest, b)
            fields, line_toks = {}, []
            for t in rest[b + 1:e + 1]:
                if t.kind == "nl" or t is rest[e]:
                    lt = [x for x in line_toks if x.kind != "string"]
                    if lt:
                        for nm, ty in _parse_param_list(lt):
                            if nm:
                                fields[nm] = ty
                            else:                        # embedded type
                                fields[ty.lstrip("*").split(".")[-1]] = ty
                    line_toks = []
                else:
                    line_toks.append(t)
            pkg.structs[name] = fields
        else:
            pkg.named_types[name] = _join([t for t in rest if t.kind != "nl"])
    return end


def _scan_var(pkg, fname, raw, i, kw):
    j = _skip_nl(raw, i + 1)
    lines = []
    if raw[j].kind == "op" and raw[j].text == "(":
        k = _match(raw, j)
        u = j + 1
        while u < k:
            u = _skip_nl(raw, u)
            if u >= k:
                break
            v = u
            while v < k and raw[v].kind != "nl":
                if raw[v].kind == "op" and raw[v].text in "([{":
                    v = _match(raw, v)
                v += 1
            lines.append(raw[u:v]); u = v
        end = k + 1
    else:
        v = j
        while v < len(raw) and raw[v].kind != "nl":
            if raw[v].kind == "op" and raw[v].text in "([{":
                v = _match(raw, v)
            v += 1
        lines.append(raw[j:v]); end = v
    for ln in lines:
        eq = next((x for x, t in enumerate(ln) if t.kind == "op" and t.text == "="), len(ln))
        lhs = _split_commas(ln[:eq])
        typ = None
        if lhs and len(lhs[-1]) > 1:
            typ = _join(lhs[-1][1:])
        for g in lhs:
            if g and g[0].kind == "ident":
                pkg.decls.append((kw, g[0].text, fname, g[0].line))
                pkg.vars[g[0].text] = typ
    return end


def _duplicates(pkg):
    seen = {}
    for kind, name, fname, line in pkg.decls:
        key = name if kind == "method" else ("·" + name)     # funcs, types, vars and consts share the package scope
        if name in ("init", "_") and kind == "func":
            continue
        if key in seen:
            pf, pl = seen[key]
            pkg.errors.append(f"{fname}:{line}: {name} redeclared in this block (other declaration at {pf}:{pl})")
        else:
            seen[key] = (fname, line)
    # a method and a field of the same name on one struct do not compile either
    for kind, name, fname, line in pkg.decls:
        if kind == "method":
            ty, m = name.split(".")
            if m in pkg.structs.get(ty, {}):
                pkg.errors.append(f"{fname}:{line}: type {ty} has both field and method named {m}")


# ---------------------------------------------------------------------------------------------------------------- cgo calls

_SCALARS = {"int", "int8_t", "int16_t", "int32_t", "int64_t", "uint8_t", "uint16_t", "uint32_t", "uint64_t", "size_t", "double", "float", "char"}


def go_type_to_c(ty: str | None):
    """Go type string → canonical C type as cgo sees it, or None if it has no C meaning the checker knows"""
    if ty is None:
        return None
    ty = ty.replace(" ", "")
    stars = len(ty) - len(ty.lstrip("*"))
    base = ty.lstrip("*")
    if base == "unsafe.Pointer":
        return "void*" + "*" * stars
    if base.startswith("C."):
        return base[2:] + "*" * stars
    return None


@dataclass
class CallReport:
    file: str
    line: int
    func: str
    cname: str
    nargs: int
    verified: int
    unverified: list


class Checker:
    def __init__(self, pkg: Package, protos, consts):
        self.pkg, self.protos, self.consts = pkg, protos, consts
        self.errors = list(pkg.errors)
        self.ext = json.load(open(GO_EXPORTS)) if os.path.exists(GO_EXPORTS) else {}   # exported names of the imported arrow-go packages
        self.ext_checked = 0      # pkg.Name uses and members of external types that were looked up
        self.local_calls_checked = 0
        self.local_args_typed = 0
        self.returns_checked = 0
        self.calls = []
        self.methods = {}
        self.func_results = {}
        for f in pkg.funcs:
            if f.recv_type:
                self.methods.setdefault(f.recv_type, {})[f.name] = f
            else:
                self.func_results[f.name] = f.results

    # -- environments
    def _env(self, f: Func):
        self._cur = f
        env = dict(self.pkg.vars)
        env.update(f.params)
        env.update(f.named_results)
        if f.recv_name:
            env[f.recv_name] = "*" + f.recv_type
        body = [t for t in f.body if t.kind != "nl"]
        self._locals(body, env)
        return env, body

    def _locals(self, body, env):
        n = len(body)
        for i, t in enumerate(body):
            if t.kind == "ident" and t.text == "func" and i + 1 < n and body[i + 1].kind == "op" and body[i + 1].text == "(":
                # a function literal: its parameters are names of the enclosing body too (flat scoping: the ExecFn closures of
                # register.go — `return func(ctx *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error { … }`)
                close = _match(body, i + 1)
                for nm, ty in _parse_param_list(body[i + 2:close]):
                    if nm and nm != "_" and ty:
                        env.setdefault(nm, ty)
                continue
            if t.kind == "ident" and t.text == "var" and i + 1 < n and body[i + 1].kind == "ident":
                j = i + 1
                names = []
                while body[j].kind == "ident" and body[j + 1].kind == "op" and body[j + 1].text == ",":
                    names.append(body[j].text); j += 2
                names.append(body[j].text); j += 1
                ty = []
                depth = 0
                while j < n:
                    u = body[j]
                    if u.kind == "op" and u.text in "([":
                        depth += 1
                    elif u.kind == "op" and u.text in ")]":
                        depth -= 1
                    if depth == 0 and ((u.kind == "op" and u.text in ("=", "}", ";")) or (u.line != body[j - 1].line)):
                        break
                    ty.append(u); j += 1
                for nm in names:
                    env[nm] = _join(ty) if ty else None
            elif t.kind == "op" and t.text == ":=":
                # names on the left (same line, back to the statement start)
                k = i - 1
                lhs = []
                while k >= 0 and body[k].line == t.line and (body[k].kind == "ident" or (body[k].kind == "op" and body[k].text == ",")):
                    if body[k].kind == "ident" and body[k].text in self._GO_KEYWORDS:     # `if err := …`, `for i, v := range …`
                        break
                    if body[k].kind == "ident":
                        lhs.insert(0, body[k].text)
                    k -= 1
                # right-hand side up to the end of the statement
                j, depth, rhs = i + 1, 0, []
                while j < n:
                    u = body[j]
                    if u.kind == "op" and u.text in "([{":
                        depth += 1
                    elif u.kind == "op" and u.text in ")]}":
                        if depth == 0:
                            break
                        depth -= 1
                    if depth == 0 and u.line != body[j - 1].line and rhs:
                        break
                    if depth == 0 and u.kind == "op" and u.text in (";", "{"):
                        break
                    rhs.append(u); j += 1
                exprs = _split_commas(rhs)
                if len(exprs) == len(lhs):
                    for nm, ex in zip(lhs, exprs):
                        ty = self._expr_type(ex, env)
                        if ty is not None or nm not in env:
                            env[nm] = ty
                elif len(exprs) == 1:                       # multi-value call
                    tys = self._call_results(exprs[0], env)
                    if tys and len(tys) != len(lhs) and getattr(self, "_cur", None) is not None:
                        self.errors.append(f"{self._cur.file}:{t.line}: assignment mismatch: {len(lhs)} variables but the call returns {len(tys)} values")
                    if tys and len(tys) == len(lhs):
                        for nm, ty in zip(lhs, tys):
                            env[nm] = ty
                    else:
                        for nm in lhs:
                            env.setdefault(nm, None)

    def _strip(self, ty):
        return ty.replace(" ", "").lstrip("*") if ty else ty

    def _resolve_selector(self, toks, env):
        """a.b.c with a in env and b, c struct fields → Go type string or None"""
        if not toks or toks[0].kind != "ident":
            return None
        ty = env.get(toks[0].text)
        i = 1
        while i + 1 < len(toks) + 1 and i < len(toks):
            if not (toks[i].kind == "op" and toks[i].text == "." and i + 1 < len(toks) and toks[i + 1].kind == "ident"):
                return None
            fields = self.pkg.structs.get(self._strip(ty) or "", None)
            if fields is None or toks[i + 1].text not in fields:
                return None
            ty = fields[toks[i + 1].text]
            i += 2
        return ty

    def _call_results(self, ex, env):
        """types returned by a call expression of the package, or None"""
        if not ex or not (ex[-1].kind == "op" and ex[-1].text == ")"):
            return None
        # find the '(' that opens the final call
        depth = 0
        for k in range(len(ex) - 1, -1, -1):
            if ex[k].kind == "op" and ex[k].text == ")":
                depth += 1
            elif ex[k].kind == "op" and ex[k].text == "(":
                depth -= 1
                if depth == 0:
                    break
        callee = ex[:k]
        if len(callee) == 1 and callee[0].kind == "ident":
            return self.func_results.get(callee[0].text)
        if len(callee) >= 3 and callee[-2].kind == "op" and callee[-2].text == ".":
            recv_ty = self._resolve_selector(callee[:-2], env)
            m = self.methods.get(self._strip(recv_ty) or "", {}).get(callee[-1].text)
            if m:
                return m.results
        return None

    def _expr_type(self, ex, env):
        """Go type string of an expression, where derivable"""
        if not ex:
            return None
        t0 = ex[0]
        txt = _join(ex)
        if len(ex) == 1 and t0.kind == "ident":
            if t0.text in ("nil", "true", "false"):
                return "untyped " + t0.text
            return env.get(t0.text)
        if all(t.kind == "number" or (t.kind == "op" and t.text in "-+<>*/|") for t in ex):
            return "untyped const"
        # C.T(expr) / C.NAME
        if t0.kind == "ident" and t0.text == "C" and len(ex) >= 3 and ex[1].text == ".":
            if len(ex) == 3:
                return "untyped const" if ex[2].text.startswith("AH_") else None
            if ex[3].kind == "op" and ex[3].text == "(" and _match(ex, 3) == len(ex) - 1:
                nm = ex[2].text
                if nm == "CString":
                    return "*C.char"
                if nm == "GoString" or nm == "GoBytes":
                    return None
                if nm.startswith("ah_"):
                    p = self.protos.get(nm)
                    if p is None:
                        return None
                    return {"int": "C.int", "char*": "*C.char", "void": None}.get(p.ret)
                return "C." + nm
        # (*C.T)(expr), (**C.T)(expr), (*T)(expr)
        if t0.kind == "op" and t0.text == "(":
            k = _match(ex, 0)
            if k + 1 < len(ex) and ex[k + 1].kind == "op" and ex[k + 1].text == "(" and _match(ex, k + 1) == len(ex) - 1:
                inner = ex[1:k]
                if inner and inner[0].kind == "op" and inner[0].text == "*":
                    return _join(inner)
        if txt.startswith("unsafe.Pointer(") and _match(ex, 3) == len(ex) - 1:
            return "unsafe.Pointer"
        # make([]T, n) / make([]T, n, cap)
        if t0.kind == "ident" and t0.text == "make" and len(ex) > 2 and ex[1].text == "(" and _match(ex, 1) == len(ex) - 1:
            parts = _split_commas(ex[2:-1])
            if parts:
                return _join(parts[0])
        if t0.kind == "op" and t0.text == "&":
            inner = self._expr_type(ex[1:], env)
            if inner and not inner.startswith("untyped"):
                return "*" + inner
            return None
        if t0.kind == "op" and t0.text == "*":
            inner = self._expr_type(ex[1:], env)
            if inner and inner.startswith("*"):
                return inner[1:]
            return None
        # selector chain
        if all((t.kind == "ident") or (t.kind == "op" and t.text == ".") for t in ex):
            return self._resolve_selector(ex, env)
        # index into a known array / slice:  id[0]
        if t0.kind == "ident" and len(ex) >= 4 and ex[1].kind == "op" and ex[1].text == "[" and _match(ex, 1) == len(ex) - 1:
            base = env.get(t0.text)
            if base:
                m = re.match(r"\[[^\]]*\](.*)", base.replace(" ", ""))
                if m:
                    return m.group(1)
            return None
        res = self._call_results(ex, env)
        if res and len(res) == 1:
            return res[0]
        return None

    # -- the checks
    def run(self):
        for f in self.pkg.funcs:
            env, body = self._env(f)
            self._check_c_calls(f, env, body)
            self._check_selectors(f, env, body)
            self._check_unused_locals(f, body)
            self._check_local_calls(f, env, body)
            self._check_returns(f)
        for fname, toks in self.pkg.files.items():      # every C.AH_* of the file, package-level initialisers included
            self._check_c_names(fname, toks)
            self._check_ext_symbols(fname, toks)
        return self

    # -- the arrow-go packages the shim imports: every pkg.Name must be exported by that package, every member selected on a value of
    #    one of its types must be a field or a method of that type (tests/golden/go_reference_exports.json, written by
    #    scripts/gen_go_exports.py from the reference's sources)
    def _ext_type(self, path, name):
        info = self.ext.get(path, {}).get("types", {}).get(name)
        seen = 0
        while info is not None and not info["fields"] and not info["methods"] and info.get("underlying") and seen < 4:
            u = info["underlying"]
            if "." in u:
                return None
            info = self.ext.get(path, {}).get("types", {}).get(u)
            seen += 1
        return info

    def _ext_members(self, path, name, depth=0):
        """(fields {name: type}, methods set, open) — open: an embedded type could not be resolved, so absence proves nothing"""
        info = self._ext_type(path, name)
        if info is None or depth > 6:
            return {}, set(), True
        fields, methods, is_open = dict(info["fields"]), set(info["methods"]), False
        for e in info.get("embeds", []):
            f2, m2, o2 = self._ext_members(path, e, depth + 1)
            for k, v in f2.items():
                fields.setdefault(k, v)
            methods |= m2
            is_open = is_open or o2
        return fields, methods, is_open

    _WELL_KNOWN_PACKAGES = {"fmt", "errors", "unsafe", "runtime", "sync", "math", "sort", "strings", "context", "os", "reflect", "time", "bytes", "atomic",
                            "arrow", "compute", "exec", "scalar", "bitutil", "array", "memory"}

    def _check_ext_symbols(self, fname, toks):
        # a package used but not imported by THIS file ("undefined: fmt"): imports are per file in Go
        have = set(self.pkg.imports.get(fname, {}))
        for i in range(len(toks) - 1):
            t = toks[i]
            if (t.kind == "ident" and t.text in self._WELL_KNOWN_PACKAGES and t.text not in have and toks[i + 1].kind == "op" and toks[i + 1].text == "."
                    and not (i > 0 and toks[i - 1].kind == "op" and toks[i - 1].text == ".")):
                self.errors.append(f"{fname}:{t.line}: undefined: {t.text} (used but not imported by this file)")
                have.add(t.text)
        imports = {loc: pth for loc, (pth, _) in self.pkg.imports.get(fname, {}).items() if pth in self.ext}
        for i in range(len(toks) - 2):
            t = toks[i]
            if t.kind != "ident" or t.text not in imports or (i > 0 and toks[i - 1].kind == "op" and toks[i - 1].text == "."):
                continue
            if not (toks[i + 1].kind == "op" and toks[i + 1].text == "." and toks[i + 2].kind == "ident"):
                continue
            self.ext_checked += 1
            if toks[i + 2].text not in self.ext[imports[t.text]]["symbols"]:
                self.errors.append(f"{fname}:{toks[i + 2].line}: {t.text}.{toks[i + 2].text} is not exported by {imports[t.text]}")
                continue
            self._check_ext_arity(fname, toks, i + 2, imports[t.text], toks[i + 2].text, f"{t.text}.{toks[i + 2].text}")

    def _check_ext_arity(self, fname, toks, at, path, key, shown):
        """toks[at] names a function / method of an imported package; if a call follows, its argument count must fit the declaration"""
        shape = self.ext.get(path, {}).get("funcs", {}).get(key)
        k = at + 1
        if k < len(toks) and toks[k].kind == "op" and toks[k].text == "[":     # explicit type arguments
            k = _match(toks, k) + 1
        if shape is None or not (k < len(toks) and toks[k].kind == "op" and toks[k].text == "("):
            return
        close = _match(toks, k)
        args = [g for g in _split_commas(toks[k + 1:close]) if g]
        if len(args) == 1 and args[0][-1].kind == "op" and args[0][-1].text == ")" and len(args[0]) > 2:
            return                                   # f(g()) may spread a multi-value result
        if any(g[-1].kind == "op" and g[-1].text == "..." for g in args):
            return                                   # f(xs...)
        nparams, variadic = shape
        self.ext_checked += 1
        if (variadic and len(args) < nparams - 1) or (not variadic and len(args) != nparams):
            self.errors.append(f"{fname}:{toks[at].line}: {shown} called with {len(args)} arguments, arrow-go declares {nparams}{' (variadic)' if variadic else ''}")

    def _check_c_names(self, fname, toks):
        for i in range(len(toks) - 2):
            if toks[i].kind == "ident" and toks[i].text == "C" and toks[i + 1].text == "." and toks[i + 2].kind == "ident":
                nm = toks[i + 2].text
                if nm.startswith("AH_") and nm not in self.consts:
                    self.errors.append(f"{fname}:{toks[i].line}: C.{nm} is not declared by include/arrowhip.h")

    def _check_c_calls(self, f, env, body):
        i, n = 0, len(body)
        while i < n - 3:
            if body[i].kind == "ident" and body[i].text == "C" and body[i + 1].text == "." and body[i + 2].kind == "ident" \
                    and body[i + 2].text.startswith("ah_") and body[i + 3].kind == "op" and body[i + 3].text == "(" \
                    and not (i > 0 and body[i - 1].kind == "op" and body[i - 1].text == "*"):
                name, line = body[i + 2].text, body[i].line
                k = _match(body, i + 3)
                args = _split_commas(body[i + 4:k])
                proto = self.protos.get(name)
                if proto is None:
                    self.errors.append(f"{f.file}:{line}: C.{name} is not declared by include/arrowhip.h")
                elif len(args) != len(proto.params):
                    self.errors.append(f"{f.file}:{line}: C.{name} called with {len(args)} arguments, the header declares {len(proto.params)}")
                else:
                    verified, unverified = 0, []
                    for pos, (a, want) in enumerate(zip(args, proto.params)):
                        gty = self._expr_type(a, env)
                        verdict = self._compatible(gty, want)
                        if verdict is None:
                            unverified.append(f"arg {pos + 1} `{_join(a)}` for {want}")
                        elif verdict is False:
                            self.errors.append(f"{f.file}:{line}: C.{name} argument {pos + 1} `{_join(a)}` has type {gty}, the header wants {want}")
                        else:
                            verified += 1
                    self.calls.append(CallReport(f.file, line, (f.recv_type + "." if f.recv_type else "") + f.name, name, len(args), verified, unverified))
                i += 3
            else:
                i += 1

    @staticmethod
    def _compatible(gty, want):
        """True / False / None (cannot tell)"""
        if gty is None:
            return None
        is_ptr = want.endswith("*")
        if gty == "untyped nil":
            return True if is_ptr else False
        if gty == "untyped const":
            return False if is_ptr else True
        if gty.startswith("untyped"):
            return False
        c = go_type_to_c(gty)
        if c is None:
            return False if (gty.replace(" ", "").lstrip("*") in ("int", "int64", "int32", "uint64", "float64", "bool", "uintptr", "[]byte", "string")) else None
        return c == want

    def _check_selectors(self, f, env, body):
        """x.name where x's type is a struct of this package: name must be one of its fields or methods"""
        n = len(body)
        for i in range(n - 2):
            t = body[i]
            if t.kind != "ident" or t.text not in env or (i > 0 and body[i - 1].kind == "op" and body[i - 1].text == "."):
                continue
            if not (body[i + 1].kind == "op" and body[i + 1].text == "." and body[i + 2].kind == "ident"):
                continue
            ty = env.get(t.text)
            j = i
            ext_path = None      # set while the chain walks through types of an imported package
            while j + 2 < n and body[j + 1].kind == "op" and body[j + 1].text == "." and body[j + 2].kind == "ident":
                base = self._strip(ty) or ""
                if base not in self.pkg.structs or ext_path:
                    ext_path, ty, j, stop = self._ext_step(f, t, body, j, base, ext_path)
                    if stop:
                        break
                    continue
                nm = body[j + 2].text
                fields = self.pkg.structs[base]
                if nm in fields:
                    ty = fields[nm]
                elif nm in self.methods.get(base, {}):
                    self._check_local_arity(f.file, body, j + 2, self.methods[base][nm], f"{base}.{nm}", env)
                    break
                else:
                    self.errors.append(f"{f.file}:{body[j + 2].line}: {t.text}.{nm}: type {base} has no field or method {nm}")
                    break
                j += 2

    def _check_local_arity(self, fname, toks, at, decl, shown, env=None):
        """toks[at] names a function / method of THIS package; a call's argument count must fit its declaration"""
        k = at + 1
        if not (k < len(toks) and toks[k].kind == "op" and toks[k].text == "("):
            return
        close = _match(toks, k)
        args = [g for g in _split_commas(toks[k + 1:close]) if g]
        if len(args) == 1 and args[0][-1].kind == "op" and args[0][-1].text == ")" and len(args[0]) > 2:
            return                                   # f(g()) may spread a multi-value result
        if any(g[-1].kind == "op" and g[-1].text == "..." for g in args):
            return
        self.local_calls_checked += 1
        if (decl.variadic and len(args) < decl.nparams - 1) or (not decl.variadic and len(args) != decl.nparams):
            self.errors.append(f"{fname}:{toks[at].line}: {shown} called with {len(args)} arguments, declared with {decl.nparams}{' (variadic)' if decl.variadic else ''}")
            return
        # argument TYPES, where both sides are known: Go converts nothing implicitly between named types (int64 → int is an error);
        # untyped constants, nil and interface-typed parameters are left alone
        ptypes = list(decl.params.values())
        if env is None or decl.variadic or len(ptypes) != len(args):
            return
        for pos, (a, want) in enumerate(zip(args, ptypes)):
            got = self._expr_type(a, env)
            if got is None or want is None or got.startswith("untyped") or want in ("any", "interface{}", "error") or "func(" in want:
                continue
            self.local_args_typed += 1
            if self._same_go_type(fname, got, want):
                continue
            self.errors.append(f"{fname}:{toks[at].line}: {shown} argument {pos + 1} `{_join(a)}` has type {got}, the parameter is {want}")

    def _same_go_type(self, fname, a, b):
        na, nb = a.replace(" ", ""), b.replace(" ", "")
        if na == nb:
            return True
        imports = {loc: pth for loc, (pth, _) in self.pkg.imports.get(fname, {}).items() if pth in self.ext}

        def canon(t):      # follow `type X = Y` of the imported packages (exec.ExecResult = exec.ArraySpan)
            m = re.fullmatch(r"((?:\*|\[\d*\])*)([A-Za-z_]\w*)\.([A-Za-z_]\w*)", t)
            if not m or m.group(2) not in imports:
                return t
            info = self.ext[imports[m.group(2)]]["types"].get(m.group(3))
            if info and info.get("alias") and "." not in info.get("underlying", "."):
                return m.group(1) + m.group(2) + "." + info["underlying"]
            return t
        if canon(na) == canon(nb):
            return True
        # an interface parameter of this package or of arrow-go accepts whatever implements it: not decided here
        base = nb.lstrip("*")
        if base in self.pkg.named_types and self.pkg.named_types[base].startswith("interface"):
            return True
        m = re.fullmatch(r"([A-Za-z_]\w*)\.([A-Za-z_]\w*)", nb)
        if m and m.group(1) in imports:
            info = self.ext[imports[m.group(1)]]["types"].get(m.group(2))
            if info is None or (not info["fields"] and not info.get("underlying")):      # an interface (or unknown): left alone
                return True
        return False

    def _check_local_calls(self, f, env, body):
        """name(…) where name is a package-level function of this package (and not shadowed by a local)"""
        pkg_funcs = {g.name: g for g in self.pkg.funcs if not g.recv_type}
        for i, t in enumerate(body):
            if t.kind != "ident" or t.text not in pkg_funcs or t.text in env:
                continue
            if i > 0 and body[i - 1].kind == "op" and body[i - 1].text == ".":
                continue
            if i > 0 and body[i - 1].kind == "ident" and body[i - 1].text == "func":
                continue
            self._check_local_arity(f.file, body, i, pkg_funcs[t.text], t.text, env)

    def _check_returns(self, f):
        """every `return` of the function itself (function literals inside it have their own results and are skipped) carries as many
        values as the signature has results"""
        raw, i = f.body, 0
        n = len(raw)
        while i < n:
            t = raw[i]
            if t.kind == "ident" and t.text == "func" and i + 1 < n and raw[i + 1].kind == "op" and raw[i + 1].text == "(":
                k = _match(raw, i + 1) + 1
                while k < n and not (raw[k].kind == "op" and raw[k].text == "{"):
                    if raw[k].kind == "op" and raw[k].text == "(":
                        k = _match(raw, k)
                    k += 1
                if k < n:
                    i = _match(raw, k) + 1
                    continue
            if t.kind == "ident" and t.text == "return":
                j, depth, expr = i + 1, 0, []
                while j < n:
                    u = raw[j]
                    if u.kind == "op" and u.text in "([{":
                        depth += 1
                    elif u.kind == "op" and u.text in ")]}":
                        if depth == 0:
                            break
                        depth -= 1
                    if depth == 0 and (u.kind == "nl" or (u.kind == "op" and u.text == ";")):
                        break
                    if u.kind != "nl":
                        expr.append(u)
                    j += 1
                parts = [g for g in _split_commas(expr) if g]
                want = len(f.results)
                self.returns_checked += 1
                spreads = len(parts) == 1 and parts[0][-1].kind == "op" and parts[0][-1].text == ")" and want > 1 and not (
                    parts[0][0].kind == "ident" and parts[0][0].text in self._CONVERSIONS and parts[0][1].kind == "op" and parts[0][1].text == "(")
                if not spreads and len(parts) != want and not (not parts and f.named_results):
                    self.errors.append(f"{f.file}:{t.line}: {f.name} returns {len(parts)} values here, its signature has {want}")
            i += 1

    _CONVERSIONS = {"int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64", "float32", "float64", "bool", "string", "byte", "uintptr"}
    _GO_KEYWORDS = {"if", "for", "switch", "select", "case", "go", "defer", "return", "else", "range", "var", "func"}

    def _check_unused_locals(self, f, body):
        """`declared and not used` is a compile error in Go: a local introduced by := or var must occur again somewhere in the function
        (flat scoping: a name declared in two scopes and used in one of them is not seen)"""
        n, decl = len(body), {}
        for i, t in enumerate(body):
            if t.kind == "op" and t.text == ":=":
                k = i - 1
                while k >= 0 and body[k].line == t.line and (body[k].kind == "ident" or (body[k].kind == "op" and body[k].text == ",")):
                    if body[k].kind == "ident" and body[k].text != "_" and body[k].text not in self._GO_KEYWORDS:
                        decl.setdefault(body[k].text, []).append(body[k].line)
                    k -= 1
            elif t.kind == "ident" and t.text == "var" and i + 1 < n and body[i + 1].kind == "ident":
                j = i + 1
                while True:
                    if body[j].text != "_":
                        decl.setdefault(body[j].text, []).append(body[j].line)
                    if j + 2 < n and body[j + 1].kind == "op" and body[j + 1].text == "," and body[j + 2].kind == "ident":
                        j += 2
                    else:
                        break
        for name, lines in decl.items():
            if sum(1 for t in body if t.kind == "ident" and t.text == name) <= len(lines):
                self.errors.append(f"{f.file}:{lines[0]}: {name} declared and not used (in {f.name})")

    def _ext_step(self, f, root, body, j, base, ext_path):
        """one `.name` step on a value whose type belongs to an imported package → (package path, member type, new j, stop)"""
        imports = {loc: pth for loc, (pth, _) in self.pkg.imports.get(f.file, {}).items() if pth in self.ext}
        base = re.sub(r"^(\[\d*\]|\*)+", "", base)
        if "." in base:
            loc, name = base.split(".", 1)
            path = imports.get(loc)
            if path is None and ext_path:      # a type of a third package named from inside the external one (scalar.Scalar from exec)
                path = next((p for p in self.ext if p.rsplit("/", 1)[-1] == loc), None)
        else:
            path, name = ext_path, base
        name = name.split("[")[0]
        if path is None or self._ext_type(path, name) is None:
            return None, None, j, True
        fields, methods, is_open = self._ext_members(path, name)
        nm = body[j + 2].text
        self.ext_checked += 1
        if nm in fields:
            ty = fields[nm]
            j += 2
            # an index right behind a slice / array field: its element
            if j + 1 < len(body) and body[j + 1].kind == "op" and body[j + 1].text == "[" and re.match(r"\[\d*\]", ty or ""):
                j = _match(body, j + 1)
                ty = re.sub(r"^\[\d*\]", "", ty)
            return path, ty, j, False
        if nm in methods or is_open:
            if nm in methods:      # declared on this type itself (not through an embedded one): the call's shape is known
                self._check_ext_arity(f.file, body, j + 2, path, f"{name}.{nm}", f"{root.text}….{nm}")
            return path, None, j, True
        self.errors.append(f"{f.file}:{body[j + 2].line}: {root.text}….{nm}: type {path.rsplit('/', 1)[-1]}.{name} has no field or method {nm}")
        return path, None, j, True

    # -- reports
    def bound(self):
        return sorted({c.cname for c in self.calls})

    def unbound(self):
        return sorted(set(self.protos) - set(self.bound()))


def check(go_dir: str = GO_DIR, header: str = HEADER) -> Checker:
    protos, consts = parse_header(header)
    return Checker(load_package(go_dir), protos, consts).run()


def binding_table(chk: Checker) -> str:
    """markdown for INTEGRATION.md: every entry point of the header, the Go method(s) that call it, or 'unbound' with the reason"""
    by_c = {}
    for c in chk.calls:
        by_c.setdefault(c.cname, set()).add(c.func)
    lines = ["| C entry point (include/arrowhip.h) | Go caller (go/arrowhip) |", "|---|---|"]
    for name in sorted(chk.protos):
        lines.append(f"| `{name}` | " + (", ".join(f"`{g}`" for g in sorted(by_c[name])) if name in by_c else "— unbound —") + " |")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    import sys
    c = check()
    for e in c.errors:
        print("ERROR", e)
    unv = [(r, u) for r in c.calls for u in r.unverified]
    for r, u in unv:
        print("unverified", f"{r.file}:{r.line}", r.cname, u)
    tot = sum(r.nargs for r in c.calls)
    print(f"{len(c.calls)} C calls, {tot} arguments, {sum(r.verified for r in c.calls)} type-checked, {len(unv)} unverified; "
          f"{len(c.bound())} of {len(c.protos)} entry points bound; {len(c.errors)} errors")
    if "--unbound" in sys.argv:
        print("unbound:", " ".join(c.unbound()))
    if "--table" in sys.argv:
        print(binding_table(c))
    sys.exit(1 if c.errors else 0)
