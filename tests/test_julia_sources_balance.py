"""julia/PDMPFluxHIP.jl and julia/dump_tape.jl cannot be parsed here (no Julia toolchain in the image or on the GPU boxes), so typos in
them are invisible to every other test.  This is the check that CAN run: a small lexer strips comments, strings and character literals
and verifies that parentheses / brackets / braces balance and that every block opener (`function`, `if`, `for`, `while`, `struct`,
`module`, `let`, `begin`, `do`, `try`, `quote`, `macro`) outside brackets is closed by an `end` -- plus that every C entry point the
shim `ccall`s is declared in include/pdmp_hip.h with the same number of arguments.  Not a parser: a tripwire."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPENERS = {"function", "if", "for", "while", "struct", "module", "let", "begin", "do", "try", "quote", "macro"}


def strip(src):
    """Comments, strings (incl. triple-quoted and interpolation-free handling) and character literals -> spaces."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("#=", i):
            j = src.index("=#", i) + 2
            out.append(" " * (j - i)); i = j
        elif c == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith('"""', i):
            j = src.index('"""', i + 3) + 3
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append(" " * (j + 1 - i)); i = j + 1
        elif c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")):
            j = i + (3 if src[i + 2] == "'" else 4)
            out.append(" " * (j - i)); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def check_balance(path):
    src = strip(open(path, encoding="utf-8").read())
    depth, stack = 0, []
    for m in re.finditer(r"[()\[\]{}]|[A-Za-z_][A-Za-z_0-9!]*", src):
        tok = m.group(0)
        line = src.count("\n", 0, m.start()) + 1
        if tok in "([{":
            stack.append((tok, line)); continue
        if tok in ")]}":
            assert stack and stack[-1][0] in "([{" and "([{".index(stack[-1][0]) == ")]}".index(tok), f"{path}:{line}: unbalanced {tok!r}"
            stack.pop(); continue
        in_brackets = any(t in "([{" for t, _ in stack)
        if in_brackets:
            continue                                   # generators / comprehensions / a[end]: no block structure inside brackets
        prev = src[max(0, m.start() - 1):m.start()]
        if prev in (".", ":"):                         # field access / symbol, e.g. :function
            continue
        if tok in OPENERS:
            if tok == "struct" and stack and stack[-1][0] == "mutable":
                stack.pop()
            stack.append((tok, line))
        elif tok in ("mutable", "abstract", "primitive"):
            stack.append((tok, line))
        elif tok == "type" and stack and stack[-1][0] in ("abstract", "primitive"):      # abstract type T end
            stack.pop()
            stack.append(("struct", line))
        elif tok == "end":
            assert stack and stack[-1][0] in OPENERS, f"{path}:{line}: `end` without an open block"
            stack.pop()
    assert not stack, f"{path}: unclosed {stack[-1][0]!r} opened at line {stack[-1][1]}"
    return src


def test_julia_sources_are_balanced():
    for f in ("PDMPFluxHIP.jl", "dump_tape.jl"):
        check_balance(os.path.join(ROOT, "julia", f))


def test_every_ccall_names_a_declared_entry_point_with_the_right_arity():
    header = open(os.path.join(ROOT, "include", "pdmp_hip.h")).read()
    decl = {}
    for m in re.finditer(r"\b(?:int|long long|const char \*)\s*(pdmp_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else len([a for a in re.sub(r"/\*.*?\*/", "", args, flags=re.S).split(",")])
    src = strip(open(os.path.join(ROOT, "julia", "PDMPFluxHIP.jl"), encoding="utf-8").read())
    calls = list(re.finditer(r"ccall\(\(:(pdmp_[a-z_0-9]+), LIB\),\s*[A-Za-z]+,\s*\(", src))
    assert len(calls) >= 20
    for m in calls:
        name = m.group(1)
        assert name in decl, f"{name} is not declared in include/pdmp_hip.h"
        i, depth = m.end(), 1                           # the argument-type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0); i += 1
        tup = src[m.end():i - 1]
        parts, d, cur = [], 0, ""
        for ch in tup:
            d += {"(": 1, "{": 1, ")": -1, "}": -1}.get(ch, 0)
            if ch == "," and d == 0:
                parts.append(cur); cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        assert len(parts) == decl[name], f"ccall of {name}: {len(parts)} argument types, the header declares {decl[name]}"
