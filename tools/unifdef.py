#!/usr/bin/env python3
"""A small unifdef: resolves the preprocessor conditionals of a source file whose condition can be decided from a table of symbols
(NAME=value decides `#if` expressions and `#ifdef`; NAME=undef means "never defined") and leaves every other conditional alone.
Used once per pruning pass (round 5: the build switches of variants that lost their measurements twice).

    tools/unifdef.py FILE NAME=VALUE|undef ...      (rewrites FILE in place; `#ifndef NAME / #define NAME v / #endif` default blocks go too)
"""
import re
import sys


def decide(expr, table):
    """True / False when the expression only involves known symbols, else None."""
    e = expr.split("//")[0].strip()
    names = set(re.findall(r"[A-Za-z_]\w*", e)) - {"defined"}
    if not names or not names <= set(table):
        return None
    def sub_defined(m):
        return "1" if table[m.group(1)] is not None else "0"
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", sub_defined, e)
    e = re.sub(r"defined\s+(\w+)", sub_defined, e)
    for n in sorted(names, key=len, reverse=True):
        e = re.sub(r"\b%s\b" % n, "0" if table[n] is None else str(table[n]), e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e, {"__builtins__": {}}))


def main():
    path, table = sys.argv[1], {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        table[k] = None if v == "undef" else int(v)
    lines = open(path).read().split("\n")
    out = []
    # stack entries: [known, emitting_before, taken_already, emitting_now]
    stack = []
    i = 0
    def emitting():
        return all(s[3] for s in stack)
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                val = decide(rest, table)
            else:
                name = rest.split()[0]
                val = None if name not in table else ((table[name] is not None) == (kind == "ifdef"))
                # `#ifndef NAME / #define NAME default / #endif` of a decided symbol: the whole block goes
            if val is None:
                if emitting():
                    out.append(ln)
                stack.append([False, None, None, True])
            else:
                stack.append([True, None, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if all(s_[3] for s_ in stack[:-1]):
                    out.append(ln)
            else:
                if top[2]:
                    top[3] = False
                else:
                    val = decide(rest, table)
                    if val is None:
                        raise SystemExit(f"{path}:{i + 1}: #elif mixes decided and undecided symbols: {rest}")
                    top[2] = top[3] = val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if all(s_[3] for s_ in stack[:-1]):
                    out.append(ln)
            else:
                top[3] = not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if not top[0] and emitting():
                out.append(ln)
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
