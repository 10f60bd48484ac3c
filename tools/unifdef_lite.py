#!/usr/bin/env python
"""Developer aid: resolve the preprocessor branches of a given set of UNDEFINED macros in a source file (a tiny `unifdef -U`):
`#ifdef M` / `#ifndef M` / `#if M == k` / `#elif M == k` / `#if defined(M) ...` (only when the whole condition is decided by
the listed macros).  Used in round 5 to move the tied experiments out of the product kernels (tools/experiments/ keeps the
originals).  usage: python tools/unifdef_lite.py FILE MACRO [MACRO ...]   (rewrites FILE in place)"""
import re
import sys


def decide(cond, undef):
    """True / False if `cond` (the text after #if / #elif) is decided with every macro of `undef` undefined, else None."""
    names = set(re.findall(r"[A-Za-z_]\w*", cond)) - {"defined"}
    if not names or not names <= undef:
        return None
    expr = re.sub(r"defined\s*\(\s*\w+\s*\)", "0", cond)
    expr = re.sub(r"defined\s+\w+", "0", expr)
    expr = re.sub(r"[A-Za-z_]\w*", "0", expr)
    expr = expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", " !=")
    try:
        return bool(eval(expr, {"__builtins__": {}}))
    except Exception:      # noqa: BLE001
        return None


def run(text, undef):
    out, stack = [], []        # stack entries: [decided (bool: we own this #if), emitting, any_branch_taken, parent_emitting]
    emitting = True
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2).split("//")[0].strip()
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "ifdef":
                val = False if rest in undef else None
            elif kind == "ifndef":
                val = True if rest in undef else None
            else:
                val = decide(rest, undef)
            if val is None:
                stack.append([False, emitting, False, emitting])
                if emitting:
                    out.append(line)
            else:
                stack.append([True, emitting and val, val, emitting])
                emitting = emitting and val
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if top[3]:
                    out.append(line)
            else:
                val = decide(rest, undef)
                if val is None:
                    raise SystemExit(f"undecidable #elif in an owned chain: {line}")
                take = (not top[2]) and val
                top[2] = top[2] or val
                emitting = top[3] and take
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if top[3]:
                    out.append(line)
            else:
                emitting = top[3] and not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if not top[0]:
                if top[3]:
                    out.append(line)
            emitting = top[3]
    return "\n".join(out)


if __name__ == "__main__":
    path, macros = sys.argv[1], set(sys.argv[2:])
    src = open(path).read()
    open(path, "w").write(run(src, macros))
