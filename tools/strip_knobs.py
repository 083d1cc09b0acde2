#!/usr/bin/env python3
"""A small `unifdef`: removes the preprocessor branches that depend on the named macros from a C/C++ header, treating
each as UNDEFINED (or, with NAME=1, as defined).  Used once per round to take rejected experiment branches out of the
product kernels; the branches live on as patches under tools/experiments/ (diff of the cleaned header against the old).

    tools/strip_knobs.py file.h GHR_B3_NOATOM GHR_B3_PIX_NT=1 ...      # rewrites file.h in place

Understands #ifdef X, #ifndef X, #if defined(X), #elif defined(X), #else, #endif; conditions on other macros are kept
verbatim (and nest correctly)."""
import re
import sys


def main():
    path, names = sys.argv[1], sys.argv[2:]
    state = {n.split("=")[0]: (n.endswith("=1")) for n in names}
    out = []
    # stack entries: dict(known=bool, taken=bool (a branch of this chain was emitted), live=bool (current branch emitted))
    stack = []

    def live():
        return all(f["live"] for f in stack)

    def cond(line):
        m = re.match(r"\s*#\s*(ifdef|ifndef)\s+(\w+)", line)
        if m and m.group(2) in state:
            v = state[m.group(2)]
            return v if m.group(1) == "ifdef" else not v
        m = re.match(r"\s*#\s*(?:if|elif)\s+(!?)\s*defined\s*\(?\s*(\w+)\s*\)?\s*(//.*)?$", line)
        if m and m.group(2) in state:
            v = state[m.group(2)]
            return (not v) if m.group(1) else v
        return None

    for line in open(path).read().split("\n"):
        s = line.strip()
        if re.match(r"#\s*(if|ifdef|ifndef)\b", s):
            c = cond(line)
            if c is None:
                if live():
                    out.append(line)
                stack.append(dict(known=False, taken=True, live=True, outer=live()))
            else:
                stack.append(dict(known=True, taken=c, live=c, outer=live()))
            continue
        if re.match(r"#\s*elif\b", s):
            f = stack[-1]
            if not f["known"]:
                if live():
                    out.append(line)
                continue
            c = cond(line)
            assert c is not None, "mixed #elif chain: " + line
            f["live"] = (not f["taken"]) and c
            f["taken"] = f["taken"] or c
            continue
        if re.match(r"#\s*else\b", s):
            f = stack[-1]
            if not f["known"]:
                if live():
                    out.append(line)
                continue
            f["live"] = not f["taken"]
            f["taken"] = True
            continue
        if re.match(r"#\s*endif\b", s):
            f = stack.pop()
            if not f["known"] and live():
                out.append(line)
            continue
        if live():
            out.append(line)
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
