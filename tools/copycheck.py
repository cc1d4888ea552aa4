#!/usr/bin/env python3
"""Token-level similarity of the package's Python files against the reference's files (build-container tool only: it reads
/root/reference, which does not exist on the GPU box).  Comments and docstrings are stripped; a difflib match over Python tokens gives
(a) the fraction of OUR tokens that lie in matching blocks and (b) the fraction lying in matching runs of >= 30 tokens.
Usage: python tools/copycheck.py            -> every package file against every reference backend/mlagents file, worst pair per file."""
import difflib
import glob
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/backend"


def toks(path):
    out, prev = [], None
    with open(path, "rb") as f:
        src = f.read().decode("utf-8", "replace")
    try:
        for t in tokenize.generate_tokens(io.StringIO(src).readline):
            if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER):
                continue
            if t.type == tokenize.STRING and (prev is None or prev in (":", "\n")) and t.string.startswith(('"""', "'''", 'r"""')):
                continue  # docstring
            out.append(t.string)
            prev = t.string
    except tokenize.TokenError:
        pass
    return out


def compare(a, b):
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    blocks = sm.get_matching_blocks()
    tot = sum(x.size for x in blocks)
    long_ = sum(x.size for x in blocks if x.size >= 30)
    return tot / max(1, len(a)), long_ / max(1, len(a))


def main():
    ours = sorted(glob.glob(os.path.join(ROOT, "three-mlagents_amd", "*.py")) + [os.path.join(ROOT, "bench.py")])
    refs = sorted(glob.glob(os.path.join(REF, "mlagents", "*.py")) + glob.glob(os.path.join(REF, "*.py")))
    rt = {r: toks(r) for r in refs}
    worst = 0.0
    for o in ours:
        a = toks(o)
        if len(a) < 50:
            continue
        best = max(((compare(a, b), r) for r, b in rt.items() if len(b) >= 50), key=lambda x: x[0][0])
        (m, lg), r = best
        worst = max(worst, m)
        print(f"{os.path.relpath(o, ROOT):42s} vs {os.path.relpath(r, REF):32s} matched {m:5.1%}  in runs>=30: {lg:5.1%}  ({len(a)} tokens)")
    return 0 if worst < 0.30 else 1


if __name__ == "__main__":
    sys.exit(main())
