"""Regenerates INTEGRATION.md's appendix of environment switches from the sources (tests/test_abi.py checks that it is complete)."""
import re, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'morbit.jl_amd/csrc/*.hip')) + glob.glob(os.path.join(ROOT, 'morbit.jl_amd/csrc/*.hpp'))):
    for i, line in enumerate(open(f), 1):
        for m in re.finditer(r'mrbf_env\("(MRBF_[A-Z0-9_]+)"\)', line):
            names.setdefault(m.group(1), (os.path.basename(f), i))
rows = ["| `%s` | `csrc/%s:%d` |" % (n, names[n][0], names[n][1]) for n in sorted(names)]
head = "\n## Appendix: every environment switch the library reads"
txt = head + """

All of them are honoured only while `MRBF_EXPERIMENTS=1` is set (`mrbf_env()`, `csrc/common.hpp`); without it the library ignores its
whole `MRBF_*` environment. They exist for A/B runs, diagnostics and the tests that hold alternative implementations to the same
results; none is part of the interface. Where each one is read (the comment at that line says what it selects; regenerate this list
with `python tools/env_appendix.py`):

| switch | read at |
|---|---|
""" + "\n".join(rows) + "\n"
p = os.path.join(ROOT, 'INTEGRATION.md')
s = open(p).read()
if head in s:
    s = s[:s.index(head)]
open(p, 'w').write(s.rstrip("\n") + "\n" + txt)
print(len(rows), "switches")
