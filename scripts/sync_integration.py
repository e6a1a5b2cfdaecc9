"""INTEGRATION.md carries julia/NEPMI355X.jl verbatim in its first ```julia block (one source of truth, checked by
tests/test_host_logic.py::test_julia_binding_symbols_exist): this rewrites the block from the file."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
jl = open(os.path.join(ROOT, "julia", "NEPMI355X.jl")).read()
body = jl[jl.index("# src/backends/MI355X.jl"):].rstrip("\n")
p = os.path.join(ROOT, "INTEGRATION.md")
md = open(p).read()
head, rest = md.split("```julia\n", 1)
_, tail = rest.split("\n```", 1)
open(p, "w").write(head + "```julia\n" + body + "\n```" + tail)
print("INTEGRATION.md: julia block rewritten (%d lines)" % (body.count("\n") + 1))
