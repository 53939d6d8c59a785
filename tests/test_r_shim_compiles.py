"""CPU: the R `.Call()` shim (integration/r/src/machisplin_shim.c) goes through a C compiler.  The image has no R, so the
check uses integration/r/check/{R.h,Rinternals.h} -- declarations of the documented R C API subset the shim uses, not
R's headers -- and `gcc -fsyntax-only` with the prototype errors switched on: every mhs_* call in the shim must match
include/machisplin_hip.h, every R API call its documented signature.  And every `.Call("mhsr_...")` in
integration/r/R/{backend_hip,multi_gpu}.R must name a function the shim defines with that many SEXP arguments."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "r", "src", "machisplin_shim.c")
RFILES = [os.path.join(ROOT, "integration", "r", "R", n) for n in ("backend_hip.R", "multi_gpu.R")]


def test_shim_passes_the_c_compiler():
    cmd = ["gcc", "-fsyntax-only", "-std=c99", "-Wall", "-Wextra", "-Werror=implicit-function-declaration",
           "-Werror=incompatible-pointer-types", "-Werror=int-conversion", "-Werror=return-type",
           "-I", os.path.join(ROOT, "integration", "r", "check"), "-I", os.path.join(ROOT, "include"), SHIM]
    pr = subprocess.run(cmd, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    assert "warning" not in pr.stderr, pr.stderr


def test_every_dot_call_names_a_shim_function_with_that_arity():
    src = re.sub(r"/\*.*?\*/", "", open(SHIM).read(), flags=re.S)      # comments may hold commas
    defs = {}
    for m in re.finditer(r"^SEXP\s+(mhsr_\w+)\s*\(([^)]*)\)\s*\{", src, re.M):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        assert all(a.strip().startswith("SEXP") for a in args), m.group(0)
        defs[m.group(1)] = len(args)
    assert len(defs) >= 8
    r = "\n".join(open(f).read() for f in RFILES)
    calls = re.findall(r'\.Call\(\s*"(mhsr_\w+)"((?:[^()]|\([^()]*\))*)\)', r)
    assert calls
    for name, rest in calls:
        assert name in defs, name
        depth, nargs, cur = 0, 0, ""
        for ch in rest:
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                nargs += 1
        assert nargs == defs[name], (name, nargs, defs[name])
