"""Source-level CPU tests of the Julia binding (morbit.jl_amd/julia/HipRbf.jl).  Julia is absent from the build container
(SURVEY.md section 8c), so the file cannot be executed; what CAN be pinned from its text is pinned here:

1. every `ccall((:mrbf_..., libmrbf), ...)` against the prototype in include/mrbf.h: arity, return type, type class of every
   argument (Int32 / Int64 / UInt64 / Float32 / Float64 / pointer), and the sizes + field offsets of the mirrored structs against
   the ctypes structs (which tests/test_abi.py pins to the header);
2. thread-correctness: a context is not thread-safe and finalizers run on any thread (the reference's benchmark runs `optimize`
   under `Threads.@threads`, /root/reference/examples/large_scale_benchmarks.jl:253) -- every ccall that passes a context handle
   sits inside `_locked(ctx) do h ... end` (holds ctx.lock) or inside a `trylock(ctx.lock)` finalizer; no unguarded global
   container;
3. inertness for users who do not select HipRbfConfig: every method the file adds to a function of Morbit / Base has a HipRbf /
   Mrbf type in its signature, except three whitelisted ones whose FIRST statements look for a HipRbf object (task-local scan state
   or a HipRbfModel in the container) and hand over to Morbit's method by `invoke` before any ccall can be reached.
"""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT

JL = os.path.join(ROOT, "morbit.jl_amd", "julia", "HipRbf.jl")
HDR = os.path.join(ROOT, "include", "mrbf.h")


def _balanced(text, start):
    """index just past the parenthesis that closes the one at text[start]"""
    depth = 0
    for i in range(start, len(text)):
        c = text[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


def _strip_jl_comments(src):
    return "\n".join(line.split("#")[0] if '"' not in line.split("#")[0] or line.split("#")[0].count('"') % 2 == 0 else line
                     for line in src.split("\n"))


def julia_ccalls():
    src = open(JL, encoding="utf-8").read()
    code = _strip_jl_comments(src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(mrbf_[a-z0-9_]+), libmrbf\)", code):
        end = _balanced(code, m.start() + len("ccall"))
        inner = code[m.start() + len("ccall("):end - 1]
        parts = _split_top(inner)
        # parts: [(:name, lib), Ret, (argtypes...), args...]
        ret = parts[1]
        at = parts[2].strip()
        assert at.startswith("(") and at.endswith(")"), (m.group(1), at)
        argtypes = _split_top(at[1:-1])
        args = parts[3:]
        calls.append(dict(name=m.group(1), ret=ret, argtypes=argtypes, args=args, pos=m.start(), line=code.count("\n", 0, m.start()) + 1))
    return src, code, calls


def c_prototypes():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mrbf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        plist = [p.strip() for p in re.sub(r"\s+", " ", params).split(",")] if params.strip() not in ("", "void") else []
        protos[name] = (ret, plist)
    return protos


def c_class(decl):
    """type class of a C parameter / return declaration"""
    if "*" in decl:
        return "ptr"
    toks = re.sub(r"\bconst\b", " ", decl).split()
    base = toks[0]
    return {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32", "int": "i32"}[base]


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr{Cvoid}"):
        return "ptr"
    return {"Int32": "i32", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Float32": "f32", "Cint": "i32"}[t]


def test_every_ccall_matches_the_header():
    _, _, calls = julia_ccalls()
    protos = c_prototypes()
    assert len(calls) >= 21
    seen = set()
    for c in calls:
        assert c["name"] in protos, "%s (HipRbf.jl:%d) is not declared in include/mrbf.h" % (c["name"], c["line"])
        ret, params = protos[c["name"]]
        where = "%s (HipRbf.jl:%d)" % (c["name"], c["line"])
        assert len(c["argtypes"]) == len(params), (where, c["argtypes"], params)
        assert len(c["args"]) == len(params), (where, "values passed", len(c["args"]), len(params))
        assert jl_class(c["ret"]) == c_class(ret + " x" if "*" not in ret else ret), (where, c["ret"], ret)
        for i, (jt, cp) in enumerate(zip(c["argtypes"], params)):
            assert jl_class(jt) == c_class(cp), (where, "argument %d" % (i + 1), jt, cp)
        seen.add(c["name"])
    # the entry points of the hot path are all bound
    for must in ("mrbf_init", "mrbf_shutdown", "mrbf_fit", "mrbf_eval", "mrbf_free_model", "mrbf_backtrack", "mrbf_ps_step_problem",
                 "mrbf_round4", "mrbf_fit_from_round4", "mrbf_free_round4", "mrbf_round4_sites", "mrbf_affine_scores", "mrbf_last_error",
                 "mrbf_dispatch_ps", "mrbf_dispatch_backtrack", "mrbf_dispatch_affine", "mrbf_dispatch_round4", "mrbf_dispatch_fit",
                 "mrbf_dispatch_after"):
        assert must in seen, must


JL_SIZES = {"Int32": 4, "UInt32": 4, "Float32": 4, "Int64": 8, "UInt64": 8, "Float64": 8}


def _jl_struct_layout(src, name):
    body = re.search(r"^struct %s\b[^\n]*\n(.*?)^end" % name, src, flags=re.S | re.M).group(1)
    fields = []
    for line in body.split("\n"):
        line = line.split("#")[0]
        for f in line.split(";"):
            f = f.strip()
            if "::" in f:
                fname, ftype = [t.strip() for t in f.split("::")]
                size = 8 if ftype.startswith("Ptr{") else JL_SIZES[ftype]
                fields.append((fname, size))
    off, layout = 0, {}
    for fname, size in fields:           # isbits struct: C layout, natural alignment
        off = (off + size - 1) // size * size
        layout[fname] = off
        off += size
    total = (off + 7) // 8 * 8 if any(s == 8 for _, s in fields) else off
    return layout, total


def test_struct_mirrors_have_the_ctypes_layout():
    from morbit.jl_amd import _lib

    src = open(JL, encoding="utf-8").read()
    for jl_name, ct in (("MrbfFitInfo", _lib.FitInfo), ("MrbfPsOptions", _lib.PsOptions), ("MrbfPsInfo", _lib.PsInfo),
                        ("MrbfPsProblem", _lib.PsProblem)):
        layout, total = _jl_struct_layout(src, jl_name)
        assert total == ctypes.sizeof(ct), (jl_name, total, ctypes.sizeof(ct))
        ct_fields = [f[0] for f in ct._fields_]
        assert list(layout) == ct_fields, (jl_name, list(layout), ct_fields)
        for fname in ct_fields:
            assert layout[fname] == getattr(ct, fname).offset, (jl_name, fname, layout[fname], getattr(ct, fname).offset)


def _enclosing_function(code, pos):
    """text from the start of the top-level definition that contains `pos` up to `pos`"""
    starts = [m.start() for m in re.finditer(r"^(?:function |mutable struct |struct |[A-Za-z_][\w\.!:\(\)=]*\([^\n]*\)\s*(?:where[^\n=]*)?=)", code, flags=re.M)]
    begin = max([s for s in starts if s <= pos] or [0])
    return code[begin:pos]


def test_every_context_call_holds_the_context_lock():
    _, code, calls = julia_ccalls()
    protos = c_prototypes()
    checked = 0
    for c in calls:
        _, params = protos[c["name"]]
        if not params or not re.match(r"(const )?mrbf_ctx \*", params[0]):
            continue                                   # host-only entry points (decision table, round4_sites, last_error(NULL) ...)
        handle = c["args"][0]
        where = "%s (HipRbf.jl:%d)" % (c["name"], c["line"])
        if c["name"] == "mrbf_last_error" and handle == "C_NULL":
            continue
        before = _enclosing_function(code, c["pos"])
        m = None
        for m in re.finditer(r"_locked\(([^\n]*?)\) do (\w+)", before):
            pass
        if m is not None and handle == m.group(2):
            # the ccall sits inside the innermost open `_locked(...) do h` block: no `end` at that block's indentation in between
            blk = before[m.start():]
            indent = len(before[:m.start()].split("\n")[-1]) - len(before[:m.start()].split("\n")[-1].lstrip())
            closed = any(re.match(r"^ {0,%d}end\b" % indent, ln) for ln in blk.split("\n")[1:])
            assert not closed, where
            checked += 1
            continue
        # finalizers / shutdown: non-blocking trylock on the same lock, handle read from the context under it
        assert "trylock(ctx.lock)" in before and handle == "ctx.handle", (where, handle)
        checked += 1
    assert checked >= 11
    # the lock helper exists as described and every compute entry point appears under it
    src = open(JL, encoding="utf-8").read()
    assert re.search(r"function _locked\(f, ctx::MrbfContext\)\s+lock\(ctx\.lock\)\s+try\b.*?finally\s+unlock\(ctx\.lock\)", src, flags=re.S)
    for entry in ("mrbf_fit", "mrbf_eval", "mrbf_backtrack", "mrbf_ps_step_problem", "mrbf_round4", "mrbf_fit_from_round4", "mrbf_affine_scores"):
        assert re.search(r"_locked\([^\n]*\) do \w+\s+ccall\(\(:%s, libmrbf\)" % entry, src), entry


def test_no_unguarded_global_state():
    src = open(JL, encoding="utf-8").read()
    code = _strip_jl_comments(src)
    consts = re.findall(r"^const (\w+)\s*=\s*([^\n]*)", code, flags=re.M)
    containers = [(n, rhs) for n, rhs in consts if re.search(r"\b(IdDict|Dict|Set|Vector|Ref)\b\s*[{(]", rhs)]
    for name, rhs in containers:
        if name == "MRBF_KERNEL_ID":       # read-only table built at load time
            continue
        lock_name = {"_CTX": "_CTX_LOCK", "_ROUND4_KEPT": "_ROUND4_LOCK"}.get(name)
        assert lock_name, "global mutable container %s has no lock (use task-local storage or add one)" % name
        # every use outside its definition sits inside `lock(<lock>) do`
        for m in re.finditer(r"\b%s\b" % re.escape(name), code):
            line_start = code.rfind("\n", 0, m.start()) + 1
            line = code[line_start:code.find("\n", m.start())]
            if line.startswith("const "):
                continue
            before = code[max(0, m.start() - 400):m.start()]
            assert "lock(%s) do" % lock_name in before, (name, line.strip())
    assert "_AFFINE_SEEDS" not in code                 # round-5 finding: the unguarded global is gone (task-local now)
    assert "task_local_storage(_HIP_AFFINE_KEY" in code


# functions of Morbit / Base that HipRbf.jl adds methods to (collected from the reference when it is present, see below)
MORBIT_FUNCTIONS = {"prepare_init_model", "prepare_update_model", "prepare_improve_model", "init_model", "update_model", "improve_model",
                    "eval_models", "get_jacobian", "get_gradient", "max_evals", "combinable", "get_saveable_type", "fully_linear",
                    "num_outputs", "set_fully_linear!", "_rbf_round4", "_backtrack", "get_criticality", "_get_signature",
                    "Base.iterate", "Base.hash", "Base.isequal", "Base.:(==)"}
GUARD_FIRST = {"Base.iterate": r"scan = _hip_affine_scan\(\)\s+scan === nothing && return invoke\(Base\.iterate,",
               "_backtrack": r"_touches_device\(sc; objectives_only = true\) \|\|\s+return invoke\(_backtrack,",
               "get_criticality": r"reference\(\) = invoke\(get_criticality,[^\n]*\n\s+_touches_device\(sc\) \|\| return reference\(\)"}


def _method_definitions(code):
    """(name, signature text, body start) of every top-level method definition"""
    defs = []
    for m in re.finditer(r"^(?:function\s+)?((?:Base\.)?(?::\(==\)|[A-Za-z_][\w!]*))\(", code, flags=re.M):
        line_start = code.rfind("\n", 0, m.start()) + 1
        head = code[line_start:m.start()]
        if head.strip() not in ("", "function"):
            continue
        is_func = code[line_start:].startswith("function")
        end = _balanced(code, m.end() - 1)
        rest = code[end:end + 200]
        if not is_func and not re.match(r"\s*(where\s*\{[^}]*\}\s*)?(::[\w{},\s]+)?\s*=(?!=)", rest):
            continue                                    # a call at top level, not a definition
        defs.append((m.group(1), code[m.end() - 1:end], end))
    return defs


def test_inert_for_users_who_do_not_select_hiprbfconfig():
    src = open(JL, encoding="utf-8").read()
    code = _strip_jl_comments(src)
    names = set(MORBIT_FUNCTIONS)
    ref_src = "/root/reference/src"
    if os.path.isdir(ref_src):                          # build container: every function name the reference defines
        for base, _, files in os.walk(ref_src):
            for f in files:
                if f.endswith(".jl"):
                    t = open(os.path.join(base, f), encoding="utf-8", errors="ignore").read()
                    names.update(re.findall(r"^\s*function\s+([A-Za-z_][\w!]*)\s*\(", t, flags=re.M))
                    names.update(re.findall(r"^([A-Za-z_][\w!]*)\([^\n]*\)\s*(?:where[^\n=]*)?=(?!=)", t, flags=re.M))
    defs = _method_definitions(code)
    assert len(defs) > 40
    guarded = set()
    for name, sig, body_at in defs:
        if name not in names and not name.startswith("Base."):
            continue                                    # a function this file introduces
        if re.search(r"HipRbf|Mrbf|HipRound4", sig):
            continue                                    # only applies to this plug-in's own types
        assert name in GUARD_FIRST, "method %s%s extends a Morbit / Base function on foreign types" % (name, sig[:80])
        body = code[body_at:body_at + 1200]
        body = re.sub(r"^\s*where\s*\{[^}]*\}", "", body)
        m = re.search(GUARD_FIRST[name], body)
        assert m, (name, body[:300])
        # nothing of libmrbf can be reached before the guard
        assert not re.search(r"ccall|_dispatch_|_container_plan|mrbf_context", body[:m.start()]), (name, body[:m.start()])
        guarded.add(name)
    assert guarded == set(GUARD_FIRST)
    # the guard helpers themselves are pure Julia
    for helper in ("_hip_affine_scan", "_touches_device"):
        h = re.search(r"^(?:function )?%s\(.*?(?=^\S)" % helper, code, flags=re.S | re.M).group(0)
        assert "ccall" not in h and "_dispatch_" not in h, helper
    # and the task-local flag is only ever set by the HipRbfConfig methods
    setters = re.findall(r"^[^\n]*task_local_storage\(_HIP_AFFINE_KEY[^\n]*", code, flags=re.M)
    assert len(setters) == 1
    assert re.search(r"_prepare_with_device_scan\(meta, cfg::HipRbfConfig", code)
