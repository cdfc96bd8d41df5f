"""Static check of julia/QILaplaceHIP.jl against include/qilaplace_hip.h (no Julia needed): every
`ccall((:qil_..., LIB), Ret, (Types...), args...)` of the shim must name a declared entry point and agree with
its C declaration in return type, arity, scalar width / signedness and pointer-ness (and in the pointee where the
Julia type names one).  INTEGRATION.md's binding table may only name entry points the shim really binds."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qilaplace_hip.h")
SHIM = os.path.join(ROOT, "julia", "QILaplaceHIP.jl")

# C scalar type -> canonical kind
C_SCALARS = {"int": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "uint8_t": "u8"}
# Julia scalar type -> canonical kind
JL_SCALARS = {"Cint": "i32", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "UInt8": "u8"}


def _c_arg_kind(arg):
    """'const int64_t* bond_dims' -> ('ptr', 'i64');  'int dtype' -> ('val', 'i32');  'qil_mps** out' -> ('ptr', 'ptr')"""
    arg = re.sub(r"/\*.*?\*/", "", arg).strip()
    stars = arg.count("*")
    base = re.sub(r"\bconst\b", "", arg.replace("*", " ")).split()
    ctype = base[0]
    if stars == 0:
        return ("val", C_SCALARS[ctype])
    if stars >= 2:
        return ("ptr", "ptr")
    return ("ptr", C_SCALARS.get(ctype, "opaque"))          # void / handle structs: opaque pointee


def parse_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    decls = {}
    for ret, name, args in re.findall(r"\b(int|const char\*)\s+(qil_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = " ".join(args.split())
        kinds = [] if args in ("", "void") else [_c_arg_kind(a) for a in args.split(",")]
        decls[name] = ("i32" if ret == "int" else "cstring", kinds)
    return decls


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _jl_arg_kind(t):
    t = t.strip()
    if t in JL_SCALARS:
        return ("val", JL_SCALARS[t])
    m = re.fullmatch(r"(Ptr|Ref)\{(.*)\}", t)
    assert m, f"unrecognised Julia ccall argument type {t!r}"
    inner = m.group(2).strip()
    if inner == "Cvoid":
        return ("ptr", "opaque")
    if inner.startswith("Ptr{"):
        return ("ptr", "ptr")
    return ("ptr", JL_SCALARS[inner])


def _balanced(src, start):
    """src[start] == '(' -> index one past its matching ')'."""
    depth = 0
    for i in range(start, len(src)):
        if src[i] == "(":
            depth += 1
        elif src[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses in the shim")


def parse_shim():
    src = re.sub(r"#[^\n]*", "", open(SHIM).read())
    calls = []
    for m in re.finditer(r"ccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end():end - 1])
        sym = re.fullmatch(r"\(:(\w+),\s*LIB\)", parts[0])
        assert sym, f"ccall without a literal (:symbol, LIB): {parts[0]!r}"
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), types
        tlist = _split_top(types[1:-1])
        calls.append((sym.group(1), parts[1].strip(), tlist, len(parts) - 3))
    return calls


def test_every_ccall_matches_its_c_declaration():
    decls = parse_header()
    calls = parse_shim()
    assert len(calls) >= 30
    problems = []
    for name, ret, tlist, nargs in calls:
        if name not in decls:
            problems.append(f"{name}: not declared in the header")
            continue
        cret, ckinds = decls[name]
        jret = {"Cint": "i32", "Cstring": "cstring"}.get(ret)
        if jret != cret:
            problems.append(f"{name}: return type {ret} vs C {cret}")
        if len(tlist) != len(ckinds):
            problems.append(f"{name}: {len(tlist)} argument types vs {len(ckinds)} in C")
            continue
        if nargs != len(tlist):
            problems.append(f"{name}: {nargs} arguments passed for {len(tlist)} declared types")
        for pos, (jt, ck) in enumerate(zip(tlist, ckinds)):
            jk = _jl_arg_kind(jt)
            if jk[0] != ck[0]:
                problems.append(f"{name} arg {pos}: {jt} is {jk[0]}, C wants {ck[0]}")
            elif jk[0] == "val" and jk[1] != ck[1]:
                problems.append(f"{name} arg {pos}: {jt} ({jk[1]}) vs C {ck[1]}")
            elif jk[0] == "ptr" and "opaque" not in (jk[1], ck[1]) and jk[1] != ck[1]:
                problems.append(f"{name} arg {pos}: {jt} points to {jk[1]}, C to {ck[1]}")
    assert not problems, "\n".join(problems)


def test_integration_table_names_only_bound_entry_points():
    bound = {c[0] for c in parse_shim()}
    decls = parse_header()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    # the binding table: rows "| Julia ... | `qil_x`, `qil_y` | replaces |"
    rows = [l for l in doc.splitlines() if l.startswith("|") and "qil_" in l and l.count("|") >= 4]
    assert rows
    named = set()
    for l in rows:
        named |= set(re.findall(r"`(qil_[a-z0-9_]+)`", l.split("|")[2]))
    assert named <= set(decls), named - set(decls)
    missing = sorted(n for n in named if n not in bound)
    assert not missing, f"INTEGRATION.md lists entry points the Julia shim does not bind: {missing}"


def test_shim_covers_the_reference_surface():
    """The reference's exported operator surface on this path (src/QILaplace.jl:21-82) has a device method."""
    bound = {c[0] for c in parse_shim()}
    for must in ("qil_apply", "qil_apply_mpo_mpo", "qil_coefficient_batch", "qil_compress", "qil_canonicalize", "qil_norm",
                 "qil_mps_to_vector", "qil_signal_mps", "qil_signal_ztmps", "qil_rsvd", "qil_svd_trunc",
                 "qil_mps_download_site", "qil_mpo_download_site", "qil_mpo_bond_dims", "qil_build_dt_mpo_batch"):
        assert must in bound, must
