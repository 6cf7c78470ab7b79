"""hip_mdd/ is the Rust shim a maintainer would add, laid out as a crate (no Rust toolchain in this image, so it is never
compiled here; INTEGRATION.md documents it).  This keeps hip_mdd/src/lib.rs mechanically consistent with include/ddo_hip.h: every `#[repr(C)]` struct has the header's
fields in the header's order with compatible types, every `extern "C"` prototype names a function the header declares
with the same number of parameters, and every `pub const DDO_*` equals the header's `#define`."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Rust struct -> C struct
STRUCTS = {"DdoDecision": "ddo_decision", "DdoSubProblem": "ddo_subproblem", "DdoCompileInput": "ddo_compile_input",
           "DdoCompletion": "ddo_completion", "DdoCutsetRows": "ddo_cutset_rows"}
# Rust type -> the C types it may stand for (pointers compare by constness only)
SCALARS = {"c_int": {"int"}, "i64": {"int64_t"}, "u64": {"uint64_t"}, "usize": {"size_t"}, "i32": {"int32_t", "int"}, "f64": {"double"}}


def _header():
    src = open(os.path.join(ROOT, "include", "ddo_hip.h")).read()
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def _rust():
    src = open(os.path.join(ROOT, "hip_mdd", "src", "lib.rs")).read()
    assert "impl DecisionDiagram for HipMdd" in src and "impl Cache for HipCache" in src, "hip_mdd/src/lib.rs lost the shim"
    return re.sub(r"//[^\n]*", "", src)


def _c_struct_fields(hdr, name):
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), hdr, flags=re.S)
    assert m, f"struct {name} not found in ddo_hip.h"
    out = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        fm = re.match(r"(.*?)(\w+)$", decl)
        ctype, fname = fm.group(1).strip(), fm.group(2)
        out.append((fname, ctype))
    return out


def _rust_struct_fields(rs, name):
    m = re.search(r"pub\s+struct\s+%s\s*\{(.*?)\}" % name, rs, flags=re.S)
    assert m, f"struct {name} not found in INTEGRATION.md"
    out = []
    for decl in m.group(1).split(","):
        decl = " ".join(decl.split())
        if not decl:
            continue
        fname, rtype = [x.strip() for x in decl.split(":", 1)]
        out.append((fname.replace("pub ", ""), rtype))
    return out


def _compatible(rtype, ctype):
    if rtype.startswith("*"):
        if "*" not in ctype:
            return False
        return rtype.startswith("*const") == ("const" in ctype.split("*")[0])
    if ctype in STRUCTS.values():
        return STRUCTS.get(rtype) == ctype
    return ctype in SCALARS.get(rtype, set())


def test_repr_c_structs_mirror_the_header_field_for_field():
    hdr, rs = _header(), _rust()
    for rname, cname in STRUCTS.items():
        cf, rf = _c_struct_fields(hdr, cname), _rust_struct_fields(rs, rname)
        assert [f for f, _ in rf] == [f for f, _ in cf], f"{rname} vs {cname}: fields {[f for f, _ in rf]} != {[f for f, _ in cf]}"
        for (fname, rtype), (_, ctype) in zip(rf, cf):
            assert _compatible(rtype, ctype), f"{rname}.{fname}: Rust `{rtype}` does not match C `{ctype}`"
    # the struct literal in HipMdd::compile initialises every field of DdoCompileInput
    lit = re.search(r"let\s+ci\s*=\s*DdoCompileInput\s*\{(.*?)\};", rs, flags=re.S)
    assert lit
    for fname, _ in _c_struct_fields(hdr, "ddo_compile_input"):
        assert re.search(r"\b%s\s*:" % fname, lit.group(1)), f"HipMdd::compile does not set DdoCompileInput.{fname}"


def test_extern_prototypes_name_header_functions_with_the_same_arity():
    hdr, rs = _header(), _rust()
    ext = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', rs, flags=re.S)
    assert ext
    protos = re.findall(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", ext.group(1), flags=re.S)
    assert len(protos) >= 15
    for name, params in protos:
        m = re.search(r"\b%s\s*\(([^;{]*?)\)\s*;" % name, hdr, flags=re.S)
        assert m, f"INTEGRATION.md binds {name}, which include/ddo_hip.h does not declare"
        cparams = m.group(1).strip()
        n_c = 0 if cparams in ("", "void") else len(_split_params(cparams))
        n_r = len(_split_params(params)) if params.strip() else 0
        assert n_c == n_r, f"{name}: {n_r} parameters in the shim, {n_c} in the header"


def _split_params(p):
    out, depth, cur = [], 0, ""
    for ch in p:
        if ch in "(<":
            depth += 1
        elif ch in ")>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_constants_equal_the_header_defines():
    hdr, rs = _header(), _rust()
    consts = re.findall(r"pub\s+const\s+(DDO_\w+)\s*:\s*c_int\s*=\s*([^;]+);", rs)
    assert consts
    for name, val in consts:
        m = re.search(r"#define\s+%s\s+(\S+)" % name, hdr)
        assert m, f"{name} is not defined in ddo_hip.h"
        assert int(val.strip(), 0) == int(m.group(1).strip("()"), 0), f"{name}: shim {val} != header {m.group(1)}"
    # return codes the shim matches on
    assert re.search(r"#define\s+DDO_CUTOFF\s+2\b", hdr) and "2 => Err(Reason::CutoffOccurred)" in rs


def test_the_crate_is_laid_out_and_its_optima_are_the_oracle_suite_s():
    """hip_mdd/ has what `cargo test` needs, and the optima its tests assert are the ones tests/test_oracle.py pins the oracle on
    (ddo/examples/misp/tests.rs:71-161) -- for instance files that exist under data/misp."""
    for f in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), os.path.join("tests", "misp.rs")):
        assert os.path.exists(os.path.join(ROOT, "hip_mdd", f)), f
    cargo = open(os.path.join(ROOT, "hip_mdd", "Cargo.toml")).read()
    assert re.search(r'^ddo\s*=', cargo, flags=re.M) and re.search(r'^bit-set\s*=', cargo, flags=re.M)
    assert "rustc-link-lib=dylib=ddo_hip" in open(os.path.join(ROOT, "hip_mdd", "build.rs")).read()
    from tests.test_oracle import MISP_KATS
    known = dict(MISP_KATS)
    known.update({"keller4": 11, "brock200_3": 15, "brock200_4": 17, "hamming8-4": 16})   # test_oracle.py's slow cases, test_gpu_parity.py
    pairs = re.findall(r'"([\w\-]+)\.clq"\s*=>\s*(\d+)', open(os.path.join(ROOT, "hip_mdd", "tests", "misp.rs")).read())
    assert len(pairs) >= 15
    for name, value in pairs:
        assert os.path.exists(os.path.join(ROOT, "data", "misp", name + ".clq")), name
        assert known[name] == int(value), (name, value, known[name])
