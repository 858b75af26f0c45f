"""CPU: the Rust binding (rust/zkcp-amd-sys/src/lib.rs, written but not compiled here -- there is no Rust toolchain in the
image) declares exactly the functions the two C headers declare, with the same number of parameters; the repr(C) structs
have the headers' field counts.  Keeps the f1 files from drifting away from the ABI they bind."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_decls():
    out = {}
    for h in ("zkcp_amd.h", "zkcp_amd_prover.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(?:int|const char \*)\s*(zk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            args = m.group(2).strip()
            out[m.group(1)] = 0 if args in ("void", "") else args.count(",") + 1
    return out


def rust_decls():
    src = open(os.path.join(ROOT, "rust", "zkcp-amd-sys", "src", "lib.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (zk_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else args.count(":")
    return out


def test_extern_block_matches_headers():
    c, r = c_decls(), rust_decls()
    assert len(c) >= 50
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name] == r[name], (name, c[name], r[name])


C_TO_RUST = [   # C parameter type (normalised) -> the Rust FFI type that has the same size, signedness and pointer constness
    (r"const void \*const \*", "*const *const c_void"), (r"const void \*", "*const c_void"), (r"void \*", "*mut c_void"),
    (r"const uint8_t \*", "*const u8"), (r"uint8_t \*", "*mut u8"), (r"const uint64_t \*", "*const u64"), (r"uint64_t \*", "*mut u64"),
    (r"const uint32_t \*", "*const u32"), (r"const int \*", "*const c_int"), (r"char \*", "*mut c_char"),
    (r"const (zk_[a-z0-9_]+) \*", "*const \\1"), (r"(zk_[a-z0-9_]+) \*", "*mut \\1"),
    (r"uint64_t", "u64"), (r"uint32_t", "u32"), (r"int64_t", "i64"), (r"int", "c_int"),
    (r"zk_curve_t|zk_field_t|zk_pairing_t", "c_int"),
]


def _c_params():
    out = {}
    for h in ("zkcp_amd.h", "zkcp_amd_prover.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(?:int|const char \*)\s*(zk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            args = " ".join(m.group(2).split())
            params = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
            types = []
            for a in params:
                t = re.sub(r"\b[a-zA-Z_][a-zA-Z0-9_]*$", "", a).strip()      # drop the parameter name
                t = re.sub(r"\s*\*\s*", " *", t).replace("* *", "**").strip()
                types.append(re.sub(r"\s+", " ", t))
            out[m.group(1)] = types
    return out


def _rust_params():
    src = open(os.path.join(ROOT, "rust", "zkcp-amd-sys", "src", "lib.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (zk_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = " ".join(m.group(2).split())
        out[m.group(1)] = [a.split(":", 1)[1].strip() for a in args.split(",") if ":" in a]
    return out


def _expected_rust(ctype):
    t = ctype.replace("const void *const*", "const void *const *").replace(" **", " * *")
    for pat, rust in C_TO_RUST:
        m = re.fullmatch(pat, t)
        if m:
            return m.expand(rust) if "\\1" in rust else rust
    raise AssertionError("no Rust mapping for C type %r" % ctype)


def test_extern_block_parameter_types_match():
    """every parameter of every bound function: pointer vs integer, integer width, pointee type and constness -- a function body
    swap or a u32 / u64 mix-up in rust/zkcp-amd-sys/src/lib.rs fails here (counts alone would not notice)"""
    c, r = _c_params(), _rust_params()
    checked = 0
    for name, ctypes_ in c.items():
        assert name in r, name
        assert len(ctypes_) == len(r[name]), (name, ctypes_, r[name])
        for i, (ct, rt) in enumerate(zip(ctypes_, r[name])):
            assert _expected_rust(ct) == rt, (name, i, ct, rt, _expected_rust(ct))
            checked += 1
    assert checked > 400


def test_patch_files_call_only_bound_functions_with_the_right_arity():
    """every `zk::zk_*(...)` call in the fork files names a function of the extern block and passes as many arguments as it takes
    (a renamed or re-shaped entry point in the headers shows up here, not at a maintainer's first cargo build)"""
    r = rust_decls()
    seen = 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rust", "patches")):
        for fn in files:
            if not fn.endswith(".rs"):
                continue
            src = open(os.path.join(dirpath, fn)).read()
            src = re.sub(r"//[^\n]*", "", src)
            for m in re.finditer(r"zk::(zk_[a-z0-9_]+)\s*\(", src):
                name = m.group(1)
                assert name in r, (fn, name)
                depth, i, args, cur = 1, m.end(), 0, ""
                while depth:
                    ch = src[i]
                    if ch in "([{":
                        depth += 1
                    elif ch in ")]}":
                        depth -= 1
                    elif ch == "," and depth == 1:
                        args += 1
                        cur = ""
                        i += 1
                        continue
                    if depth:
                        cur += ch
                    i += 1
                if cur.strip():
                    args += 1
                assert args == r[name], (fn, name, args, r[name])
                seen += 1
    assert seen >= 25


def test_repr_c_structs_match():
    hdr = open(os.path.join(ROOT, "include", "zkcp_amd.h")).read() + open(os.path.join(ROOT, "include", "zkcp_amd_prover.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "zkcp-amd-sys", "src", "lib.rs")).read()
    for name in ("zk_msm_opts", "zk_ntt_opts", "zk_msm_profile", "zk_msm_totals", "zk_ntt_totals", "zk_ark_span", "zk_ark_pk_index",
                 "zk_groth16_assembly", "zk_expr_op"):
        body = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % name, hdr, flags=re.S).group(1)
        n_c = 0
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            decl = re.sub(r"^(const\s+)?(void|zk_ark_span|[a-z0-9_]+_t|int|float|double)\s*", "", decl)
            n_c += len([x for x in decl.split(",") if x.strip()])
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S).group(1)
        n_r = len(re.findall(r"pub [a-z0-9_]+:", rbody))
        assert n_c == n_r, (name, n_c, n_r)


def test_patch_files_cite_their_upstream_targets():
    for rel, needle in (("patches/ark-ec-0.3/src/msm/variable_base.rs", "ark-ec 0.3.0"),
                        ("patches/ark-poly-0.3/src/domain/radix2/fft.rs", "ark-poly 0.3.0"),
                        ("patches/ark-groth16-0.3/src/r1cs_to_qap.rs", "ark-groth16 0.3.0"),
                        ("patches/ark-groth16-0.3/src/prover.rs", "ark-groth16 0.3.0"),
                        ("patches/halo2_proofs-0.2/src/arithmetic.rs", "halo2_proofs 0.2.0"),
                        ("patches/halo2_proofs-0.2/src/poly/domain.rs", "halo2_proofs 0.2.0"),
                        ("patches/halo2_proofs-0.2/src/poly/multiopen/prover.rs", "halo2_proofs 0.2.0"),
                        ("patches/halo2_proofs-0.2/src/poly/commitment/prover.rs", "halo2_proofs 0.2.0"),
                        ("patches/halo2_proofs-0.2/src/plonk/vanishing/prover.rs", "halo2_proofs 0.2.0")):
        src = open(os.path.join(ROOT, "rust", rel)).read()
        assert needle in src and "NOT COMPILED" in src, rel
