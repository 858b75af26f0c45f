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
                        ("patches/halo2_proofs-0.2/src/poly/domain.rs", "halo2_proofs 0.2.0")):
        src = open(os.path.join(ROOT, "rust", rel)).read()
        assert needle in src and "NOT COMPILED" in src, rel
