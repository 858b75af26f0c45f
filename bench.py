#!/usr/bin/env python3
"""Headline bench: the GPU work-list of ONE proof of the synthetic 2^20-row halo2 PoE circuit (BASELINE.json configs[2],
the configuration `metric` is quoted on), inputs resident in HBM, through the C ABI (libzkcp_amd.so).

  step (default, --workload halo2): the device work-list of one halo2_proofs 0.2 create_proof over the column layout of the
      reference's ElGamalGadget (13 advice, 8 fixed, a lookup, equality over 16 columns, degree-9 gates;
      circuits-halo2/src/encryption.rs:83-161; SURVEY 8a a11 / 8d "Config 3"), in upstream's order -- see bench_halo2():
      advice commits + NTT chains, lookup and permutation grand products with their commits and chains, the quotient
      (expression evaluation on the extended coset, division by the vanishing polynomial, extended_to_coeff, 8 h-piece
      commits), the evaluations at x and its rotations with their multiopen folds, and the k-round inner-product argument.  NTT
      chains run on a second HIP stream beside the batched MSMs.  RNG, transcript and the lookup's sort stay on the CPU.
  metric    = constraints/sec = rows / wall-clock of the timed region (whole job)
  --workload column : BASELINE configs[1], one 2^20 MSM + one 2^20 NTT per step (the microbench; prints msm_mops)
  --workload groth16: the GPU work of one Groth16 proof at domain 2^logn (SURVEY 8d "Config 4")
  N > 1     = every MSM of 2^17 points or more is window-range sharded over the N ranks (one process per GPU) and combined with
              one all_gather of Jacobian points over RCCL (contangle-zkcp_amd/dist.py), the IPA's generator collapse by output
              range; NTT chains, the quotient expression and the small IPA rounds run on every rank.  Total work per step is
              fixed -> "scaling": "strong".

Launch: `python bench.py [--gpus 1 --steps K --warmup W]`, or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.
Inputs come from contangle-zkcp_amd/synth.py; only the cpu_baseline leg touches oracle/.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
NCOL = 13   # advice columns of the reference's halo2 circuit


KERNEL_SOURCES = ("zk_field.h", "zk_field29.h", "zk_mul_asm.h", "zk_params.h", "zk_params29.h", "zk_curve.h", "zk_curve29.h",
                  "zk_msm_kernels.h", "zk_ntt_kernels.h", "zk_ntt29_kernels.h", "zk_msm.inl", "zk_ntt.inl")


def kernel_src_sha16():
    """identity of the device code (kernels, field / curve arithmetic, launch plans): profiles/traffic.json is only
    believed for the kernels it was measured on; host-side files (the ABI, codecs) do not enter"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "contangle-zkcp_amd", "csrc")
    for f in KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` in `workload` from the committed PMC summary, or None when it is missing or was
    measured on other kernel sources"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None, None
    if tj.get("kernel_src_sha16") != kernel_src_sha16():
        return None, None
    e = tj.get("workloads", {}).get(workload, {}).get(kernel)
    if not e:
        return None, None
    return e.get("hbm_bytes_per_launch"), e.get("valu_wave_insts_per_launch")


class Env:
    pass


def setup():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--curve", default="Vesta", choices=["Vesta", "Pallas", "Bn254G1", "Bls381G1", "Bn254G2", "Bls381G2"],
                    help="MSM curve of the column workload (G2: Groth16's b_g2_query shape); pairing family of the groth16 workload")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--realistic", action="store_true", help="0/1-heavy witness mix (SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="NTTs and MSMs on one stream, MSMs one at a time (no overlap)")
    ap.add_argument("--workload", default="halo2", choices=["halo2", "column", "groth16"])
    ap.add_argument("--ipa", default="collapse", choices=["collapse", "virtual", "fold"],
                    help="halo2 work-list, opening: 'virtual' runs every round's two MSMs over the resident SRS; 'collapse' (default) does "
                         "that for the first --ipa-collapse-after rounds, then materialises the surviving generators in one step "
                         "(zk_ipa_collapse_device) and continues over them; 'fold' collapses the generator vector every round as upstream "
                         "does (one scalar multiplication per surviving point)")
    ap.add_argument("--ipa-collapse-after", default="6", help="round count(s) after which the generators are materialised, e.g. 6 or 6,10")
    ap.add_argument("--precomputed", action="store_true",
                    help="single-GPU A/B: the MSMs use ONE bucket set over a table of window multiples of the bases (zk_bases_precompute, 16 x the "
                         "key in HBM) instead of one bucket set per window")
    ap.add_argument("--ntt-limbs", type=int, default=0, choices=[0, 32], help="0: lazy 29-bit limbs inside the NTT tiles (default); 32: saturated words (A/B)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    e = Env()
    e.args, e.np, e.torch, e.dist = args, np, torch, dist
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    e.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if e.world != args.gpus:
        if e.world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = e.world
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # ZK_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices; RCCL
    # itself refuses two ranks on one GPU).  The driver's multi-GPU runs use the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if e.world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import contangle_zkcp_amd as zk
    from contangle_zkcp_amd import dist as zkdist, synth
    e.zk, e.zkdist, e.synth = zk, zkdist, synth
    zk.load()
    zk.init(local_rank)
    if args.ntt_limbs:
        zk.ntt_configure(limb_bits=args.ntt_limbs)
    e.st = torch.cuda.current_stream().cuda_stream
    return e


def to_dev(e, arr):
    return e.torch.from_numpy(e.np.ascontiguousarray(arr).view(e.np.int64)).cuda()


def make_bases(e, curve, n, seed):
    """seeded SRS P_i = [k_i]G, generated on the GPU by the windowed fixed-base path and left resident"""
    ks = e.synth.scalars_for(curve, n, seed)
    d = e.torch.empty((n, 2 * e.zk.base_limbs(curve)), dtype=e.torch.int64, device="cuda")
    e.zk.fixed_base_msm_device(curve, to_dev(e, ks), d, n, stream=e.st)
    e.torch.cuda.synchronize()
    return e.zk.Bases(curve, device_tensor=d, n=n), d


def barrier(e):
    e.torch.cuda.synchronize()
    if e.world > 1:
        e.dist.barrier()
    e.torch.cuda.synchronize()


def timed(e, step):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
    for i in range(e.args.warmup):
        step(i, False)
    barrier(e)
    e.zk.msm_profile_totals(reset=True)
    e.zk.ntt_profile_enable(True)
    e.zk.ntt_profile_read()
    barrier(e)
    t0 = time.perf_counter()
    for i in range(e.args.steps):
        step(i, True)
    barrier(e)
    elapsed = time.perf_counter() - t0
    if e.world > 1:
        t = e.torch.tensor([elapsed], dtype=e.torch.float64, device="cuda" if e.dist.get_backend() == "nccl" else "cpu")
        e.dist.all_reduce(t, op=e.dist.ReduceOp.MAX)
        elapsed = float(t.item())
    e.msm_tot = e.zk.msm_profile_totals(reset=True)
    e.ntt_tot = e.zk.ntt_profile_read()
    e.zk.ntt_profile_enable(False)
    return elapsed


def rooflines(e, workload_key):
    """HBM roofline of the two hot kernels of the timed region, measured live with HIP events on the launch streams
    (library side: zk_msm_profile_totals / zk_ntt_profile_read): achieved = algorithmic bytes / kernel time."""
    out = {}
    m, t = e.msm_tot, e.ntt_tot
    if m["msms"]:
        ach = m["algorithmic_bytes"] / (m["accumulate_kernel_ms"] * 1e-3) / 1e9
        traffic, valu = load_traffic("msm_accumulate_kernel", workload_key)
        out["msm_accumulate_kernel"] = {
            "kernel": "msm_accumulate_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "launches": m["launches"], "msms": m["msms"], "avg_launch_us": m["accumulate_kernel_ms"] / m["launches"] * 1e3,
            "algorithmic_bytes_per_launch": m["algorithmic_bytes"] / m["launches"], "kernel_ms_total": m["accumulate_kernel_ms"],
            "valu_wave_insts_per_launch": valu,
            "note": "integer-VALU-bound by construction (SURVEY 8d): see int_mad_roofline / valu_issue_roofline"}
    if t["launches"]:
        ach = t["algorithmic_bytes"] / (t["kernel_ms"] * 1e-3) / 1e9
        traffic, valu = load_traffic("ntt_pass_kernel", workload_key)
        out["ntt_pass_kernel"] = {
            "kernel": "ntt_pass_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "launches": t["launches"], "avg_launch_us": t["kernel_ms"] / t["launches"] * 1e3,
            "algorithmic_bytes_per_launch": t["algorithmic_bytes"] / t["launches"], "kernel_ms_total": t["kernel_ms"],
            "transforms": t["transforms"], "valu_wave_insts_per_launch": valu,
            "note": "algorithmic bytes = 32 B x (elements read + written) per transform, spread over its 1-3 pass launches; integer-VALU-bound"}
    return out


def emit(e, line, roofs):
    """the dominant kernel (most kernel time in the timed region) is `roofline`; the other hot kernel rides along"""
    if roofs:
        order = sorted(roofs.values(), key=lambda r: -r["kernel_ms_total"])
        line["roofline"] = order[0]
        if len(order) > 1:
            line["roofline_second_kernel"] = order[1]
        acc = roofs.get("msm_accumulate_kernel")
        if acc:
            # second view: the 32x32->64 MADs a mixed add strictly needs vs the measured v_mad_u64_u32 peak
            # (tools/microbench: 33.7 T/s).  MADs per Montgomery product in the lazy limb form: L^2 for a*b plus L per
            # non-zero modulus limb for m*p.
            # an Fq2 product on lazy limbs is two double products sharing one reduction each (zk_field29.h): 2 x (2 L^2 + reduction)
            mads = {"Vesta": 135, "Pallas": 135, "Bn254G1": 162, "Bls381G1": 392, "Bn254G2": 4 * 81 + 2 * 81, "Bls381G2": 4 * 196 + 2 * 196}.get(
                line["config"].get("msm_curve", ""), None)
            if mads:
                adds = line["config"]["msm_points"] * line["config"]["msm_windows_done"]
                a = adds * 10 * mads / (acc["avg_launch_us"] * 1e-6) / 1e12
                line["int_mad_roofline"] = {"achieved_tmad_s": a, "peak_tmad_s": 33.7, "frac": a / 33.7, "mads_per_field_mul": mads,
                                            "note": "mixed adds x 10 field mul x MAD-equivalents per mul over accumulate-kernel time"}
            if acc.get("valu_wave_insts_per_launch"):
                peak = 256 * 4 * 2.4e9 / 4
                v = acc["valu_wave_insts_per_launch"] / (acc["avg_launch_us"] * 1e-6)
                line["valu_issue_roofline"] = {"achieved_winst_s": v, "peak_winst_s": peak, "frac": v / peak,
                                               "note": "SQ_INSTS_VALU per launch (profiles/traffic.json, same kernel sources) / launch time; peak = 1024 SIMDs x 2.4 GHz / 4"}
    print(json.dumps(line), flush=True)


def base_line(e, value, elapsed, workload, cfg):
    a = e.args
    return {"metric": "constraints/sec", "value": value, "unit": "constraints/s", "n_gpus": e.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32 limbs (256/384-bit Montgomery integers; MSM buckets and NTT tiles on 9 x 29 / 14 x 28-bit lazy limbs)",
            "data": "synthetic", "config": dict({"workload": workload, "parallelism": "msm-window-shard x%d + all_gather" % e.world
                                                  if e.world > 1 else "single-gpu"}, **cfg)}


# ---------------------------------------------------------------------------------------------------- halo2 (default)
N_FIXED, N_PERM_COLS, PERM_CHUNK, N_H_PIECES = 8, 16, 7, 8     # reference circuit: 8 fixed columns; equality over 16 columns; degree 9


def quotient_program(n_adv, n_fix):
    """A gate set in the style of the reference's circuit (circuits-halo2/src/encryption.rs:83-161: ECC-style multiplication
    gates, a Pow5 S-box with a rotation, boolean / range checks behind fixed selectors, the lookup and permutation argument
    constraints), folded with y: columns [0, n_adv) advice, [n_adv, n_adv + n_fix) fixed, then A', S', Z_lookup, Z_perm x3.
    consts: [y, 5, 1, beta, gamma]"""
    A = lambda i, r=0: ("col", i % n_adv, r)
    Fx = lambda i, r=0: ("col", n_adv + i % n_fix, r)
    X = n_adv + n_fix
    ap, sp, zl = ("col", X, 0), ("col", X + 1, 0), ("col", X + 2, 0)
    zp = [("col", X + 3 + c, 0) for c in range(3)]
    prog = []
    # every term after the first is folded in as  acc = acc * y + term
    # multiplication gates  q (a b - c), eight of them
    first = True
    for g in range(8):
        term = [Fx(g), A(g), A(g + 1), ("mul",), A(g + 2), ("sub",), ("mul",)]
        prog.extend(term if first else [("scale", 0)] + term + [("add",)])
        first = False
    # Pow5 with a rotation  q (a^5 + 5 - b(omega X)), three of them
    for g in range(3):
        a = A(3 * g + 1)
        prog.extend([("scale", 0), Fx(g + 3), a, a, ("mul",), a, ("mul",), a, ("mul",), a, ("mul",), ("const", 1), ("add",),
                     ("col", (3 * g + 2) % n_adv, 1), ("sub",), ("mul",), ("add",)])
    # boolean checks  q a (a - 1), four of them
    for g in range(4):
        a = A(g + 9)
        prog.extend([("scale", 0), Fx(g + 4), a, a, ("const", 2), ("sub",), ("mul",), ("mul",), ("add",)])
    # lookup: Z(omega X)(A' + beta)(S' + gamma) - Z(X)(A + beta)(S + gamma), and (A' - S')(A' - A'(omega^-1 X))
    prog.extend([("scale", 0), ("col", X + 2, 1), ap, ("const", 3), ("add",), ("mul",), sp, ("const", 4), ("add",), ("mul",),
                 zl, A(0), ("const", 3), ("add",), ("mul",), Fx(7), ("const", 4), ("add",), ("mul",), ("sub",), ("add",)])
    prog.extend([("scale", 0), ap, sp, ("sub",), ap, ("col", X, -1), ("sub",), ("mul",), ("add",)])
    # permutation, per chunk: Z(omega X) prod (v + beta s + gamma) - Z(X) prod (v + delta-term + gamma), two columns of each chunk written out
    for c in range(3):
        v0, v1 = A(2 * c), A(2 * c + 1)
        prog.extend([("scale", 0), ("col", X + 3 + c, 1), v0, Fx(c), ("scale", 3), ("add",), ("const", 4), ("add",), ("mul",),
                     v1, Fx(c + 1), ("scale", 3), ("add",), ("const", 4), ("add",), ("mul",),
                     zp[c], v0, ("const", 4), ("add",), ("mul",), v1, ("const", 3), ("add",), ("mul",), ("sub",), ("add",)])
    return prog


def bench_halo2(e):
    """The device work-list of one halo2_proofs 0.2 create_proof over a 2^k-row circuit with the reference circuit's column
    layout, in upstream's order; RNG (blinding rows / scalars), the transcript and the sort of the lookup's permuted columns
    stay on the CPU and are represented by pre-made inputs / fixed challenges:
      1 advice        13 x commit (batched MSM, Lagrange basis)   +  13 x (lagrange_to_coeff ; coeff_to_extended)
      2 lookup        commit A', S' ; product Z_L (grand product) ; commit Z_L ; 3 x (l2c ; c2e)
      3 permutation   3 chunks of <= 7 columns: product Z_P (each chunk continues the previous) ; 3 commits ; 3 x (l2c ; c2e)
      4 quotient      commit the random polynomial ; gate / lookup / permutation expressions on the extended coset (one stack
                      program, 268 ops of which 81 products, 27 columns) ; divide by the vanishing polynomial ; extended_to_coeff ; commit the 8 pieces of h
      5 opening       60 evaluations of resident polynomials at x, omega x, omega^-1 x and their multiopen folds; the inner-product argument on the combined
                      polynomial: k rounds of 2 MSMs + 2 inner products + 3 folds
    """
    a, zk, torch, np = e.args, e.zk, e.torch, e.np
    curve = a.curve if a.curve in ("Vesta", "Pallas") else "Vesta"
    sfield = e.synth.CURVE_SCALAR_FIELD[curve]
    k, ext = a.logn, a.logn + 3
    n, ne = 1 << k, 1 << (a.logn + 3)
    g_lagrange, d_pts = make_bases(e, curve, n, 0x5EED)
    g_coeff, d_pts_c = make_bases(e, curve, n, 0x5EEE)          # Params::g (coefficient basis): h pieces, random poly, IPA
    dom = zk.halo2.EvaluationDomain(sfield, 9, k)      # degree-9 gates -> extended_k = k + 3 (Orchard-style)
    assert dom.extended_k == ext
    rf = lambda seed, m=n: e.synth.rand_field(sfield, m, seed)
    cols_host = np.stack([rf(0xC0DE + c) for c in range(NCOL)])
    d_cols = to_dev(e, cols_host)                                   # [13, n, 4] advice, Lagrange form
    d_fixed_ext = [to_dev(e, rf(0xF1 + c, ne)) for c in range(N_FIXED)]          # fixed columns on the extended coset: key material
    d_sigma = [to_dev(e, rf(0x51 + c)) for c in range(N_PERM_COLS)]              # permutation polynomials (Lagrange): key material
    d_lookup = to_dev(e, np.stack([rf(0xA0), rf(0xA1), rf(0xA2), rf(0xA3)]))     # A, S and the permuted A', S' (the sort is upstream's CPU step)
    d_rand = to_dev(e, rf(0xBB))                                                  # the vanishing argument's random polynomial
    beta, gamma, delta, y = (rf(0xC1, 1)[0], rf(0xC2, 1)[0], rf(0xC3, 1)[0], rf(0xC4, 1)[0])
    us = [rf(0xD0 + j, 1)[0] for j in range(k)]                                   # IPA challenges
    consts = np.stack([y, zk.halo2._mont_limbs(5, dom._p), zk.halo2._mont_limbs(1, dom._p), beta, gamma])
    # every chain (a Lagrange column -> coefficients -> extended coset) is owned by one rank; MSMs are sharded over all
    # NTTs stay single-GPU (north_star): with N > 1 ranks every rank runs every chain and the quotient (replicated), the
    # MSMs are what is sharded
    chains = list(range(NCOL + 3 + 3))                              # advice 0..12, lookup A' S' Z_L, permutation Z_P x3
    mine = chains
    d_ext = {c: torch.empty((ne, 4), dtype=torch.int64, device="cuda") for c in chains}      # (all resident: the quotient reads all of them)
    d_zl = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    d_zp = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(3)]
    d_h = torch.empty((ne, 4), dtype=torch.int64, device="cuda")
    d_ipa = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
    # the polynomials create_proof evaluates at x and its rotations (advice / fixed / permutation / lookup / h / random: ~60 queries
    # at 3 points for this column layout); stand-ins in coefficient form, resident
    N_EVAL = (45, 10, 5)
    d_evalsrc = to_dev(e, e.synth.rand_field(sfield, n, 0xE7A1)).unsqueeze(0).repeat(max(N_EVAL), 1, 1).contiguous()
    x_points = [rf(0xE0 + j, 1)[0] for j in range(3)]
    d_open = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d_g = torch.empty_like(d_pts_c)
    prog = quotient_program(NCOL, N_FIXED)
    main = torch.cuda.current_stream()
    side = main if a.serial else torch.cuda.Stream()
    result = {}
    phase_ms = {"advice": 0.0, "lookup": 0.0, "permutation": 0.0, "quotient": 0.0, "opening": 0.0}

    def chain(c, src, stream):
        """Lagrange column -> coefficients -> extended coset (only on the rank that owns chain c)"""
        if c in mine:
            d_ext[c][:n].copy_(src, non_blocking=True)
            dom.lagrange_to_coeff(d_ext[c][:n], stream=stream)
            dom.coeff_to_extended(d_ext[c], stream=stream)

    collapse_at = {int(x) for x in str(a.ipa_collapse_after).split(",") if x.strip()}

    pre = a.precomputed and e.world == 1
    if pre:
        g_lagrange.precompute(a.window_bits)
        g_coeff.precompute(a.window_bits)

    def commit_batch(bases, cols):
        if pre:
            return zk.msm_batch(bases, cols, montgomery=True, window_bits=a.window_bits, stream=e.st, precomputed=True)
        if a.serial:
            return [e.zkdist.msm_sharded(bases, cols[c], montgomery=True, window_bits=a.window_bits, stream=e.st) for c in range(cols.shape[0])]
        return e.zkdist.msm_batch_sharded(bases, cols, montgomery=True, window_bits=a.window_bits, stream=e.st)

    def step(i, timed_):
        t0 = time.perf_counter()
        # ---- 1 advice: NTT chains on the side stream beside the batched commitments
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for c in range(NCOL):
                chain(c, d_cols[c], side.cuda_stream)
        result["commitments"] = commit_batch(g_lagrange, d_cols)
        main.wait_stream(side)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # ---- 2 lookup
        commit_batch(g_lagrange, d_lookup[2:4])
        zk.halo2.lookup_product(sfield, d_lookup[0], d_lookup[1], d_lookup[2], d_lookup[3], beta, gamma, d_zl, stream=e.st)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            chain(NCOL, d_lookup[2], side.cuda_stream)
            chain(NCOL + 1, d_lookup[3], side.cuda_stream)
            chain(NCOL + 2, d_zl, side.cuda_stream)
        e.zkdist.msm_sharded(g_lagrange, d_zl, montgomery=True, window_bits=a.window_bits, stream=e.st)
        main.wait_stream(side)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # ---- 3 permutation: chunk c + 1 continues from the last value of chunk c
        z_first = None
        pcols = [d_cols[c % NCOL] for c in range(N_PERM_COLS)]
        for c in range(3):
            lo, hi = c * PERM_CHUNK, min(N_PERM_COLS, (c + 1) * PERM_CHUNK)
            z_first = zk.halo2.permutation_product(sfield, pcols[lo:hi], d_sigma[lo:hi], beta, gamma, delta, k, d_zp[c], first_column_index=lo,
                                                   z_first=z_first, stream=e.st)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for c in range(3):
                chain(NCOL + 3 + c, d_zp[c], side.cuda_stream)
        commit_batch(g_lagrange, torch.stack(d_zp))
        main.wait_stream(side)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        # ---- 4 quotient
        e.zkdist.msm_sharded(g_coeff, d_rand, montgomery=True, window_bits=a.window_bits, stream=e.st)
        ext_cols = [d_ext[c] for c in range(NCOL)] + d_fixed_ext + [d_ext[NCOL + c] for c in range(6)]
        zk.halo2.evaluate_expression(sfield, prog, ext_cols, consts, ext, 1 << (ext - k), d_h, stream=e.st)
        dom.divide_by_vanishing_poly(d_h, stream=e.st)
        dom.extended_to_coeff(d_h, stream=e.st)
        commit_batch(g_coeff, d_h.view(N_H_PIECES, n, 4))
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        # ---- 5 opening: the evaluations at x, omega x, omega^-1 x (one batched launch per point), then the inner-product argument
        # (the combined polynomial stands in: one of the coefficient vectors)
        for cnt, xp in zip(N_EVAL, x_points):
            zk.halo2.eval_polynomials(sfield, d_evalsrc[:cnt], xp, stream=e.st)
        for cnt in N_EVAL:                       # multiopen: fold each point set's polynomials with powers of x_1 (Horner)
            for q in range(1, cnt):
                zk.halo2.vec_muladd(sfield, d_open, d_evalsrc[q], x_points[0], stream=e.st)
        d_ipa[0].copy_(d_h[:n])
        d_ipa[1].copy_(d_ext[0][:n])
        if a.ipa == "fold":       # upstream's literal structure: collapse the generators every round
            d_g.copy_(d_pts_c)
            ipa = zk.halo2.IpaProver(curve, d_ipa[0], d_ipa[1], d_g, stream=e.st)
        else:                     # L, R over the resident SRS with challenge-weighted scalars: no generator is ever folded
            ipa = zk.halo2.IpaProverVirtual(curve, d_ipa[0], d_ipa[1], g_coeff, lambda shape: torch.zeros(shape, dtype=torch.int64, device="cuda"),
                                            stream=e.st)
        for j in range(k):
            ipa.round(sharded=e.world > 1)
            ipa.fold(us[j])
            if a.ipa == "collapse" and (j + 1) in collapse_at and j + 1 < k:
                ipa.collapse(sharded=e.world > 1)
        ipa.free()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        if timed_:
            for name, dt in zip(phase_ms, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                phase_ms[name] += dt * 1e3

    elapsed = timed(e, step)
    if e.rank == 0:
        prof = zk.msm_last_profile()
        m = e.msm_tot
        n_msm = NCOL + 3 + 3 + 1 + N_H_PIECES
        line = base_line(e, n * a.steps / elapsed, elapsed,
                         "halo2 create_proof device work-list, 2^%d rows, the reference circuit's column layout (BASELINE configs[2]): advice 13 x (commit + l2c + c2e) ; "
                         "lookup 3 commits + product + 3 NTT chains ; permutation 3 products + 3 commits + 3 NTT chains ; quotient: random-poly commit, "
                         "%d-op expression over 27 extended columns, divide by Z_H, extended_to_coeff, 8 h-piece commits ; opening: 60 evaluations at 3 points + the multiopen folds, %d-round IPA "
                         "(2 MSMs + 2 inner products + 3 folds per round; generators %s)" % (
                             k, len(prog), k, {"fold": "folded every round", "virtual": "never folded: every MSM over the SRS",
                                               "collapse": "materialised after round(s) %s by zk_ipa_collapse_device" % a.ipa_collapse_after}[a.ipa]),
                         {"rows_per_step": n, "msm_curve": curve, "msm_points": n, "msm_windows_done": 16, "msm_windows": 16,
                          "window_bits": 16, "columns": NCOL, "full_size_msms_per_step": n_msm, "ntt_2p%d_per_step" % k: 19, "ntt_2p%d_per_step" % ext: 20,
                          "streams": "one (serial)" if a.serial else "MSM batches on two library streams + NTT chains on a third",
                          "not_in_list": "RNG, transcript, the lookup argument's sort (CPU)"})
        line["phases_ms"] = {kk: v / a.steps for kk, v in phase_ms.items()}
        line["msm_ms_mean_all_sizes"] = m["device_ms"] / max(1, m["msms"])
        line["msms_per_step"] = m["msms"] / a.steps
        line["ntt_kernel_ms_per_step"] = e.ntt_tot["kernel_ms"] / a.steps
        line["accumulate_kernel_ms_per_step"] = m["accumulate_kernel_ms"] / a.steps
        if not a.no_cpu_baseline and e.world == 1:
            line["cpu_baseline"] = cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, cols_host[0], result["commitments"][0], sum(1 for o in prog if o[0] in ("mul", "scale")))
        roofs = rooflines(e, "halo2_2p%d" % k)
        if "msm_accumulate_kernel" in roofs:     # the IPA's small MSMs are in the totals: per-launch figures are means over all sizes
            roofs["msm_accumulate_kernel"]["note"] += "; launches include the 2 x %d shrinking MSMs of the IPA (means over all sizes)" % k
        line["config"].pop("msm_curve")
        emit(e, line, roofs)


def cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, col0, gpu_commit0, n_expr_muls):
    """The oracle ('port': CPU restatements of halo2_proofs 0.2 best_multiexp -- chunk per thread over ALL host cores -- ,
    best_fft, and per-element field / curve arithmetic) timed on this box on a bounded sample of the same work-list:
    ONE commitment, ONE iNTT 2^k, ONE NTT 2^(k+3), 2^16 field products (grand products / expression), 2^12 point
    multiplications (the IPA generator fold); the step is assembled from the counts of each kind, with perfect scaling
    over the cores assumed for the per-element work.  ark-ec's window-parallel Pippenger is timed beside the chunked
    one and the FASTER MSM is used.  Also a last bit-exact check of the GPU's first commitment."""
    from oracle import zk_oracle as orc
    np, zk = e.np, e.zk
    cores = os.cpu_count() or 1
    n = 1 << k
    pts = d_pts.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    exp = orc.msm_halo2(curve, pts, col0, threads=cores)
    t_h2 = time.perf_counter() - t0
    canon = orc.from_mont(sfield, col0)
    ark_threads = min(cores, -(-255 // orc.ark_window_bits(n)))
    t0 = time.perf_counter()
    exp_ark = orc.msm_ark(curve, pts, canon, threads=ark_threads)
    t_ark = time.perf_counter() - t0
    w, w_ext = zk.root_of_unity(sfield, k), zk.root_of_unity(sfield, ext)
    t0 = time.perf_counter()
    orc.halo2_best_fft(sfield, col0, w, k, threads=cores)
    t_n = time.perf_counter() - t0
    big = e.synth.rand_field(sfield, 1 << ext, 0xF00D)
    t0 = time.perf_counter()
    orc.halo2_best_fft(sfield, big, w_ext, ext, threads=cores)
    t_e = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.to_mont(sfield, col0[: 1 << 16])                       # one Montgomery product per element, one thread
    t_mul = (time.perf_counter() - t0) / (1 << 16)
    ks = e.synth.scalars_for(curve, 1 << 12, 77)
    t0 = time.perf_counter()
    orc.fixed_base_mul(curve, ks, threads=cores)               # 255-bit double-and-add per point, all cores
    t_pmul = (time.perf_counter() - t0) / (1 << 12)
    t_msm = min(t_h2, t_ark)
    n_msm, n_ntt_k, n_ntt_e = NCOL + 3 + 3 + 1 + N_H_PIECES, 19, 20
    muls_products = n * (4 * 8 + 3 * (4 * PERM_CHUNK + 8))      # factors, batched inversion, scan: lookup + three permutation chunks
    muls_expr = (1 << ext) * n_expr_muls                       # the products of the quotient program, at every row of the extended domain
    muls_eval = 2 * 60 * n                                      # evaluations (Horner) + multiopen folds: one product per coefficient each
    t_field = (muls_products + muls_expr + muls_eval) * t_mul / cores
    # IPA: MSMs of 2 * (n/2 + n/4 + ...) = 2n points ~ two full MSMs; n point multiplications for the generator folds
    t_ipa = 2 * t_msm + n * t_pmul
    t_step = n_msm * t_msm + n_ntt_k * t_n + n_ntt_e * t_e + t_field + t_ipa
    ok = bool((zk.point_to_affine(curve, gpu_commit0) == exp).all() and (exp == exp_ark).all())
    return {"value": n / t_step, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "best_multiexp 2^%d (halo2 chunk-per-thread, %d threads: %.3f s; ark window-parallel, %d threads: %.3f s; faster one used) ; best_fft 2^%d %.3f s, "
                      "2^%d %.3f s (%d threads) ; field product %.0f ns (one thread) ; point multiplication %.1f us per point on %d threads ; "
                      "step = %d MSMs + %d + %d FFTs + %.2g products / %d cores + IPA (2 MSM + 2^%d point multiplications) = %.2f s"
                      % (k, cores, t_h2, ark_threads, t_ark, k, t_n, ext, t_e, cores, t_mul * 1e9, t_pmul * 1e6, cores, n_msm, n_ntt_k, n_ntt_e,
                         muls_products + muls_expr, cores, k, t_step),
            "msm_mops": n / t_msm / 1e6, "msm_s": {"halo2_chunked": t_h2, "ark_window_parallel": t_ark}, "gpu_result_matches": ok}


# ---------------------------------------------------------------------------------------------------- column (configs[1])
def bench_column(e):
    a, zk, torch = e.args, e.zk, e.torch
    curve = a.curve
    sfield = e.synth.CURVE_SCALAR_FIELD[curve]
    n = 1 << a.logn
    bases, d_pts = make_bases(e, curve, n, 0x5EED)
    sc_host = e.synth.scalars_for(curve, n, 0xC0DE, realistic=a.realistic)
    d_sc = to_dev(e, sc_host)
    a_host = e.synth.rand_field(sfield, n, 0xF00D)
    d_a = to_dev(e, a_host)
    omega = zk.root_of_unity(sfield, a.logn)
    main = torch.cuda.current_stream()
    side = main if a.serial else torch.cuda.Stream()
    result = {}
    if a.precomputed and e.world == 1:
        t0 = time.perf_counter()
        bases.precompute(a.window_bits)
        sys.stderr.write("zk_bases_precompute: %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))

    def step(i, timed_):
        if i % e.world == e.rank:
            side.wait_stream(main)
            zk.ntt(sfield, d_a, omega, stream=side.cuda_stream)
        if a.precomputed and e.world == 1:
            result["msm"] = zk.msm(bases, d_sc, window_bits=a.window_bits, stream=e.st, precomputed=True)
        else:
            result["msm"] = e.zkdist.msm_sharded(bases, d_sc, window_bits=a.window_bits, stream=e.st)
        main.wait_stream(side)

    elapsed = timed(e, step)
    if e.rank == 0:
        prof = zk.msm_last_profile()
        m = e.msm_tot
        line = base_line(e, n * a.steps / elapsed, elapsed,
                         "2^%d-point %s MSM + 2^%d %s NTT per step (BASELINE configs[1])" % (a.logn, curve, a.logn, sfield),
                         {"rows_per_step": n, "scalars": "realistic-0/1-mix" if a.realistic else "uniform", "msm_curve": curve, "msm_points": n,
                          "msm_windows_done": prof["windows_done"], "msm_windows": prof["windows_total"], "window_bits": prof["window_bits"],
                          "ntt_stream": "same as MSM" if a.serial else "second HIP stream, overlapped with the MSM"})
        line["msm_ms"] = m["device_ms"] / max(1, m["msms"]) + m["host_tail_ms"] / max(1, m["msms"])
        line["msm_mops"] = n / (line["msm_ms"] * 1e-3) / 1e6 * (prof["windows_total"] / max(1, prof["windows_done"])) if e.world == 1 \
            else n / (elapsed / a.steps) / 1e6
        line["ntt_ms"] = e.ntt_tot["kernel_ms"] / max(1, e.ntt_tot["transforms"])
        line["msm_phases_ms"] = {kk: m[kk] / max(1, m["msms"]) for kk in ("sort_ms", "accumulate_kernel_ms", "accumulate_ms", "reduce_ms", "host_tail_ms")}
        if not a.no_cpu_baseline and e.world == 1:
            line["cpu_baseline"] = cpu_baseline_column(e, curve, sfield, a.logn, d_pts, sc_host, a_host, omega, result["msm"])
        emit(e, line, rooflines(e, "column_%s_2p%d" % (curve, a.logn)))


def cpu_baseline_column(e, curve, sfield, logn, d_pts, sc_host, a_host, omega, gpu_msm):
    """one full step on the host cores: the faster of the two reference MSM algorithms + best_fft, all cores"""
    from oracle import zk_oracle as orc
    np, zk = e.np, e.zk
    cores = os.cpu_count() or 1
    n = 1 << logn
    pts = d_pts.cpu().numpy().view(np.uint64)
    ark_threads = min(cores, -(-255 // orc.ark_window_bits(n)))
    t0 = time.perf_counter()
    exp = orc.msm_ark(curve, pts, sc_host, threads=ark_threads)
    t_ark = time.perf_counter() - t0
    mont = orc.to_mont(sfield, sc_host)
    t0 = time.perf_counter()
    exp2 = orc.msm_halo2(curve, pts, mont, threads=cores)
    t_h2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.halo2_best_fft(sfield, a_host, omega, logn, threads=cores)
    t_ntt = time.perf_counter() - t0
    t_msm = min(t_ark, t_h2)
    ok = bool((zk.point_to_affine(curve, gpu_msm) == exp).all() and (exp == exp2).all())
    return {"value": n / (t_msm + t_ntt), "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "1 full step: 2^%d MSM (halo2 chunk-per-thread on %d threads %.3f s; ark window-parallel on %d threads %.3f s; faster one used) "
                      "+ 2^%d best_fft on %d threads %.3f s" % (logn, cores, t_h2, ark_threads, t_ark, logn, cores, t_ntt),
            "msm_mops": n / t_msm / 1e6, "msm_s": {"halo2_chunked": t_h2, "ark_window_parallel": t_ark}, "gpu_result_matches": ok}


# ---------------------------------------------------------------------------------------------------- groth16 (configs[3])
def bench_groth16(e):
    """GPU work of ONE Groth16 proof (ark-groth16 0.3 create_proof, SURVEY 3.6 / 8d 'Config 4') at domain size m = 2^logn:
    R1CStoQAP::witness_map (3 iFFT + 3 coset FFT + pointwise + 1 coset iFFT, all resident) -> h; then the five MSMs
    h_query.h (m - 1, scalars straight from the NTT output, Montgomery form), a_query.z, b_g1_query.z (m), l_query.aux
    (0.75 m) on G1 and b_g2_query.z (m) on G2, submitted back to back and collected afterwards (deferred results).
    Synthetic SRS: seeded points, one array per query vector; witness z with the 0/1-heavy mix of a real assignment."""
    a, zk, torch, np = e.args, e.zk, e.torch, e.np
    fam = "Bn254" if a.curve.startswith("Bn254") else "Bls381"
    g1, g2, fr = fam + "G1", fam + "G2", fam + "Fr"
    m = 1 << a.logn
    n_l = (3 * m) // 4
    t_setup = time.perf_counter()
    (a_query, _), (b_g1_query, d_b1) = make_bases(e, g1, m, 0xA11), make_bases(e, g1, m, 0xB11)
    (h_query, _), (l_query, _) = make_bases(e, g1, m - 1, 0xC11), make_bases(e, g1, n_l, 0xD11)
    b_g2_query, d_b2 = make_bases(e, g2, m, 0xE11)
    t_setup = time.perf_counter() - t_setup
    z_host = e.synth.scalars_for(g1, m, 0xC0DE, realistic=True)
    z = to_dev(e, z_host)
    z_aux = z[:n_l].contiguous()
    abc_host = [e.synth.rand_field(fr, m, 0xF00D + i) for i in range(3)]
    d_abc0 = [to_dev(e, x) for x in abc_host]
    d_abc = [torch.empty_like(x) for x in d_abc0]
    phases = {"witness_map_ms": 0.0}
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    result = {}

    def step(i, timed_):
        for dst, src in zip(d_abc, d_abc0):
            dst.copy_(src)
        if e.rank == 0:
            evs[0].record()
            zk.groth16_witness_map(fr, d_abc[0], d_abc[1], d_abc[2], stream=e.st)
            evs[1].record()
        if e.world > 1:
            torch.cuda.synchronize()
            e.dist.broadcast(d_abc[0], src=0)
        jobs = ((h_query, d_abc[0][:m - 1], True), (a_query, z, False), (b_g1_query, z, False), (l_query, z_aux, False), (b_g2_query, z, False))
        result["pts"] = e.zkdist.msm_many_sharded(jobs, window_bits=a.window_bits, stream=e.st)
        if timed_ and e.rank == 0:
            torch.cuda.synchronize()
            phases["witness_map_ms"] += evs[0].elapsed_time(evs[1])

    elapsed = timed(e, step)
    if e.rank == 0:
        mt = e.msm_tot
        line = base_line(e, m * a.steps / elapsed, elapsed,
                         "Groth16 prover GPU work, domain 2^%d over %s: witness_map (7 NTTs) + MSMs h (m-1), a, b_g1 (m), l (0.75 m) on G1 + b_g2 (m) on G2"
                         % (a.logn, fam), {"rows_per_step": m, "srs_setup_s": t_setup, "msm_curve": g1, "msm_points": m, "msm_windows_done": 16})
        line["phases_ms"] = {"witness_map_ms": phases["witness_map_ms"] / a.steps, "msm_device_ms_total": mt["device_ms"] / a.steps,
                             "msm_accumulate_kernel_ms_total": mt["accumulate_kernel_ms"] / a.steps}
        if not a.no_cpu_baseline and e.world == 1:
            line["cpu_baseline"] = cpu_baseline_groth16(e, g1, g2, fr, a.logn, d_b1, d_b2, z_host, abc_host)
        line["config"].pop("msm_curve")   # five MSMs of three shapes: no single int_mad figure
        emit(e, line, rooflines(e, "groth16_%s_2p%d" % (fam, a.logn)))


def cpu_baseline_groth16(e, g1, g2, fr, logn, d_b1, d_b2, z_host, abc_host):
    """ark-ec 0.3 Pippenger + ark-poly 0.3 FFT restatements on the host cores, bounded: one G1 MSM and one NTT at full size,
    the G2 MSM on a 2^18 prefix (scaled linearly); step = 7 NTTs + 3.75 G1 MSMs + 1 G2 MSM"""
    from oracle import zk_oracle as orc
    np = e.np
    cores = os.cpu_count() or 1
    m = 1 << logn
    thr = min(cores, -(-255 // orc.ark_window_bits(m)))
    pts1 = d_b1.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    orc.msm_ark(g1, pts1, z_host, threads=thr)
    t_g1 = time.perf_counter() - t0
    ms = min(m, 1 << 18)
    pts2 = d_b2[:ms].cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    orc.msm_ark(g2, pts2, z_host[:ms], threads=min(cores, -(-255 // orc.ark_window_bits(ms))))
    t_g2 = (time.perf_counter() - t0) * (m / ms)
    t0 = time.perf_counter()
    orc.ark_fft(fr, abc_host[0], "coset_fft", threads=cores)
    t_ntt = time.perf_counter() - t0
    t_step = 7 * t_ntt + 3.75 * t_g1 + t_g2
    return {"value": m / t_step, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "ark restatements: one G1 MSM 2^%d (%d window threads) %.2f s, G2 MSM on a 2^%d prefix scaled to 2^%d %.2f s, one coset FFT 2^%d (%d threads) %.2f s; "
                      "step = 7 FFT + 3.75 G1 MSM + 1 G2 MSM = %.2f s" % (logn, thr, t_g1, ms.bit_length() - 1, logn, t_g2, logn, cores, t_ntt, t_step)}


def main():
    e = setup()
    {"halo2": bench_halo2, "column": bench_column, "groth16": bench_groth16}[e.args.workload](e)
    if e.world > 1:
        e.dist.barrier()
        e.dist.destroy_process_group()


if __name__ == "__main__":
    main()
