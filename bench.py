#!/usr/bin/env python3
"""Headline bench: the GPU work-list of ONE proof of the synthetic 2^20-row halo2 PoE circuit (BASELINE.json configs[2],
the configuration `metric` is quoted on), inputs resident in HBM, through the C ABI (libzkcp_amd.so).

  step (default, --workload halo2): the device work-list of one halo2_proofs 0.2 create_proof over the column layout of the
      reference's ElGamalGadget (13 advice, 8 fixed, a lookup, equality over 16 columns, degree-9 gates;
      circuits-halo2/src/encryption.rs:83-161; SURVEY 8a a11 / 8d "Config 3"), in upstream's order -- see bench_halo2():
      advice commits + NTT chains, lookup and permutation grand products with their commits and chains, the quotient
      (expression evaluation on the extended coset -- in a kernel compiled for that expression on first use --, division by the
      vanishing polynomial, extended_to_coeff, 8 h-piece commits), the evaluations at x and its rotations with their multiopen folds, and the k-round inner-product argument.  NTT
      chains run on a second HIP stream beside the batched MSMs.  RNG, transcript and the lookup's sort stay on the CPU.
  metric    = constraints/sec = rows / wall-clock of the timed region (whole job)
  --workload column : BASELINE configs[1], one 2^20 MSM + one 2^20 NTT per step (the microbench; prints msm_mops)
  --workload groth16: the GPU work of one Groth16 proof at domain 2^logn (SURVEY 8d "Config 4")
  N > 1     = every MSM of 2^17 points or more is sharded over the N ranks (one process per GPU): a single MSM by scalar-window range,
              a batch with at least one vector per rank by vector; one all_gather of Jacobian points over RCCL either way
              (contangle-zkcp_amd/dist.py).  The quotient AND its commitments are sharded by sub-coset of the extended domain (8 sub-cosets
              dealt round-robin: expression, division, an inverse transform of size 2^20 and the MSM of the result per sub-coset; the
              ranks exchange 8 commitments and one folded vector each, never h), the IPA's generator collapse by output range.  The
              size-2^20 transforms, the grand products and the small IPA rounds run on every rank.  Total work per step is fixed ->
              "scaling": "strong".  `digest` in the line is the same for every N and every order of work.

Launch: `python bench.py [--gpus 1 --steps K --warmup W]`, or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.
Inputs come from contangle-zkcp_amd/synth.py; only the cpu_baseline leg touches oracle/.
"""
import argparse
import concurrent.futures
import gc
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


def host_cores():
    """CPUs this process may actually use: the scheduler affinity and the cgroup CPU quota of the box, not the host's core count
    (a one-GPU box of the pool shows 256 logical CPUs and grants about 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except (OSError, ValueError):
        pass
    return n


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
    ap.add_argument("--quotient-parts", type=int, default=1, choices=[1, 2, 4, 8],
                    help="halo2 work-list: the extended coset handled as this many sub-cosets -- expression, division and an inverse transform of "
                         "size 2^23 / parts per sub-coset, whose folded coefficients are committed while the next sub-coset is evaluated; the "
                         "pieces' commitments are combinations of those (commitments are linear).  1 (default on one GPU) = upstream's order: one "
                         "expression pass, extended_to_coeff, then the 8 commitments -- measured faster there (the GPU is issue-bound either way: "
                         "profiles/r03_j_*); N > 1 ranks always use 8, dealt round-robin: that is how the quotient and its commitments shard")
    ap.add_argument("--precomputed", action="store_true",
                    help="single-GPU A/B: the MSMs use ONE bucket set over a table of window multiples of the bases (zk_bases_precompute, 16 x the "
                         "key in HBM) instead of one bucket set per window")
    ap.add_argument("--ntt-limbs", type=int, default=0, choices=[0, 32], help="0: lazy 29-bit limbs inside the NTT tiles (default); 32: saturated words (A/B)")
    ap.add_argument("--expr-kernel", default="auto", choices=["auto", "never", "always"],
                    help="halo2 work-list, lazy quotient evaluator: auto = the kernel compiled for the expression (hiprtc, first call) for 2^16 rows "
                         "and more; never = the interpreter kernel (A/B)")
    ap.add_argument("--expr-limbs", type=int, default=0, choices=[0, 32],
                    help="halo2 work-list, quotient expression: 0 = lazy 29-bit limbs over cosets written in the R' radix (default); 32 = the saturated evaluator (A/B)")
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
    gc.collect()       # the set-up's garbage (host copies of the columns) is collected here, not at some point inside a timed step
    gc.freeze()
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
            "kernel": "ntt_pass_kernel" if e.args.ntt_limbs == 32 else "ntt_pass29_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "launches": t["launches"], "avg_launch_us": t["kernel_ms"] / t["launches"] * 1e3,
            "algorithmic_bytes_per_launch": t["algorithmic_bytes"] / t["launches"], "kernel_ms_total": t["kernel_ms"],
            "transforms": t["transforms"], "valu_wave_insts_per_launch": valu,
            "note": "algorithmic bytes = 32 B x (elements read + written) per transform, spread over its 1-3 pass launches; integer-VALU-bound"}
    return out


def emit(e, line, roofs):
    """the dominant kernel (most kernel time in the timed region) is `roofline`; the other hot kernel rides along"""
    for r in roofs.values():
        # both kernels are bound by vector-ALU issue, not by HBM: the second figure every roofline entry carries
        if r.get("valu_wave_insts_per_launch"):
            peak = 256 * 4 * 2.4e9 / 4
            v = r["valu_wave_insts_per_launch"] / (r["avg_launch_us"] * 1e-6)
            r["valu_issue"] = {"achieved_winst_s": v, "peak_winst_s": peak, "frac": v / peak,
                               "note": "SQ_INSTS_VALU per launch (profiles/traffic.json: same kernel sources) / mean launch time; peak = 1024 SIMDs x 2.4 GHz / 4 "
                                       "cycles per wave instruction.  Launches of the work-list share the CUs with kernels of other streams: run alone the "
                                       "NTT pass measures 0.40 ms for the 2^23 middle pass = 2.1e8 wave instructions (profiles/r03_h_*), the accumulate "
                                       "kernel 0.93 of this peak (column workload)"}
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
                adds = line["config"]["msm_points"] * line["config"]["msm_windows_done"] * acc["msms"] / max(1, acc["launches"])   # a launch sums up to 4 scalar vectors
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
# the reference circuit (circuits-halo2/src/encryption.rs:83-161, :143-148): 13 advice, 8 fixed, 3 instance columns, one lookup,
# equality over 16 columns (13 advice + 3 instance) in chunks of 7, degree-9 gates -> 8 quotient pieces
N_INST, N_FIXED, N_PERM_COLS, PERM_CHUNK, N_H_PIECES = 3, 8, 16, 7, 8


def bench_halo2(e):
    """The device work-list of one halo2_proofs 0.2 create_proof over a 2^k-row circuit with the reference circuit's column
    layout, in upstream's order; RNG (blinding rows / scalars, the random polynomials), the transcript and the sort of the
    lookup's permuted columns stay on the CPU and are represented by pre-made inputs / fixed challenges:
      0 instance      3 x commit (Lagrange basis) + 3 x (lagrange_to_coeff ; coeff_to_extended)
      1 advice        13 x commit (one batched MSM call with the instance columns) + 13 NTT chains
      2 lookup        commit A', S' ; 2 chains               [permute_expression_pair: CPU]
      3 permutation   3 chunks of <= 7 columns: product Z_P (each chunk continues the previous) ; 3 commits ; 3 chains
      4 lookup        product Z_L ; commit ; chain
      5 vanishing     commit the random polynomial
      6 quotient      the gate / lookup / permutation expressions on the extended coset (one stack program, 268 ops of which 81
                      products, 30 columns) ; divide by the vanishing polynomial ; extended_to_coeff ; commit the 8 pieces of h
      7 evaluations   h(X) = sum x^(n i) h_i ; 58 evaluations of resident coefficient vectors at x, omega x, omega^-1 x, omega^last x
      8 multiopen     4 point sets: fold each set's polynomials with x_1 (Horner) ; kate_division by every point of the set ; fold
                      the sets with x_2 -> q' ; commit q' ; evaluate the 4 set polynomials at x_3 ; fold q' and the sets with x_4 -> p
      9 opening       the inner-product argument on p: commit the blinding polynomial s ; p' = s xi + p ; p'(x_3) ; b = powers of
                      x_3 ; k rounds of 2 MSMs + 2 inner products + 3 folds
    A chain keeps BOTH forms of its column: the coefficients (out of place, for steps 7-8) and the extended coset.
    With --quotient-parts P > 1 (always 8 on N > 1 ranks) the quotient (step 6) runs sub-coset by sub-coset: each sub-coset's numerator is divided, brought to
    its folded coefficients by ONE transform of size 2^23 / parts, and committed slice by slice while the next sub-coset is being
    evaluated; the 8 pieces' commitments are fixed linear combinations of the slices' commitments (EvaluationDomain.piece_scalars) and
    the folded h(X) of step 7 a linear combination of the slices -- the same group elements / field elements as upstream's order
    (tests: check_quotient_by_parts), which --quotient-parts 1 runs for comparison.
    N > 1 ranks (one process per GPU): every column MSM of 2^17 points or more is window-range sharded (batches by vector); the 8
    sub-cosets are dealt round-robin, so coset transforms, expression, division, inverse transforms AND the quotient's MSMs are
    sharded with no exchange but one all_gather of the slices' commitments and one of a folded vector (32 B per row) per rank."""
    a, zk, torch, np = e.args, e.zk, e.torch, e.np
    H = zk.halo2
    curve = a.curve if a.curve in ("Vesta", "Pallas") else "Vesta"
    sfield = e.synth.CURVE_SCALAR_FIELD[curve]
    k, ext = a.logn, a.logn + 3
    n, ne = 1 << k, 1 << (a.logn + 3)
    g_lagrange, d_pts = make_bases(e, curve, n, 0x5EED)
    g_coeff, d_pts_c = make_bases(e, curve, n, 0x5EEE)          # Params::g (coefficient basis): h pieces, random polys, q', IPA
    dom = H.EvaluationDomain(sfield, 9, k)             # degree-9 gates -> extended_k = k + 3 (Orchard-style)
    assert dom.extended_k == ext
    # sub-cosets of the extended domain (dom.coeff_to_extended_part): QP in all, this rank's are my_parts (round-robin over the ranks)
    QP = 8 if e.world > 1 else a.quotient_parts
    my_parts = [j for j in range(QP) if j % e.world == e.rank]
    PL = len(my_parts)
    m, rsc = ne // QP, dom.rot_scale_part(QP)
    r_sl = m // n                                           # n-coefficient slices (= quotient pieces' worth) per sub-coset
    rf = lambda seed, cnt=n: e.synth.rand_field(sfield, cnt, seed)
    newbuf = lambda *shape: torch.empty(shape + (4,), dtype=torch.int64, device="cuda")
    # ---- inputs: Lagrange columns (instance 0..2, advice 3..15: one buffer, one batched commitment), lookup columns, challenges
    NLAG = N_INST + NCOL
    lag_host = np.stack([rf(0x1D0 + c) for c in range(N_INST)] + [rf(0xC0DE + c) for c in range(NCOL)])
    d_lag = to_dev(e, lag_host)
    d_lookup = to_dev(e, np.stack([rf(0xA0), rf(0xA1), rf(0xA2), rf(0xA3)]))     # A, S and the permuted A', S' (the sort is upstream's CPU step)
    d_sigma = [to_dev(e, rf(0x51 + c)) for c in range(N_PERM_COLS)]              # permutation polynomials (Lagrange): key material
    # fixed columns on this rank's sub-cosets: key material (one seeded extended vector per column, so that every split of the coset sees the same column)
    d_fixed_cos = []
    for c in range(N_FIXED):
        full = rf(0xF1 + c, ne)
        d_fixed_cos.append(to_dev(e, np.stack([full[j::QP] for j in my_parts])))
    del full
    beta, gamma, delta, y = (rf(0xC1, 1)[0], rf(0xC2, 1)[0], rf(0xC3, 1)[0], rf(0xC4, 1)[0])
    x, x1, x2, x3, x4, xi = (rf(0xE0 + j, 1)[0] for j in range(6))
    us = [rf(0xD0 + j, 1)[0] for j in range(k)]                                   # IPA challenges
    p_int = dom._p
    to_int = lambda v: sum(int(w) << (64 * i) for i, w in enumerate(v.tolist())) * pow(1 << 256, -1, p_int) % p_int
    mont = lambda v: H._mont_limbs(v % p_int, p_int)
    w_int, x_int = to_int(dom.omega), to_int(x)
    x_next, x_prev, x_last = mont(x_int * w_int), mont(x_int * pow(w_int, -1, p_int)), mont(x_int * pow(w_int, -6, p_int))   # omega^last: -(blinding + 1) rows
    xn = mont(pow(x_int, n, p_int))
    consts = np.stack([y, mont(5), mont(1), beta, gamma])
    # ---- the resident coefficient forms, one table [NP, n, 4]: rows grouped by point set, in the order S0 | S2 | S1 | S3 so that
    # every evaluation point reads ONE contiguous range.  S0 = {x}, S1 = {x, wx}, S2 = {x, w^-1 x}, S3 = {x, wx, w^last x}
    rot_adv = e.synth.rotated_advice(NCOL)
    names = ([("inst", c) for c in range(N_INST)] + [("adv", c) for c in range(NCOL) if c not in rot_adv] + [("fixed", c) for c in range(N_FIXED)]
             + [("random", 0), ("h", 0)] + [("sigma", c) for c in range(N_PERM_COLS)] + [("lk", "S'")])
    n_s0 = len(names)
    names += [("lk", "A'")]
    n_s2 = 1
    names += [("adv", c) for c in rot_adv] + [("lk", "Z"), ("zp", 2)]
    n_s1 = len(rot_adv) + 2
    names += [("zp", 0), ("zp", 1)]
    n_s3 = 2
    NP = len(names)
    row = {nm: i for i, nm in enumerate(names)}
    sets = [(0, n_s0, [x]), (n_s0 + n_s2, n_s1, [x, x_next]), (n_s0, n_s2, [x, x_prev]), (n_s0 + n_s2 + n_s1, n_s3, [x, x_next, x_last])]
    d_coef = newbuf(NP, n)
    for nm in names:                                 # key material / the prover's random polynomial: pre-made, resident
        if nm[0] in ("fixed", "sigma", "random"):
            d_coef[row[nm]].copy_(to_dev(e, rf(0xAB00 + row[nm])))
    d_spoly = to_dev(e, rf(0xBB5))                                                # the argument's blinding polynomial s (RNG: CPU)
    # ---- chains: every committed Lagrange column -> coefficients (kept) -> this rank's sub-coset of the extended domain
    chain_names = [("inst", c) for c in range(N_INST)] + [("adv", c) for c in range(NCOL)] + [("lk", "A'"), ("lk", "S'"), ("lk", "Z")] + [("zp", c) for c in range(3)]
    d_cos = {nm: newbuf(PL, m) for nm in chain_names}
    d_z = newbuf(4, n)                           # the grand products Z_P (3 chunks) and Z_L: one buffer, one batched commitment
    d_zp, d_zl = d_z[:3], d_z[3]
    d_hp = newbuf(PL, m)                       # the quotient's numerator on this rank's sub-cosets, then their folded coefficients
    hp_flat = d_hp.view(PL * r_sl, n, 4)       # ... as n-coefficient slices: what gets committed
    d_hsum = newbuf(e.world, n) if e.world > 1 else None
    d_q = newbuf(len(sets), n)           # the point sets' folded polynomials
    d_qprime, d_tmp = newbuf(n), newbuf(n)
    d_ipa = [newbuf(n) for _ in range(2)]
    d_S, d_W = torch.zeros((2, n, 4), dtype=torch.int64, device="cuda"), newbuf(n)    # IPA scalar / weight buffers, reused every step
    prog = e.synth.quotient_program(NCOL, N_FIXED, N_INST)
    lazy_expr = a.expr_limbs != 32
    H.expr_configure(a.expr_kernel)
    if lazy_expr:                                     # key material in the evaluator's radix, once
        for d in d_fixed_cos:
            H.to_lazy_form(sfield, d.view(PL * m, 4), stream=e.st)
    main = torch.cuda.current_stream()
    side = main if a.serial else torch.cuda.Stream()
    msm_stream = side       # the quotient's slice commitments, issued from a worker thread: the chains' stream is idle by then (one more
                            # stream would shift every stream's hardware queue: measured +6 ms per step with an unused extra stream)
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    ev_zp, ev_zl = torch.cuda.Event(), torch.cuda.Event()
    result = {}
    PH = ("advice", "lookup", "permutation", "quotient", "evaluations", "multiopen", "opening")
    phase_ms = dict.fromkeys(PH, 0.0)
    cls = {"commit": None, "ipa_full": None, "ipa_small": None}      # MSM totals by size class (read at the class boundaries)
    cls_sum = {kk: None for kk in cls}

    def chain(nm, src, stream):
        """Lagrange column -> coefficients (out of place, kept for the openings) -> this rank's sub-coset"""
        co = d_coef[row[nm]]
        dom.lagrange_to_coeff(src, stream=stream, out=co)
        if e.world == 1:      # every sub-coset is this rank's: ONE transform of the whole coset, stored sub-coset by sub-coset
            dom.coeff_to_extended(d_cos[nm].view(ne, 4), stream=stream, coeffs=co, lazy_out=lazy_expr, parts=QP)
        else:                 # a rank's own sub-cosets: one transform of size 2^23 / 8 each
            for jj, j in enumerate(my_parts):
                dom.coeff_to_extended_part(co, d_cos[nm][jj], j, QP, stream=stream, lazy_out=lazy_expr)

    collapse_at = {int(v) for v in str(a.ipa_collapse_after).split(",") if v.strip()}
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

    def commit(bases, col):
        return e.zkdist.msm_sharded(bases, col, montgomery=True, window_bits=a.window_bits, stream=e.st)

    def take_class(name):
        t = zk.msm_profile_totals(reset=True)
        if cls_sum[name] is None:
            cls_sum[name] = dict.fromkeys(t, 0)
        for kk, v in t.items():
            cls_sum[name][kk] += v

    def step(i, timed_):
        # Phase times are host timestamps taken where the host holds a phase's commitments (a transcript point); there is no
        # device-wide barrier between the phases: the NTT chains float on the side stream -- the next phase's MSMs do not need
        # them -- until the quotient reads the cosets.  (--serial: everything on one stream.)
        t0 = time.perf_counter()
        # ---- 0 + 1 instance and advice columns: NTT chains on the side stream beside one batched commitment call
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for c in range(N_INST):
                chain(("inst", c), d_lag[c], side.cuda_stream)
            for c in range(NCOL):
                chain(("adv", c), d_lag[N_INST + c], side.cuda_stream)
            chain(("lk", "A'"), d_lookup[2], side.cuda_stream)     # (the permuted columns are inputs here: the sort is a CPU step)
            chain(("lk", "S'"), d_lookup[3], side.cuda_stream)
        result["commitments"] = commit_batch(g_lagrange, d_lag)
        if "prof" not in result:
            result["prof"] = zk.msm_last_profile()
        t1 = time.perf_counter()
        # ---- 2 lookup: the permuted columns
        commit_batch(g_lagrange, d_lookup[2:4])
        t2 = time.perf_counter()
        # ---- 3 permutation: chunk c + 1 continues from the last value of chunk c; columns = 13 advice + 3 instance
        z_first = None
        pcols = [d_lag[N_INST + c] for c in range(NCOL)] + [d_lag[c] for c in range(N_INST)]
        for c in range(3):
            lo, hi = c * PERM_CHUNK, min(N_PERM_COLS, (c + 1) * PERM_CHUNK)
            z_first = H.permutation_product(sfield, pcols[lo:hi], d_sigma[lo:hi], beta, gamma, delta, k, d_zp[c], first_column_index=lo,
                                            z_first=z_first, stream=e.st)
        # ---- 4 lookup product.  Upstream writes the Z_P commitments, then the Z_L commitment, and squeezes no challenge in between
        # (both products only need beta, gamma): the four go into ONE batched commitment, behind both products
        H.lookup_product(sfield, d_lookup[0], d_lookup[1], d_lookup[2], d_lookup[3], beta, gamma, d_zl, stream=e.st)
        ev_zp.record(main)
        side.wait_event(ev_zp)
        with torch.cuda.stream(side):
            for c in range(3):
                chain(("zp", c), d_zp[c], side.cuda_stream)
            chain(("lk", "Z"), d_zl, side.cuda_stream)
        # the vanishing argument's random polynomial (5) depends on nothing but the RNG: its commitment (coefficient basis, another
        # key) runs on a library stream of its own beside the batch
        t_rand = None
        if e.world == 1 and not a.serial:
            t_rand = zk.msm_submit(g_coeff, d_coef[row[("random", 0)]], montgomery=True, window_bits=a.window_bits, stream=e.st, own_stream=True)
        commit_batch(g_lagrange, d_z)
        t3 = time.perf_counter()
        t4 = t3
        # ---- 5 vanishing argument's random polynomial, 6 quotient
        if t_rand is not None:
            result["random_commitment"] = t_rand.collect()
        else:
            result["random_commitment"] = commit(g_coeff, d_coef[row[("random", 0)]])
        main.wait_stream(side)                                      # every coset (and coefficient form) is complete from here on
        xn_int = to_int(xn)
        cols_of = lambda jj: ([d_cos[("adv", c)][jj] for c in range(NCOL)] + [d[jj] for d in d_fixed_cos]
                              + [d_cos[("lk", "A'")][jj], d_cos[("lk", "S'")][jj], d_cos[("lk", "Z")][jj]]
                              + [d_cos[("zp", c)][jj] for c in range(3)] + [d_cos[("inst", c)][jj] for c in range(N_INST)])
        if QP == 1:
            # upstream's order: the whole extended coset at once, extended_to_coeff, then the 8 pieces' commitments
            H.evaluate_expression(sfield, prog, cols_of(0), consts, ext, rsc, d_hp[0], stream=e.st, lazy=lazy_expr)
            dom.divide_by_vanishing_poly_part(d_hp[0], 0, 1, stream=e.st)
            dom.extended_to_coeff(d_hp[0], stream=e.st)
            result["h_commitments"] = commit_batch(g_coeff, hp_flat)
            H.vec_fold_many(sfield, d_coef[row[("h", 0)]], hp_flat, xn, stream=e.st, reverse=True)
        else:
            # Sub-coset by sub-coset: numerator, division by its one vanishing value, inverse transform of size m -> the folded
            # coefficients A_j (EvaluationDomain.part_to_coeff).  Their n-coefficient slices are committed in batches of 4 from a worker
            # thread on its own stream WHILE the next sub-cosets are evaluated here: upstream's chain expression -> extended_to_coeff ->
            # 8 MSMs has nothing to overlap, this one hides the expression behind the MSMs.  The pieces' commitments and the folded h(X)
            # the evaluation phase needs are fixed linear combinations of the slices' commitments / of the slices.
            futs, launched = [], 0
            for jj, j in enumerate(my_parts):
                H.evaluate_expression(sfield, prog, cols_of(jj), consts, ext - (QP.bit_length() - 1), rsc, d_hp[jj], stream=e.st, lazy=lazy_expr)
                dom.divide_by_vanishing_poly_part(d_hp[jj], j, QP, stream=e.st)
                dom.part_to_coeff(d_hp[jj], j, QP, stream=e.st)
                ready = (jj + 1) * r_sl
                while ready - launched >= 4 or (jj + 1 == PL and launched < ready):
                    hi = min(launched + 4, ready)
                    if a.serial:
                        futs.append(zk.msm_batch(g_coeff, hp_flat[launched:hi], montgomery=True, window_bits=a.window_bits, stream=e.st))
                    else:
                        ev_q = torch.cuda.Event()
                        ev_q.record(main)
                        msm_stream.wait_event(ev_q)
                        futs.append(pool.submit(zk.msm_batch, g_coeff, hp_flat[launched:hi], montgomery=True, window_bits=a.window_bits,
                                                stream=msm_stream.cuda_stream))
                    launched = hi
            # the folded quotient: sum over this rank's slices of e[j][s] * A_j[slice s] (then over the ranks)
            e_js = dom.fold_scalars(QP, xn_int)
            dst = d_coef[row[("h", 0)]] if e.world == 1 else d_hsum[e.rank]
            for jj, j in enumerate(my_parts):
                for s_ in range(r_sl):
                    if jj == 0 and s_ == 0:
                        dst.copy_(hp_flat[0])
                        zk.vec_op(sfield, "scale", dst, scalar=mont(e_js[j][0]), stream=e.st)
                    else:
                        H.vec_muladd(sfield, hp_flat[jj * r_sl + s_], dst, mont(e_js[j][s_]), stream=e.st, out=dst)
            C_loc = np.concatenate([f if a.serial else f.result() for f in futs])
            if not a.serial:
                main.wait_stream(msm_stream)
            if e.world > 1:
                max_rows = -(-QP // e.world) * r_sl                 # (an uneven deal -- 3, 5, 6, 7 ranks -- is padded)
                C_all, order = e.zkdist.gather_rows(C_loc, max_rows), {}
                for rk in range(e.world):
                    for jj, j in enumerate(jx for jx in range(QP) if jx % e.world == rk):
                        for s_ in range(r_sl):
                            order[(j, s_)] = rk * max_rows + jj * r_sl + s_
                e.zkdist.gather_stack(d_hsum[e.rank], d_hsum)
                hsum = d_coef[row[("h", 0)]]
                hsum.copy_(d_hsum[0])
                for rk in range(1, e.world):
                    zk.vec_op(sfield, "add", hsum, d_hsum[rk], stream=e.st)
            else:
                C_all, order = C_loc, {(j, s_): j * r_sl + s_ for j in range(QP) for s_ in range(r_sl)}
            rows_q = dom.piece_scalars(QP)
            used = sorted(order.values())                                # (padding rows of an uneven deal are not points)
            C_all, order = C_all[used], {kk: used.index(v) for kk, v in order.items()}
            result["h_commitments"] = H.combine_commitments(curve, C_all, [[(order[(j, s_)], sc) for j, s_, sc in terms] for _, terms in rows_q],
                                                            to_device=lambda arr: to_dev(e, arr), stream=e.st)
        t5 = time.perf_counter()
        # ---- 7 evaluations: h(X) = sum_i x^(n i) h_i, then every committed polynomial at x and at its rotations
        result["evals"] = [H.eval_polynomials(sfield, d_coef, x, stream=e.st),
                           H.eval_polynomials(sfield, d_coef[sets[1][0]:], x_next, stream=e.st),
                           H.eval_polynomials(sfield, d_coef[sets[2][0]:sets[2][0] + sets[2][1]], x_prev, stream=e.st),
                           H.eval_polynomials(sfield, d_coef[sets[3][0]:], x_last, stream=e.st)]
        t6 = time.perf_counter()
        # ---- 8 multiopen.  (The opening's blinding polynomial s depends on nothing but the RNG: its commitment is submitted here, on a
        # library stream of its own, and collected where upstream writes it -- at the start of the inner-product argument.)
        t_s = None
        if e.world == 1 and not a.serial:
            t_s = zk.msm_submit(g_coeff, d_spoly, montgomery=True, window_bits=a.window_bits, stream=e.st, own_stream=True)
        for s_, (first, cnt, pts) in enumerate(sets):          # each set's polynomials folded with x_1: one pass per set
            H.vec_fold_many(sfield, d_q[s_], d_coef[first:first + cnt], x1, stream=e.st)
        for s_, (first, cnt, pts) in enumerate(sets):          # q' = q' x_2 + (set polynomial / prod (X - point))
            dst = d_qprime if s_ == 0 else d_tmp
            H.kate_division(sfield, d_q[s_], pts[0], out=dst, stream=e.st)
            for pt in pts[1:]:
                H.kate_division(sfield, dst, pt, stream=e.st)
            if s_:
                H.vec_muladd(sfield, d_qprime, d_tmp, x2, stream=e.st)
        commit(g_coeff, d_qprime)
        result["q_evals"] = H.eval_polynomials(sfield, d_q, x3, stream=e.st)
        for s_ in range(len(sets)):                             # p = q' x_4^4 + ... : one Horner step per set
            H.vec_muladd(sfield, d_qprime, d_q[s_], x4, stream=e.st)
        t7 = time.perf_counter()
        if timed_:
            take_class("commit")
        # ---- 9 opening: the inner-product argument on p at x_3
        result["s_commitment"] = t_s.collect() if t_s is not None else commit(g_coeff, d_spoly)
        H.vec_muladd(sfield, d_spoly, d_qprime, xi, stream=e.st, out=d_ipa[0])            # p' = s xi + p
        result["v"] = H.eval_polynomial(sfield, d_ipa[0], x3, stream=e.st)               # (p'[0] -= v: one element, host side upstream)
        H.vec_powers(sfield, d_ipa[1], x3, stream=e.st)                                  # b
        if timed_:
            take_class("commit")
        if a.ipa == "fold":       # upstream's literal structure: collapse the generators every round
            d_g = result.setdefault("d_g", torch.empty_like(d_pts_c))
            d_g.copy_(d_pts_c)
            ipa = H.IpaProver(curve, d_ipa[0], d_ipa[1], d_g, stream=e.st)
        else:                     # L, R over the resident SRS with challenge-weighted scalars: no generator is ever folded
            ipa = H.IpaProverVirtual(curve, d_ipa[0], d_ipa[1], g_coeff, lambda shape: torch.zeros(shape, dtype=torch.int64, device="cuda"),
                                     stream=e.st, buffers=(d_S, d_W))
        small = False
        for j in range(k):
            ipa.round(sharded=e.world > 1)
            ipa.fold(us[j])
            if a.ipa == "collapse" and (j + 1) in collapse_at and j + 1 < k:
                ipa.collapse(sharded=e.world > 1)
                if timed_ and not small:
                    take_class("ipa_full")
                    small = True
        ipa.free()
        torch.cuda.synchronize()
        if timed_:
            take_class("ipa_small" if small else "ipa_full")
        t8 = time.perf_counter()
        if timed_:
            for name, dt in zip(PH, (t1 - t0, (t2 - t1) + (t4 - t3), t3 - t2, t5 - t4, t6 - t5, t7 - t6, t8 - t7)):
                phase_ms[name] += dt * 1e3

    elapsed = timed(e, step)
    # the library totals were drained class by class inside the steps: rebuild the whole-run sums for rooflines()
    tot = dict.fromkeys(e.msm_tot, 0)
    for cs in cls_sum.values():
        if cs:
            for kk, v in cs.items():
                tot[kk] += v
    for kk, v in e.msm_tot.items():
        tot[kk] += v
    e.msm_tot = tot
    if e.rank == 0:
        prof = result["prof"]
        mm = e.msm_tot
        n_chain = len(chain_names)
        n_commit = NLAG + 2 + 3 + 1 + 1 + N_H_PIECES + 1 + 1          # columns, A' S', Z_P x3, Z_L, random, h pieces, q', s
        n_eval = NP + n_s1 + n_s3 + n_s2 + n_s3
        n_kate = sum(len(pts) for _, _, pts in sets)
        line = base_line(e, n * a.steps / elapsed, elapsed,
                         "halo2 create_proof device work-list, 2^%d rows, the reference circuit's column layout (BASELINE configs[2]): %d instance + %d advice commits "
                         "(Lagrange basis) + %d NTT chains (l2c kept + coeff_to_extended) ; lookup: 2 + 1 commits, product ; permutation: 3 products + 3 commits ; "
                         "random-poly commit ; quotient: %d-op expression over %d extended columns, divide by Z_H, %s, %d h-piece commitments ; "
                         "%d evaluations at 4 points ; multiopen: 4 point sets (x_1 folds, %d kate divisions, x_2 fold, q' commit, evaluations at x_3, x_4 fold) ; "
                         "%d-round IPA on p (s commit, p' = s xi + p, b = powers of x_3 ; 2 MSMs + 2 inner products + 3 folds per round; generators %s)" % (
                             k, N_INST, NCOL, n_chain, len(prog), NCOL + N_FIXED + 6 + N_INST,
                             "extended_to_coeff" if QP == 1 else "per sub-coset (%d): inverse transform of size 2^%d, its %d slice commitment(s) issued beside the next "
                             "sub-coset's expression; pieces = linear combinations" % (QP, ext - (QP.bit_length() - 1), r_sl), N_H_PIECES, n_eval, n_kate, k,
                             {"fold": "folded every round", "virtual": "never folded: every MSM over the SRS",
                              "collapse": "materialised after round(s) %s by zk_ipa_collapse_device" % a.ipa_collapse_after}[a.ipa]),
                         {"rows_per_step": n, "msm_curve": curve, "msm_points": n, "msm_windows_done": prof["windows_done"], "msm_windows": prof["windows_total"],
                          "window_bits": prof["window_bits"], "columns": NCOL, "instance_columns": N_INST, "full_size_msms_per_step": n_commit,
                          "ntt_2p%d_per_step" % k: n_chain, "ntt_2p%d_per_step" % (ext - (QP.bit_length() - 1)): (n_chain + 1) * QP, "kate_divisions_per_step": n_kate, "evaluations_per_step": n_eval,
                          "extended_coset_parts": QP,
                          "streams": "one (serial)" if a.serial else "MSM batches on two library streams + NTT chains on a third",
                          "quotient_kernel": ("saturated interpreter" if not lazy_expr else "lazy limbs, interpreter" if a.expr_kernel == "never"
                                              else "lazy limbs, compiled for this expression (hiprtc, inside the first warm-up step)"),
                          "not_in_list": "RNG (blinding, random polynomials), transcript, the lookup argument's sort (CPU)"})
        if e.world > 1:
            line["config"]["parallelism"] = ("msm-window-shard x%d + all_gather ; extended coset in %d sub-cosets dealt round-robin: expression, inverse transform and "
                                             "slice commitments per rank, all_gather of %d commitments + one folded vector per rank" % (e.world, QP, QP * r_sl))
        line["phases_ms"] = {kk: v / a.steps for kk, v in phase_ms.items()}
        # the same proof elements whatever the order of work / number of ranks: a digest to compare runs by
        line["digest"] = {"h_commitments": hashlib.sha256(b"".join(zk.point_to_affine(curve, c_).tobytes() for c_ in result["h_commitments"])).hexdigest()[:16],
                          "evals_at_x": hashlib.sha256(result["evals"][0].tobytes()).hexdigest()[:16]}
        line["msms_per_step"] = mm["msms"] / a.steps
        by_class = {}
        for name, cs in cls_sum.items():
            if cs and cs["msms"]:
                by_class[name] = {"msms_per_step": cs["msms"] / a.steps, "launches_per_step": cs["launches"] / a.steps,
                                  "device_ms_per_msm": cs["device_ms"] / cs["msms"], "host_tail_ms_per_msm": cs["host_tail_ms"] / cs["msms"],
                                  "accumulate_kernel_us_per_launch": cs["accumulate_kernel_ms"] / cs["launches"] * 1e3}
        line["msm_by_class"] = by_class
        cm = cls_sum["commit"]
        if cm and cm["msms"]:     # BASELINE metric, second half: MSM Mop/s of the full-size (2^k-point, dense) commitments inside the work-list
            line["msm_mops"] = n * cm["msms"] / ((cm["device_ms"] + cm["host_tail_ms"]) * 1e-3) / 1e6
            line["msm_ms"] = (cm["device_ms"] + cm["host_tail_ms"]) / cm["msms"]
        line["ntt_kernel_ms_per_step"] = e.ntt_tot["kernel_ms"] / a.steps
        line["accumulate_kernel_ms_per_step"] = mm["accumulate_kernel_ms"] / a.steps
        if not a.no_cpu_baseline and e.world == 1:
            counts = {"msm": n_commit, "ntt_k": n_chain, "ntt_ext": n_chain + 1, "evals": n_eval, "kate": n_kate,
                      "folds": NP - len(sets) + len(sets) - 1 + len(sets) + N_H_PIECES - 1 + 1,
                      "expr_muls": sum(1 for o in prog if o[0] in ("mul", "scale"))}
            line["cpu_baseline"] = cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, lag_host[0], result["commitments"][0], counts)
        roofs = rooflines(e, "halo2_2p%d" % k)
        if "msm_accumulate_kernel" in roofs:     # per-launch figures of the full-size commitments (the IPA's launches are listed in msm_by_class)
            r = roofs["msm_accumulate_kernel"]
            if cm and cm["launches"]:
                ach = cm["algorithmic_bytes"] / (cm["accumulate_kernel_ms"] * 1e-3) / 1e9
                r.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "launches": cm["launches"], "msms": cm["msms"],
                          "avg_launch_us": cm["accumulate_kernel_ms"] / cm["launches"] * 1e3,
                          "algorithmic_bytes_per_launch": cm["algorithmic_bytes"] / cm["launches"]})
                r["note"] += "; per-launch figures = the %d full-size commitment MSMs of a step (up to 4 scalar vectors per launch); kernel_ms_total = all classes" % n_commit
        emit(e, line, roofs)


def cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, col0, gpu_commit0, counts):
    """The oracle ('port': CPU restatements of halo2_proofs 0.2 best_multiexp -- chunk per thread -- , best_fft, and per-element
    field / curve arithmetic) timed on this box on a bounded sample of the same work-list: ONE commitment (both reference MSM
    algorithms, the faster is used), ONE best_fft at 2^k and ONE at 2^(k+3), 2^k field products on one thread, 2^10 point
    multiplications on one thread and 2^17 on all threads.  `cores` = the CPUs this process may use (affinity / cgroup quota, not
    the host's core count); the embarrassingly parallel point multiplications also give the MEASURED parallel speed-up, and the
    threaded routines run with both thread counts (detected cores, measured speed-up) -- the faster time is used.  The step is
    assembled from the counts of each kind; the per-element field work is divided by the measured speed-up.  Also a last
    bit-exact check of the GPU's first commitment."""
    from oracle import zk_oracle as orc
    np, zk = e.np, e.zk
    cores = host_cores()
    n = 1 << k
    pts = d_pts.cpu().numpy().view(np.uint64)

    def best(fn, counts_):
        out, tbest, thr = None, None, None
        for t in counts_:
            t0 = time.perf_counter()
            r = fn(t)
            dt = time.perf_counter() - t0
            if tbest is None or dt < tbest:
                out, tbest, thr = r, dt, t
        return out, tbest, thr

    ks1 = e.synth.scalars_for(curve, 1 << 10, 78)
    t0 = time.perf_counter()
    orc.fixed_base_mul(curve, ks1, threads=1)                  # 255-bit double-and-add per point, one thread
    t_pmul1 = (time.perf_counter() - t0) / (1 << 10)
    ks = e.synth.scalars_for(curve, 1 << 17, 77)
    t0 = time.perf_counter()
    orc.fixed_base_mul(curve, ks, threads=min(256, cores))     # ... all threads, 2^17 points (>= 512 per thread)
    t_pmul = (time.perf_counter() - t0) / (1 << 17)
    speedup = max(1.0, t_pmul1 / t_pmul)                       # what the box really grants
    tc = sorted({min(256, cores), max(1, min(256, int(speedup + 0.5)))}, reverse=True)
    exp, t_h2, thr_h2 = best(lambda t: orc.msm_halo2(curve, pts, col0, threads=t), tc)
    canon = orc.from_mont(sfield, col0)
    ark_threads = min(cores, -(-255 // orc.ark_window_bits(n)))
    t0 = time.perf_counter()
    exp_ark = orc.msm_ark(curve, pts, canon, threads=ark_threads)
    t_ark = time.perf_counter() - t0
    w, w_ext = zk.root_of_unity(sfield, k), zk.root_of_unity(sfield, ext)
    _, t_n, thr_n = best(lambda t: orc.halo2_best_fft(sfield, col0, w, k, threads=t), tc)
    t0 = time.perf_counter()
    orc.halo2_best_fft(sfield, col0, w, k, threads=1)
    t_n1 = time.perf_counter() - t0
    big = e.synth.rand_field(sfield, 1 << ext, 0xF00D)
    _, t_e, thr_e = best(lambda t: orc.halo2_best_fft(sfield, big, w_ext, ext, threads=t), tc)
    t0 = time.perf_counter()
    orc.to_mont(sfield, col0)                                  # one Montgomery product per element, one thread, 2^k elements
    t_mul = (time.perf_counter() - t0) / n
    t_msm = min(t_h2, t_ark)
    muls_products = n * (4 * 8 + 3 * (4 * PERM_CHUNK + 8))      # factors, batched inversion, scan: lookup + three permutation chunks
    muls_expr = (1 << ext) * counts["expr_muls"]               # the products of the quotient program, at every row of the extended domain
    muls_open = n * (counts["evals"] + counts["folds"] + counts["kate"] + 2)   # Horner evaluations, folds, kate_division, b: one product per coefficient each
    t_field = (muls_products + muls_expr + muls_open) * t_mul / speedup
    # IPA: MSMs of 2 * (n/2 + n/4 + ...) = 2n points ~ two full MSMs; n point multiplications for the generator folds
    t_ipa = 2 * t_msm + n * t_pmul
    t_step = counts["msm"] * t_msm + counts["ntt_k"] * t_n + counts["ntt_ext"] * t_e + t_field + t_ipa
    ok = bool((zk.point_to_affine(curve, gpu_commit0) == exp).all() and (exp == exp_ark).all())
    return {"value": n / t_step, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "host: %d logical CPUs, %d usable (affinity / cgroup), measured parallel speed-up x%.1f (2^17 point multiplications: %.0f us on one thread, %.2f us per point "
                      "on %d threads) ; best_multiexp 2^%d: halo2 chunk-per-thread %.3f s (%d threads), ark window-parallel %.3f s (%d threads), faster one used ; best_fft 2^%d %.3f s "
                      "(%d threads; %.3f s on one: x%.1f -- upstream's serial bit-reversal and twiddle table bound it), 2^%d %.3f s (%d threads) ; field product %.0f ns (one thread, 2^%d "
                      "products) ; step = %d MSMs + %d + %d FFTs + %.2g products / %.1f + IPA (2 MSM + 2^%d point multiplications) = %.2f s"
                      % (os.cpu_count() or 1, cores, speedup, t_pmul1 * 1e6, t_pmul * 1e6, min(256, cores), k, t_h2, thr_h2, t_ark, ark_threads, k, t_n, thr_n, t_n1, t_n1 / t_n,
                         ext, t_e, thr_e, t_mul * 1e9, k, counts["msm"], counts["ntt_k"], counts["ntt_ext"], muls_products + muls_expr + muls_open, speedup, k, t_step),
            "msm_mops": n / t_msm / 1e6, "msm_s": {"halo2_chunked": t_h2, "ark_window_parallel": t_ark},
            "fft_s": {"2p%d" % k: t_n, "2p%d_one_thread" % k: t_n1, "2p%d" % ext: t_e},
            "point_mul_us": {"one_thread": t_pmul1 * 1e6, "per_point_all_threads": t_pmul * 1e6},
            "parallel_speedup_measured": speedup, "logical_cpus": os.cpu_count() or 1,
            "step_s": {"msm": counts["msm"] * t_msm, "fft": counts["ntt_k"] * t_n + counts["ntt_ext"] * t_e, "field": t_field, "ipa": t_ipa},
            "gpu_result_matches": ok}


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
    cores = host_cores()
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
        # (the G2 MSM first: its host tail -- Horner on Fq2 host limbs, the longest of the five -- then runs beside the G1 MSMs' device work)
        jobs = ((b_g2_query, z, False), (h_query, d_abc[0][:m - 1], True), (a_query, z, False), (b_g1_query, z, False), (l_query, z_aux, False))
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
    cores = host_cores()
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
