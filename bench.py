#!/usr/bin/env python3
"""Headline bench: the GPU work-list of ONE proof of the synthetic 2^20-row halo2 PoE circuit (BASELINE.json configs[2],
the configuration `metric` is quoted on), inputs resident in HBM, through the C ABI (libzkcp_amd.so).

  step (default, --workload halo2): column layout of the reference's ElGamalGadget (13 advice columns,
      circuits-halo2/src/encryption.rs:83-161; SURVEY 8a a11 / 8d "Config 3"):
        13 x commit(advice column)   = best_multiexp over Params::g_lagrange (Vesta, 2^k Montgomery scalars) -- one batched
                                       call, the columns share the bases
        13 x lagrange_to_coeff       = iNTT 2^k
        13 x coeff_to_extended       = zero-extend to 2^(k+3), zeta-coset shift, NTT 2^(k+3)
         1 x extended_to_coeff       = iNTT 2^(k+3) + coset un-shift (the quotient polynomial)
      The NTT chain of a column does not depend on the column's commitment, so it runs on a second HIP stream beside
      the MSMs.  IPA opening, permutation / lookup products and the transcript are not part of this line.
  metric    = constraints/sec = rows / wall-clock of the timed region (whole job)
  --workload column : BASELINE configs[1], one 2^20 MSM + one 2^20 NTT per step (the microbench; prints msm_mops)
  --workload groth16: the GPU work of one Groth16 proof at domain 2^logn (SURVEY 8d "Config 4")
  N > 1     = every MSM is window-range sharded over the N ranks (one process per GPU) and combined with one all_gather
              of Jacobian points over RCCL (contangle-zkcp_amd/dist.py); NTTs stay single-GPU, columns are dealt
              round-robin.  Total work per step is fixed -> "scaling": "strong".

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
                  "zk_msm_kernels.h", "zk_ntt_kernels.h", "zk_msm.inl", "zk_ntt.inl")


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
    ap.add_argument("--curve", default="Vesta", choices=["Vesta", "Pallas", "Bn254G1", "Bls381G1"])
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--realistic", action="store_true", help="0/1-heavy witness mix (SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="NTTs and MSMs on one stream, MSMs one at a time (no overlap)")
    ap.add_argument("--workload", default="halo2", choices=["halo2", "column", "groth16"])
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
            "launches": m["msms"], "avg_launch_us": m["accumulate_kernel_ms"] / m["msms"] * 1e3,
            "algorithmic_bytes_per_launch": m["algorithmic_bytes"] / m["msms"], "kernel_ms_total": m["accumulate_kernel_ms"],
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
            mads = {"Vesta": 135, "Pallas": 135, "Bn254G1": 162, "Bls381G1": 392}.get(line["config"].get("msm_curve", ""), None)
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
            "dtype": "u32 limbs (256/384-bit Montgomery integers; MSM buckets on 9 x 29 / 14 x 28-bit lazy limbs, NTT on 8 x 32)",
            "data": "synthetic", "config": dict({"workload": workload, "parallelism": "msm-window-shard x%d + all_gather" % e.world
                                                  if e.world > 1 else "single-gpu"}, **cfg)}


# ---------------------------------------------------------------------------------------------------- halo2 (default)
def bench_halo2(e):
    a, zk, torch, np = e.args, e.zk, e.torch, e.np
    curve = a.curve if a.curve in ("Vesta", "Pallas") else "Vesta"
    sfield = e.synth.CURVE_SCALAR_FIELD[curve]
    k, ext = a.logn, a.logn + 3
    n = 1 << k
    bases, d_pts = make_bases(e, curve, n, 0x5EED)
    dom = zk.halo2.EvaluationDomain(sfield, 9, k)      # degree-9 gates -> extended_k = k + 3 (Orchard-style)
    assert dom.extended_k == ext
    # 13 advice columns in Lagrange form (Montgomery residues, as halo2 holds them)
    cols_host = np.stack([e.synth.rand_field(sfield, n, 0xC0DE + c) for c in range(NCOL)])
    d_cols = to_dev(e, cols_host)                                   # [13, n, 4]
    mine = [c for c in range(NCOL) if c % e.world == e.rank]        # NTT chains this rank owns
    d_ext = [torch.empty((1 << ext, 4), dtype=torch.int64, device="cuda") for _ in mine]
    d_quot = torch.empty((1 << ext, 4), dtype=torch.int64, device="cuda")
    d_quot.copy_(to_dev(e, e.synth.rand_field(sfield, 1 << ext, 0xF00D)))
    main = torch.cuda.current_stream()
    side = main if a.serial else torch.cuda.Stream()
    result = {}

    def step(i, timed_):
        side.wait_stream(main)
        # NTT chains on the side stream ...
        with torch.cuda.stream(side):
            for j, c in enumerate(mine):
                d_ext[j][:n].copy_(d_cols[c], non_blocking=True)
                dom.lagrange_to_coeff(d_ext[j][:n], stream=side.cuda_stream)
                dom.coeff_to_extended(d_ext[j], stream=side.cuda_stream)
            if e.rank == 0:
                dom.extended_to_coeff(d_quot, stream=side.cuda_stream)
        # ... beside the 13 commitments (one batched call: same bases, MSMs alternate between two library streams)
        if a.serial:
            outs = [e.zkdist.msm_sharded(bases, d_cols[c], montgomery=True, window_bits=a.window_bits, stream=e.st) for c in range(NCOL)]
        else:
            outs = e.zkdist.msm_batch_sharded(bases, d_cols, montgomery=True, window_bits=a.window_bits, stream=e.st)
        main.wait_stream(side)
        result["commitments"] = outs

    elapsed = timed(e, step)
    if e.rank == 0:
        prof = zk.msm_last_profile()
        m = e.msm_tot
        line = base_line(e, n * a.steps / elapsed, elapsed,
                         "halo2 prover GPU work-list, 2^%d rows (BASELINE configs[2]): 13 x commit (%s MSM 2^%d) + 13 x lagrange_to_coeff (iNTT 2^%d) "
                         "+ 13 x coeff_to_extended (NTT 2^%d) + 1 x extended_to_coeff (iNTT 2^%d)" % (k, curve, k, k, ext, ext),
                         {"rows_per_step": n, "msm_curve": curve, "msm_points": n, "msm_windows_done": prof["windows_done"],
                          "msm_windows": prof["windows_total"], "window_bits": prof["window_bits"], "columns": NCOL,
                          "streams": "one (serial)" if a.serial else "MSM batch on two library streams + NTT chain on a third"})
        line["msm_ms"] = m["device_ms"] / max(1, m["msms"])
        line["msm_mops"] = n / (line["msm_ms"] * 1e-3) / 1e6 * (prof["windows_total"] / max(1, prof["windows_done"]))
        line["msm_phases_ms"] = {kk: m[kk] / max(1, m["msms"]) for kk in ("sort_ms", "accumulate_kernel_ms", "accumulate_ms", "reduce_ms", "host_tail_ms")}
        line["ntt_kernel_ms_per_step"] = e.ntt_tot["kernel_ms"] / a.steps
        if not a.no_cpu_baseline and e.world == 1:
            line["cpu_baseline"] = cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, cols_host[0], result["commitments"][0])
        emit(e, line, rooflines(e, "halo2_2p%d" % k))


def cpu_baseline_halo2(e, curve, sfield, k, ext, d_pts, col0, gpu_commit0):
    """The oracle ('port': CPU restatement of halo2_proofs 0.2 best_multiexp -- chunk per thread over ALL host cores -- and
    best_fft) timed on this box on a bounded sample of the same work-list: ONE column's commitment, ONE iNTT 2^k and ONE
    NTT 2^(k+3); the step is 13 x (msm + ntt_k + ntt_ext) + ntt_ext.  ark-ec's window-parallel Pippenger is timed beside
    it and the FASTER of the two MSMs is used.  Also a last bit-exact check of the GPU's first commitment."""
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
    t_msm = min(t_h2, t_ark)
    t_step = NCOL * (t_msm + t_n + t_e) + t_e
    ok = bool((zk.point_to_affine(curve, gpu_commit0) == exp).all() and (exp == exp_ark).all())
    return {"value": n / t_step, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "1 of the 13 columns: best_multiexp 2^%d (halo2 chunk-per-thread, %d threads: %.3f s; ark window-parallel, %d threads: %.3f s; "
                      "faster one used) + best_fft 2^%d (%.3f s) + best_fft 2^%d (%.3f s), %d threads; step = 13 x (msm + fft + ext fft) + ext fft = %.2f s"
                      % (k, cores, t_h2, ark_threads, t_ark, k, t_n, ext, t_e, cores, t_step),
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

    def step(i, timed_):
        if i % e.world == e.rank:
            side.wait_stream(main)
            zk.ntt(sfield, d_a, omega, stream=side.cuda_stream)
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
