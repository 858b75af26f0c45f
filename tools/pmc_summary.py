#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/pmc_run.sh (kernel stats, SQ counters, FETCH_SIZE / WRITE_SIZE) into
one text file and update profiles/traffic.json.
usage: pmc_summary.py gpurun_out/<dir> profiles/<name>.txt <workload key, e.g. halo2_2p20> [title]

traffic.json is stamped with the hash of the kernel sources it was measured on (bench.py drops `roofline.traffic` when the
sources have changed since) and keeps one entry per workload and hot kernel.

HBM traffic follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in units of 32 B... (the guide: on gfx950 FETCH_SIZE
reports kilobytes-like units of 64 B requests and HALF the bytes of wide coalesced streaming reads); the raw counter sums
are printed beside the byte figure so the correction stays visible."""
import collections, csv, glob, json, os, sys


def short(name):
    return name.split("(")[0].strip()


def main():
    src, dst, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    title = sys.argv[4] if len(sys.argv) > 4 else ""
    out = [title or "rocprofv3 passes of bench.py; per-launch averages", ""]
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        out.append("== --kernel-trace --stats")
        rows = list(csv.DictReader(open(stats[0])))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        for r in rows:
            out.append("%-78s calls %4s  avg %10.1f us  min %10.1f us  total %6.2f%%" % (
                short(r["Name"])[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])))
        out.append("")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for tag in ("pmc_sq", "pmc_fetch", "pmc_write"):
        files = glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k, v in seen.items():
            launches[(k, tag)] = len(v)
    out.append("== counters, per launch")
    traffic = {}
    for k in sorted(per, key=lambda k: -per[k].get("SQ_BUSY_CYCLES", 0)):
        c = per[k]
        n_sq = max(1, launches[(k, "pmc_sq")])
        out.append("%s   (launches %d)" % (k[:100], n_sq))
        for name in sorted(c):
            tag = "pmc_fetch" if name == "FETCH_SIZE" else "pmc_write" if name == "WRITE_SIZE" else "pmc_sq"
            out.append("    %-24s %.4g" % (name, c[name] / max(1, launches[(k, tag)])))
        if c.get("SQ_INSTS_VALU"):
            out.append("    lanes per VALU instr     %.2f of 64" % (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] if c.get("SQ_THREAD_CYCLES_VALU") else 0))
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            # counter unit on gfx950: kilobytes?  The guide's recipe: bytes = FETCH_SIZE * 1024 (KB units); wide streaming
            # reads are under-reported by 2x -- kept uncorrected here (gathers dominate these kernels), noted in DESIGN.md
            fb = c.get("FETCH_SIZE", 0) / max(1, launches[(k, "pmc_fetch")]) * 1024
            wb = c.get("WRITE_SIZE", 0) / max(1, launches[(k, "pmc_write")]) * 1024
            out.append("    HBM bytes (FETCH + WRITE, KB units)   %.4g + %.4g = %.4g" % (fb, wb, fb + wb))
            traffic[k] = fb + wb
    open(dst, "w").write("\n".join(out) + "\n")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_src_sha16
    tpath = os.path.join(os.path.dirname(dst), "traffic.json")
    sha = kernel_src_sha16()
    try:
        tj = json.load(open(tpath))
    except Exception:
        tj = {}
    if tj.get("kernel_src_sha16") != sha:
        tj = {"kernel_src_sha16": sha, "workloads": {}}
    tj["note"] = ("per launch: FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes (KB units x 1024).  gfx950 FETCH_SIZE reports half the "
                  "bytes of wide coalesced streaming reads (MI355X_MICROARCH.md): doubled for ntt_pass_kernel (16-B-per-lane streams), taken at "
                  "face value for msm_accumulate_kernel (random gathers of 64-byte packed records: uncalibrated pattern; Infinity-Cache hits are counted)")
    entry = {}
    for want, corr in (("msm_accumulate_kernel", 1.0), ("ntt_pass_kernel", 2.0)):
        ks = [k for k in per if (want in k or (want == "ntt_pass_kernel" and "ntt_pass29_kernel" in k)) and "big" not in k]
        if not ks:
            continue
        fetch = sum(per[k].get("FETCH_SIZE", 0) for k in ks) * 1024
        write = sum(per[k].get("WRITE_SIZE", 0) for k in ks) * 1024
        nf = sum(launches[(k, "pmc_fetch")] for k in ks)
        nw = sum(launches[(k, "pmc_write")] for k in ks)
        valu = sum(per[k].get("SQ_INSTS_VALU", 0) for k in ks)
        ns = sum(launches[(k, "pmc_sq")] for k in ks)
        entry[want] = {"hbm_bytes_per_launch": int(corr * fetch / max(1, nf) + write / max(1, nw)),
                       "fetch_bytes_raw_per_launch": int(fetch / max(1, nf)), "write_bytes_per_launch": int(write / max(1, nw)),
                       "fetch_correction": corr, "valu_wave_insts_per_launch": int(valu / max(1, ns)) if ns else None,
                       "launches_profiled": int(nf), "source": os.path.basename(dst)}
    tj["workloads"][workload] = entry
    json.dump(tj, open(tpath, "w"), indent=1)
    print("\n".join(out[:40]))


if __name__ == "__main__":
    main()
