#!/usr/bin/env python3
"""Summaries of round 6's profiling session (tools/r06_profile.sh): from the raw rocprofv3 CSVs under <dir> (kept under
profiles/raw/r06_prof/) to profiles/r06_*.txt and the static figures bench.py reports (profiles/pmc_traffic.json).
usage: r06_profile_summary.py <dir with bench/ put/ fetch/ write/ sq/>"""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(P, "raw", "r06_prof")
cite = "profiles/raw/r06_prof"


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


def durations(trace, substr, full_batch_only=False):
    rows = [r for r in csv.DictReader(open(trace)) if substr in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if full_batch_only and d:   # the same kernel also serves the host paths' small launches (tens of microseconds): keep the 1024-block ones
        d = [x for x in d if x > 0.6 * max(d)]
    return d


def per_dispatch(path, substr, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return sum(acc.values()) / len(acc) if acc else None


KERN = {
    "rs10_4_encode": ("gf_apply_nibble<1, 0, 10, 1, true, 256>", 14 * 104896 * 1024),
    "rs10_4_encode_sum": ("gf_apply_nibble_sum<1, 0, 10, true, 256>", 14 * 104896 * 1024),
    "rs20_8_encode": ("gf_apply_nibble<2, 0, 5, 1, true, 512>", 28 * 209728 * 256),
    "rs20_8_encode_sum": ("gf_apply_nibble_sum<2, 0, 5, true, 512>", 28 * 209728 * 256),
    "mlh_leaves": ("mlh_leaves", None),
    "mlh_roots": ("mlh_roots", None),
}

# ---- 1. the default bench command
bt, bs = one("bench/**/*kernel_trace.csv"), one("bench/**/*kernel_stats.csv")
line = None
try:
    line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
except Exception:  # noqa: BLE001
    pass
if bt:
    with open(os.path.join(P, "r06_bench_default_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5   (the driver's command)\n")
        f.write(f"# raw: {cite}/bench/ (kernel trace reduced to the library's kernels)\n")
        d = durations(bt, KERN["rs10_4_encode"][0], full_batch_only=True)
        if d:
            # the 20 timed launches are the last 20 before the verify launch; under the tracer the pre-conditioning launches dominate the list
            f.write(f"# gf_apply_nibble<1,0,10,1,true,256>: {len(d)} launches, median {statistics.median(d):.1f} us, min {min(d):.1f}, mean of the fastest half "
                    f"{statistics.mean(sorted(d)[:len(d) // 2]):.1f} us -> {14 * 104896 * 1024 / statistics.median(d) / 1e3:.0f} GB/s = "
                    f"{14 * 104896 * 1024 / statistics.median(d) / 1e3 / 8000:.3f} of 8 TB/s at the median (1024-block launches only)\n")
        if line:
            r = line["roofline"]
            f.write(f"# the line printed by the same run: value {line['value']} GiB/s, kernel_ms {r['kernel_ms']} (HIP events), frac {r['frac']}\n")
            for key in ("encode_hash", "rs20_8_encode"):
                if key in line and "roofline" in line[key]:
                    f.write(f"#   {key}: {line[key].get('ms', line[key].get('kernel_ms'))} ms, frac {line[key]['roofline']['frac']}, bit_exact {line[key].get('bit_exact')}\n")
        f.write("%7s %10s %10s %10s %7s  kernel\n" % ("calls", "avg_us", "min_us", "max_us", "%"))
        for r in list(csv.DictReader(open(bs)))[:18]:
            f.write("%7s %10.1f %10.1f %10.1f %7.2f  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                         float(r["Percentage"]), r["Name"][:120]))
    if line:
        with open(os.path.join(P, "r06_bench_line.json"), "w") as f:
            f.write(json.dumps(line) + "\n")

# ---- 2. the put-path kernels, 200 launches each back to back
pt = one("put/**/*kernel_trace.csv")
put = {}
if pt:
    with open(os.path.join(P, "r06_put_path_kernels.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace -- python tools/prof_encode_hash.py 200: 200 back-to-back launches of each device-resident put-path kernel\n")
        f.write(f"# (RS(10,4) x 1024 x 1 MiB and RS(20,8) x 256 x 4 MiB); median of the last 150 launches (the first 50 warm the clocks).  raw: {cite}/put/\n")
        f.write("%-46s %9s %9s %9s %12s\n" % ("kernel", "median_us", "min_us", "launches", "frac_of_8TBps"))
        for name, (sub, algo) in KERN.items():
            d = durations(pt, sub)
            if not d:
                continue
            # mlh_leaves / mlh_roots run for both codes: report all launches together
            tail = d[-150:] if len(d) >= 200 else d
            med = statistics.median(tail)
            put[name] = med
            f.write("%-46s %9.1f %9.1f %9d %12s\n" % (sub[:46], med, min(d), len(d), f"{algo / med / 1e3 / 8000:.3f}" if algo else "-"))
        if "rs10_4_encode_sum" in put and "mlh_roots" in put:
            f.write(f"# the put trip on config 2 = gf_apply_nibble_sum + mlh_roots: {put['rs10_4_encode_sum'] + put['mlh_roots']:.1f} us of kernel time; the encode alone {put.get('rs10_4_encode', 0):.1f} us\n")

# ---- 3. PMC passes
fp, wp, sp = one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv"), one("sq/**/*counter_collection.csv")
js_path = os.path.join(P, "pmc_traffic.json")
js = json.load(open(js_path))
if fp and wp:
    with open(os.path.join(P, "r06_pmc_hbm_traffic.txt"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domain beside them) -- python tools/prof_encode_hash.py 5\n")
        f.write(f"# both in KiB; on gfx950 FETCH_SIZE reports half of wide coalesced reads (calibrated in round 4 on a known copy: x2), WRITE_SIZE 1.00x.  raw: {cite}/fetch/, {cite}/write/\n")
        f.write("%-46s %14s %14s %16s %16s %8s\n" % ("kernel", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "traffic_bytes", "algorithmic", "ratio"))
        for name, (sub, algo) in KERN.items():
            fe, wr = per_dispatch(fp, sub, "FETCH_SIZE"), per_dispatch(wp, sub, "WRITE_SIZE")
            if fe is None or wr is None:
                continue
            traffic = int(2 * fe * 1024 + wr * 1024)
            f.write("%-46s %14.1f %14.1f %16d %16s %8s\n" % (sub[:46], fe, wr, traffic, algo or "-", f"{traffic / algo:.4f}" if algo else "-"))
            key = {"rs10_4_encode": "rs10_4_encode_1MiB_x1024", "rs10_4_encode_sum": "rs10_4_encode_hash_1MiB_x1024", "rs20_8_encode": "rs20_8_encode_4MiB_x256"}.get(name)
            if key:
                js[key] = {"traffic_bytes": traffic, "algorithmic_bytes": algo, "round": 6,
                           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round 06 (2*{fe:.1f} KiB + {wr:.1f} KiB), profiles/r06_pmc_hbm_traffic.txt"}
if sp:
    with open(os.path.join(P, "r06_pmc_sq.txt"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE (one pass) -- python tools/prof_encode_hash.py 5; per launch.  raw: " + cite + "/sq/\n")
        f.write("# derived figures (profiles/pmc_traffic.json, round 4's conventions): gpu cycles = GRBM_GUI_ACTIVE / 8 XCDs; VALU issue cycles per SIMD =\n"
                "# SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs; LDS array cycles per CU = SQ_LDS_IDX_ACTIVE / 256 CUs\n")
        cnts = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
        f.write("%-46s " % "kernel" + " ".join("%16s" % c[-16:] for c in cnts) + "\n")
        for name, (sub, algo) in KERN.items():
            vals = [per_dispatch(sp, sub, c) for c in cnts]
            if vals[0] is None:
                continue
            f.write("%-46s " % sub[:46] + " ".join("%16.0f" % v for v in vals) + "\n")
            v = dict(zip(cnts, vals))
            # round 4's conventions (profiles/r04_pmc_sq.txt): GRBM_GUI_ACTIVE is summed over 8 XCDs; LDS array cycles per CU = SQ_LDS_IDX_ACTIVE / 256;
            # VALU issue cycles per SIMD = SQ_ACTIVE_INST_VALU * 4 / 1024 (a wave64 VALU instruction occupies its SIMD for 4 cycles)
            gpu_cycles = v["GRBM_GUI_ACTIVE"] / 8
            sec = {"_comment": f"per launch of {sub}, rocprofv3 SQ pass of round 06 (profiles/r06_pmc_sq.txt); cycles = GRBM_GUI_ACTIVE / 8 XCDs",
                   "gpu_cycles": int(gpu_cycles), "lds_array_cycles_per_cu": int(v["SQ_LDS_IDX_ACTIVE"] / 256),
                   "lds_busy_frac": round(v["SQ_LDS_IDX_ACTIVE"] / 256 / gpu_cycles, 3),
                   "lds_bank_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4) if v["SQ_LDS_IDX_ACTIVE"] else 0.0,
                   "valu_issue_cycles_per_simd": int(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024), "valu_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gpu_cycles, 3),
                   "waves_parked_on_waitcnt_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)}
            key = {"rs10_4_encode": "rs10_4_secondary_bounds", "rs10_4_encode_sum": "rs10_4_encode_hash_secondary_bounds", "rs20_8_encode": "rs20_8_secondary_bounds"}.get(name)
            if key:
                if name == "rs10_4_encode_sum":
                    # the 56 v_mad_u64_u32 per lane and tile are half-rate (tools/csum_probe): 4 more cycles each than the 4 counted above
                    tiles = 1024 * ((104896 // 16 + 255) // 256)
                    extra = tiles * 4 * 56 * 4 / 1024
                    sec["valu_busy_frac_with_half_rate_multiplies"] = round((v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 + extra) / gpu_cycles, 3)
                    sec["statement"] = ("no single resource is saturated: with the checksums accumulated in its registers the kernel issues 18 % more VALU instructions "
                                        "(56 half-rate v_mad_u64_u32 per lane and tile) and 30 % more LDS cycles (the cross-lane reduction), which compete with the "
                                        "table lookups for issue slots while HBM traffic stays at 1.01x algorithmic; the launch is 11 % longer than the plain encode")
                elif name == "rs10_4_encode":
                    sec["statement"] = "HBM is the binding resource; next would be VALU issue (busy this fraction of the launch's cycles on every SIMD), then the LDS array"
                else:
                    sec["statement"] = "HBM (28 strided streams per workgroup) is the binding resource; VALU issue and LDS stay below it"
                js[key] = sec
# ---- 4. the striped decode's three exchanges at world 1
stt, sts = one("striped/**/*kernel_trace.csv"), one("striped/**/*kernel_stats.csv")
if sts:
    sline = None
    try:
        sline = json.loads(open(os.path.join(src, "striped_line.json")).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        pass
    with open(os.path.join(P, "r06_striped.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --op striped-decode --steps 400: BASELINE config 5 at WORLD 1 (RCCL with one\n"
                "# rank: the only world this pool has): RS(20,8), 256 x 4 MiB, 8 erasures, the three exchanges one after the other on VALID stripes, every\n"
                f"# object of the timed batch checked after each exchange's loop.  raw: {cite}/striped/\n")
        if sline:
            f.write("# the line of the same run: " + json.dumps({k: {kk: v.get(kk) for kk in ("ms_per_step", "GiBps", "bit_exact", "bit_exact_objects", "verify_batch_dev", "oracle_sample_objects")}
                                                              for k, v in sline.get("exchange", {}).items()}) + ("  ERROR: " + sline["error"] if "error" in sline else "") + "\n")
        f.write("%7s %10s %10s %10s %7s  kernel\n" % ("calls", "avg_us", "min_us", "max_us", "%"))
        for r in list(csv.DictReader(open(sts)))[:12]:
            f.write("%7s %10.1f %10.1f %10.1f %7.2f  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                         float(r["Percentage"]), r["Name"][:120]))
        f.write("# peer form per call = gf_apply_ptrs<2,5,...> alone (tables cached on the device, rebuilt ranges stored in place); all-gather form = ncclAllGather\n"
                "# (rcclGenericKernel) + gf_apply_nibble<2,0,5,1,true,512> + range exchange; all-to-all = a2a_pack + grouped ncclSend/Recv in 512 MiB pieces +\n"
                "# gf_apply_nibble + rebuilt_unpack.  See profiles/r06_experiments.txt sections 2 and 3.\n")
json.dump(js, open(js_path, "w"), indent=1)
print("summaries written under", P)
