"""bench.py's one-line JSON contract (task statement, section 4), checked on CPU against a line
recorded on an MI355X (profiles/r03_bench_line.json = `python bench.py` with no flags) and against
bench.py's source, so that a later edit cannot silently drop a field the driver or the judge reads."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOP = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
       "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict,
       "roofline": dict, "cpu_baseline": dict}
ROOFLINE = {"bound": str, "achieved": (int, float), "peak": (int, float), "unit": str, "frac": (int, float)}
CPU = {"value": (int, float), "unit": str, "cores": int, "kind": str, "sample": str}


def test_recorded_line_has_the_contract_fields():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_line.json")).read().strip().splitlines()[-1])
    for k, t in TOP.items():
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md has no published number
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k, t in ROOFLINE.items():
        assert isinstance(r[k], t), k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-3
    # config 2 batch = 1024 * 14 * 104896 bytes
    assert r["algorithmic_bytes_per_launch"] == 1503789056
    c = d["cpu_baseline"]
    for k, t in CPU.items():
        assert isinstance(c[k], t), k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # value = payload of all timed steps / time: 1024 blocks of 1 MiB per step
    assert abs(d["value"] - 1024 * 2**20 / (d["ms_per_step"] * 1e-3) / 2**30) / d["value"] < 2e-3
    # round 2 additions
    assert d["rccl_ranks"] == 1 and d["config"]["blocks_per_rank"] == [1024]
    assert d["parity_checked_blocks"] >= 32  # r03 line: 32 strided blocks; since then every block of the batch
    assert str(r["traffic_source"]).startswith("static:")
    assert 0.3 < r["cold_burst_frac"] < r["frac"] + 0.05
    sec = r["secondary"]
    assert 0 < sec["lds_busy_frac"] < 1 and 0 < sec["valu_busy_frac"] < 1 and sec["gpu_cycles"] > 0
    assert d["pcie_inclusive"]["encode_pageable_GiBps"] > 20 and d["block_manager"]["rpc_put_blocks_GiBps"] > 5
    # neither the PCIe-inclusive nor the BlockManager rate is the value
    assert d["value"] > 20 * d["pcie_inclusive"]["encode_pageable_GiBps"]
    # round 3 additions: the baseline is taken in a process of its own and says where,
    # the in-process figure and the product's CPU backend stand beside it, the degraded read is in the line
    assert "fresh subprocess" in c["process"] and c["host"]["omp_env"]["OMP_PROC_BIND"] == "close"
    assert c["host"]["lscpu_model_name"] and int(c["host"]["nproc"]) >= c["cores"]
    assert c["in_process"]["value"] > 0 and c["in_process"]["cores"] >= 1
    assert c["cpu_backend"]["value"] > 50 and c["cpu_backend"]["cpus_allowed"] >= c["cpu_backend"]["threads"]
    assert d["block_manager"]["rpc_get_blocks_4_nodes_down_GiBps"] >= 22
    assert "native callers" in d["block_manager"]["batcher_48_threads_put_source"]


def test_round5_line_carries_the_put_trip_and_config5s_code():
    """profiles/r05_bench_line.json = `python bench.py` with no flags on an MI355X (round 5)."""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")).read().strip().splitlines()[-1])
    for k, t in TOP.items():
        assert isinstance(d[k], t), k
    r = d["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 1503789056 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] >= r["algorithmic_bytes_per_launch"] and "round 05" in r["traffic_source"]
    eh, r2 = d["encode_hash"], d["rs20_8_encode"]
    assert eh["bit_exact"] is True and eh["ms"] <= 0.35 and eh["roofline"]["bound"] == "hbm"                     # VERDICT r04 item 2's bar
    assert abs(eh["roofline"]["frac"] - eh["roofline"]["algorithmic_bytes_per_launch"] / (eh["ms"] * 1e-3) / 1e9 / 8000) < 2e-3
    assert 0 < eh["binding_resource"]["valu_busy_frac"] < 1 and 1.0 <= eh["traffic"] / eh["roofline"]["algorithmic_bytes_per_launch"] < 1.05
    assert r2["bit_exact"] is True and abs(r2["roofline"]["frac"] - 28 * 209728 * 256 / (r2["kernel_ms"] * 1e-3) / 1e9 / 8000) < 2e-3
    bm = d["block_manager"]
    assert bm["small_trips"]["get_one_block_healthy_ms"]["rebuilt"] < 0.1 and bm["rpc_get_blocks_GiBps"] >= 35
    assert "rebuilt" in bm["verify_mode_default"]


def test_round6_line_times_both_decode_patterns_and_reports_the_default_mode():
    """profiles/r06_bench_line.json = `python bench.py` with no flags on an MI355X (round 6)."""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read().strip().splitlines()[-1])
    for k, t in TOP.items():
        assert isinstance(d[k], t), k
    r = d["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 1503789056 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] > 0.7
    assert r["traffic"] >= r["algorithmic_bytes_per_launch"] and "round 06" in r["traffic_source"]
    # config 3: both patterns BASELINE.md names, each timed, checked after its loop, with a roofline object that names the kernel
    dec = d["decode"]
    assert [p["lost"] for p in dec["patterns"]] == [[0, 3, 7, 9], [0, 3, 7, 11]] and dec["bit_exact"] is True
    for p in dec["patterns"]:
        rf = p["roofline"]
        assert p["bit_exact"] is True and rf["bound"] == "hbm" and rf["kernel"].startswith("gf_apply_nibble<1,0,10")
        assert abs(rf["frac"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9 / 8000) < 2e-3 and 0.6 < rf["frac"] < 1
        assert p["kernel_ms"] <= p["ms_per_step"] * 1.02
    assert dec["roofline"] == dec["patterns"][0]["roofline"] and "oracle" in dec["checked"]
    # the CPU backend's put trip: all 14 checksums for a fraction more than the encode (round 5: 6x; the two figures are separate
    # best-of-n on a shared host whose encode rate alone moves by a quarter, profiles/r06_experiments.txt section 6: 0.75 - 1.0)
    cb = d["cpu_baseline"]["cpu_backend"]
    assert cb["encode_plus_14_checksums_GiBps"] >= 0.7 * cb["value"]
    # the manager: the DEFAULT mode is what rpc_get_blocks_GiBps reports, and a healthy rebuilt-mode get costs what an off-mode get costs
    bm = d["block_manager"]
    modes = bm["rpc_get_blocks_by_verify_mode_GiBps"]
    assert bm["verify_mode_default"].startswith("always") and bm["rpc_get_blocks_GiBps"] == modes["always"]
    assert modes["rebuilt"] >= 0.9 * modes["off"]          # (two runs of one box differ by more than the 5 % the paths differ by)
    assert bm["small_trips"]["get_one_block_healthy_ms"]["rebuilt"] < 0.1 and bm["small_trips"]["put_one_block_ms"] < 0.13


def test_bench_source_still_emits_every_field():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in list(TOP) + ["vs_baseline"] + list(ROOFLINE) + ["traffic", "secondary", "cold_burst_frac", "traffic_source"] + list(CPU) + [
            "parity_checked_blocks", "rccl_ranks", "blocks_per_rank", "striped_decode", "exchange", "host_fed", "in_process",
            "cpu_backend", "process", "bit_exact_against", "host_load_during_sweep", "checked", "encode_plus_14_checksums_GiBps"]:
        assert re.search(rf'"{key}"\s*:', src) or f'["{key}"]' in src, key
    assert "max_over_ranks" in src and "barrier()" in src and "torch.cuda.synchronize()" in src
