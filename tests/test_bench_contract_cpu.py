"""CPU: the JSON contract of bench.py that does not need a GPU -- the `roofline` / `rooflines` objects built from an
instrumented pass, the PMC traffic attachment, and the committed bench lines of the round (profiles/r02_bench*.json)."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _prof():
    # one instrumented 512-pair step, numbers of the order of profiles/r02_bench.json; bytes = FLOPs for the MFMA classes and the
    # phases, aux = algorithmic HBM bytes (MFMA classes: of the launch; phases: of every instrumented launch inside)
    return {
        "gemm_kernel": {"ms": 53.3, "launches": 130, "bytes": 4.51e12, "aux": 6.5e10},
        "knn_query_kernel": {"ms": 10.4, "launches": 12, "bytes": 2.66e9, "aux": 0.0},
        "geo_table_kernel": {"ms": 3.52, "launches": 1, "bytes": 6.78e9, "aux": 0.0},
        "geo_embed_reference_flops": {"ms": 0.0, "launches": 1, "bytes": 3.27e12, "aux": 0.0},
        "local_attn_kernel": {"ms": 20.4, "launches": 15, "bytes": 4.4e10, "aux": 0.0},
        "phase.global_transformer": {"ms": 20.3, "launches": 1, "bytes": 7.43e11, "aux": 3.9e10},
        "phase.forward": {"ms": 111.7, "launches": 1, "bytes": 4.51e12, "aux": 1.6e11},
    }


def test_rooflines_from_an_instrumented_pass():
    b = _bench()
    roofs = b.rooflines(_prof(), 1, "f32")
    kinds = [r["kernel"].split(" ")[0] for r in roofs]
    assert kinds[0] == "gemm_kernel" and "knn+ppf" in kinds and "global_transformer" in kinds and "geo_table_kernel" in kinds
    for r in roofs:
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert "traffic" in r and r["avg_launch_ms"] > 0
        assert "frac_reference_formulation" not in r       # not a roofline fraction (round-2 review): gone from the entries
    g = roofs[0]
    assert g["bound"] == "mfma" and g["peak"] == b.MFMA_F32_PEAK_TFLOPS
    assert abs(g["achieved"] - 4.51e12 / 53.3e-3 / 1e12) < 1e-2
    assert g["algorithmic_bytes_per_launch"] == int(6.5e10 / 130)       # the dominant kernel's wasted-traffic ratio is computable
    ph = roofs[kinds.index("global_transformer")]
    # the phase is the E stream: priced against HBM on algorithmic bytes, its executed matrix work rides along
    assert ph["bound"] == "hbm" and abs(ph["achieved"] - 3.9e10 / 20.3e-3 / 1e9) < 1.0
    assert abs(ph["mfma_tflops_executed"] - 7.43e11 / 20.3e-3 / 1e12) < 1e-2 and ph["mfma_frac_executed"] < 1.0
    assert "not executed" in ph["note"]
    t = roofs[kinds.index("geo_table_kernel")]
    assert t["bound"] == "hbm" and t["peak"] == b.HBM_PEAK_GBS
    # the FLOP carrier of the reference formulation is not a kernel: never the dominant entry, never a roofline of its own
    assert all("geo_embed_reference_flops" not in r["kernel"] for r in roofs)
    assert b.rooflines({}, 0, "f32") == []
    assert b.rooflines(_prof(), 1, "bf16")[0]["peak"] == b.MFMA_BF16_PEAK_TFLOPS


def test_whole_forward_entry():
    b = _bench()
    w = b.whole_forward(_prof(), 1, "f32", 111.7, {"total_hbm_bytes_per_step": 2.47e11})
    assert abs(w["tflops"] - 4.51e12 / 111.7e-3 / 1e12) < 1e-2 and abs(w["mfma_frac"] - w["tflops"] / b.MFMA_F32_PEAK_TFLOPS) < 1e-4
    assert abs(w["hbm_gbs"] - 2.47e11 / 111.7e-3 / 1e9) < 1.0 and abs(w["hbm_frac"] - w["hbm_gbs"] / b.HBM_PEAK_GBS) < 1e-4
    assert b.whole_forward(_prof(), 1, "f32", 111.7, None)["executed_flops_per_step"] == int(4.51e12)


def test_traffic_is_attached_from_the_committed_pmc_summary(monkeypatch):
    b = _bench()
    from roitr_amd import build
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    # counters ride on a timing only when they were collected from the SAME kernel sources (stamp of scripts/pmc_summary.py):
    # another build's counters leave `traffic` null and say why
    stale = b.rooflines(_prof(), 1, "f32")
    monkeypatch.setattr(build, "source_hash", lambda: "0123456789abcdef")
    assert b.attach_traffic(stale, pmc["pairs_per_step"], pmc.get("baseline_config", 2)) is None
    assert stale[0]["traffic"] is None and "re-collect" in stale[0]["traffic_note"]
    monkeypatch.setattr(build, "source_hash", lambda: pmc.get("kernel_source_sha16"))
    roofs = b.rooflines(_prof(), 1, "f32")
    b.attach_traffic(roofs, pmc["pairs_per_step"], pmc.get("baseline_config", 2))
    # the instrumented class spans both kernels of csrc/gemm.hip: launch-weighted mean, the same launch set as `achieved`
    ks = [pmc["kernels"][n] for n in ("gemm_kernel", "gemm_small_kernel") if n in pmc["kernels"]]
    want = round(sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / sum(k["launches"] for k in ks))
    assert roofs[0]["traffic"] == want and "traffic_source" in roofs[0]
    assert abs(roofs[0]["traffic_over_algorithmic"] - roofs[0]["traffic"] / roofs[0]["algorithmic_bytes_per_launch"]) < 1e-3
    knn = [r for r in roofs if r["kernel"].startswith("knn+ppf")][0]
    assert knn["traffic"] and knn["traffic"] > 0
    other = b.rooflines(_prof(), 1, "f32")
    b.attach_traffic(other, pmc["pairs_per_step"] + 1, 2)      # a different workload: nothing is attached
    assert other[0]["traffic"] is None


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench*.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        for k in REQUIRED:
            assert k in d, (f, k)
        if d["config"].get("baseline_config") == 5:      # the kNN + PPF kernel micro-benchmark: its own metric
            assert d["unit"] == "queries/s" and d["roofline"]["bound"] == "hbm"
            continue
        assert d["metric"] == "point-cloud pairs/s" and d["unit"] == "pairs/s" and d["higher_is_better"] is True
        assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
        assert d["dtype"] in ("f32", "bf16", "f32x3") and "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(d["value"] - d["config"]["pairs_per_step"] * d["n_gpus"] * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]
            assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "pairs/s" and c["sample"]
