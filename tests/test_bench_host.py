"""Host-side logic of bench.py that needs no GPU: the PMC-summary lookup behind `roofline.traffic` withholds a byte count
that was profiled on other kernel sources than the ones this tree holds."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _summary(tmp_path, sha):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "latest_pmc_x3.csv").write_text(
        'kernel,counter,dispatches,mean_per_launch\n'
        '"k_gemm_bf16_256<true, 3, true, true, false, false, true>",FETCH_SIZE,8,1000.0\n'
        '"k_gemm_bf16_256<true, 3, true, true, false, false, true>",WRITE_SIZE,8,500.0\n'
        '"k_gemm_bf16_256<false, 6, true, true, false, false, true>",FETCH_SIZE,4,300.0\n'
        '"k_gemm_bf16_256<false, 6, true, true, false, false, true>",WRITE_SIZE,4,100.0\n'
        '"k_gemm_f32_sk",FETCH_SIZE,4,700.0\n"k_gemm_f32_sk",WRITE_SIZE,4,70.0\n'
        '"k_gemm_f32_sk_fixup",FETCH_SIZE,4,7.0\n"k_gemm_f32_sk_fixup",WRITE_SIZE,4,3.0\n')
    (prof / "latest_pmc.json").write_text(json.dumps({"latest_pmc_x3": {"date": "d", "commit": "c", "command": "x",
                                                                        "csrc_sha16": sha}}))


def test_pmc_traffic_is_withheld_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    real_root = bench.ROOT
    sha = bench.csrc_hash()
    assert len(sha) == 16 and sha == bench.csrc_hash()
    _summary(tmp_path, sha)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_hash", lambda: sha)
    # prefix lookup (trailing template arguments left open); FETCH_SIZE doubled (gfx950), KiB -> bytes
    tr, src = bench.pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name="latest_pmc_x3")
    assert tr == (2 * 1000.0 + 500.0) * 1024 and "latest_pmc_x3.csv" in src and "STALE" not in src
    tr, _ = bench.pmc_traffic("k_gemm_bf16_256<false, 6, true, true, false", name="latest_pmc_x3")
    assert tr == (2 * 300.0 + 100.0) * 1024
    # an exact name wins over a longer name that starts with it (round 5: the stream-K launch was reported with its fix-up's bytes)
    assert bench.pmc_traffic("k_gemm_f32_sk", name="latest_pmc_x3")[0] == (2 * 700.0 + 70.0) * 1024
    # a summary of ANOTHER workload is withheld like one of other kernel sources (stamps without a workload: config 1's)
    tr, src = bench.pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name="latest_pmc_x3", workload="rows=10000000 batch=8192 mode=inbatch")
    assert tr is None and "OTHER WORKLOAD" in src
    tr, _ = bench.pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name="latest_pmc_x3", workload="rows=1000000 batch=4096 mode=inbatch")
    assert tr == (2 * 1000.0 + 500.0) * 1024
    # another tree: the same summary is not evidence for its kernels
    monkeypatch.setattr(bench, "csrc_hash", lambda: "0" * 16)
    tr, src = bench.pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name="latest_pmc_x3")
    assert tr is None and "STALE" in src
    # a summary without a stamp (rounds 1-3) is treated the same way
    (tmp_path / "profiles" / "latest_pmc.json").write_text(json.dumps({"latest_pmc_x3": {"date": "d", "commit": "c"}}))
    assert bench.pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name="latest_pmc_x3")[0] is None
    assert bench.pmc_traffic("k_gemm_bf16_256<true, 3", name="no_such_summary") == (None, None)
    assert real_root == ROOT
