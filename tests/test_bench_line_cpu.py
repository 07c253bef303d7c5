"""bench.py's driver-facing line: short (< 4 KB), strict JSON, carries the contract's keys; `roofline.traffic` comes from the
PMC summaries committed under profiles/, not from a typed constant (BENCH_r04 `parsed` was null)."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOF = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms")


def _canned():
    """a full `result` as main() builds it: the largest one committed (20 KB as a line) + a few hostile additions"""
    with open(os.path.join(ROOT, "profiles", "r04b_default_bench.json")) as f:
        r = json.load(f)
    r["config"]["workload"] = "NS: 40 views x 256 ch x 480x640 fp32 maps (channels_last) -> grid 192x192x192, N=300"
    r["config"]["scenes_per_step"] = r["scenes_per_step"]
    r["kernels"]["x" * 300] = dict(ms_per_scene=1.0)
    r["conv_layers"] = r["conv_layers"] * 4
    r["train_S"] = dict(value=40.0, ms_per_step=25.0, note="n" * 5000)
    r["A"] = dict(value=240.0, ms_per_step=100.0, graph_nodes_per_scene=290, n_classes=17, n_reg_outs=8, workload="A" * 400,
                  windows_scenes_per_s=[240.0, 241.0, 239.0])
    r["dist"] = dict(world_size=8, backend="nccl", rank=0, per_rank_window_s=[[2.0] * 8] * 3)
    return r


def test_line_is_short_strict_json_with_the_contract_keys():
    r = _canned()
    text = bench.compact_line(r)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(r["value"], rel=1e-3)
    assert line["unit"] == "scenes/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    for roof in (line["roofline"], line["S"]["roofline"]):
        assert tuple(roof) == ROOF
        assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
        assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-3)
    cb = line["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] in ("port", "reference")
    assert len(cb["sample"]) <= 200
    assert isinstance(line["config"]["workload"], str) and len(line["config"]["workload"]) < 300
    assert "model" not in line["config"]
    for k in ("kernels", "conv_layers", "stage_ms"):
        assert k not in line and k not in line["S"]
    assert line["through_plugin"]["value"] and line["nchw_input"]["value"] and line["value_f32_conv"]
    assert line["A"] == dict(value=240.0, ms_per_step=100.0, graph_nodes_per_scene=290, n_reg_outs=8)      # BASELINE configs[2]
    assert line["detail"] == bench.DETAIL_NAME


def test_line_never_exceeds_the_limit_even_with_oversized_fields():
    r = _canned()
    r["config"]["workload"] = "w" * 3000
    r["config"]["level_rows"] = list(range(400))
    text = bench.compact_line(r)
    assert len(text) < 4096
    json.loads(text)


def test_nan_is_refused_not_printed():
    r = _canned()
    r["value"] = float("nan")
    with pytest.raises(ValueError):
        bench.compact_line(r)


def test_line_without_secondary_blocks():
    r = _canned()
    for k in ("S", "through_plugin", "nchw_input", "f32_conv", "value_f32_conv", "cpu_baseline", "dist", "train_S"):
        r.pop(k, None)
    line = json.loads(bench.compact_line(r))
    assert line["cpu_baseline"] is None and "S" not in line


def test_emit_writes_detail_and_prints_one_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r = _canned()
    bench.emit(r)
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1
    json.loads(out)
    with open(tmp_path / bench.DETAIL_NAME) as f:
        detail = json.load(f)
    assert "kernels" in detail and "conv_layers" in detail and detail["S"]["kernels"]


def test_roofline_traffic_is_read_from_the_committed_pmc_passes():
    import csv
    for wl in ("NS", "S"):
        for fam in ("dense", "conv"):
            got = bench.pmc_traffic(wl, fam)
            assert got is not None, (wl, fam, bench.PMC_TAG)
            total, src = got
            assert f"profiles/{bench.PMC_TAG}_{wl.lower()}_pmc_FETCH_SIZE.csv" in src
            # recompute by hand from the two files
            want = 0.0
            for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
                kb = n = 0
                with open(os.path.join(ROOT, "profiles", f"{bench.PMC_TAG}_{wl.lower()}_pmc_{counter}.csv"), newline="") as f:
                    for row in csv.DictReader(f):
                        if any(k in row["Kernel_Name"] for k in bench.PMC_KERNELS[fam]):
                            kb += float(row["Sum"])
                            n += int(row["Dispatches"])
                want += mult * kb * 1024 / n
            assert total == pytest.approx(want, rel=1e-12)
    # the dense kernel at NS: ~60 GB per launch against 19.86 GB algorithmic
    assert 40e9 < bench.pmc_traffic("NS", "dense")[0] < 80e9
    assert bench.pmc_traffic("NS", "dense", tag="no_such_round") is None
