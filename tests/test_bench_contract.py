"""CPU suite: the committed bench line of the round (profiles/r03/bench_R512.json, written by `python bench.py` on an MI355X) carries
what the measurement contract asks for, and its roofline block can be recomputed from the committed counters."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    return json.load(open(os.path.join(ROOT, "profiles", "r03", "bench_R512.json")))


def test_contract_keys(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "ns/day" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12)
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["cores"] == 1 and c["kind"] == "port" and c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] > c["value"]


def test_roofline_is_recomputable_from_the_committed_counters(line):
    r = line["roofline"]
    # achieved = algorithmic bytes per launch / measured launch duration
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["usec_per_launch"] * 1e-6) / 1e9, rel=1e-9)
    # the launch duration is the mean of the two kinds of pass (current pruned lists / every list re-derived), weighted with the
    # share of atoms that re-derived their list in the force passes of this run
    k = r["usec_per_launch_by_kind"]
    w = k["share_of_atoms_re_deriving"]
    assert 0.0 < w < 0.2 and k["pruned_lists"] < k["re_deriving_every_list"]
    assert r["usec_per_launch"] == pytest.approx((1.0 - w) * k["pruned_lists"] + w * k["re_deriving_every_list"], rel=1e-9)
    R = line["config"]["replicas_per_gpu"]
    assert r["algorithmic_bytes_per_launch"] == 36.0 * 23400 * R
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_nonbonded.json")))["rotmove_R%d" % R]
    assert r["pmc_source"]["source_sha"] == pmc["source_sha"]
    c = pmc["counters_per_launch"]
    assert r["traffic"] == pytest.approx(1024.0 * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]), rel=1e-9)
    v = r["valu"]
    assert v["insts_per_launch"] == c["SQ_INSTS_VALU"]
    assert v["frac"] == pytest.approx(c["SQ_INSTS_VALU"] / (r["usec_per_launch"] * 1e-6) / (1024 * 2.4e9 / 2.0), rel=1e-9)
    assert 0.5 < v["valu_busy_frac"] < 1.0 and 0.3 < r["frac"] < 0.5
    # the traffic the kernel really causes is below the nominal bytes: no wasted re-reads
    assert r["traffic"] < r["algorithmic_bytes_per_launch"]
    # fewer listed pairs than the full lists hold, all pairs in range among them
    p = r["pairs"]
    assert p["in_range_per_launch"] < p["listed_per_launch"] < p["full_lists_per_launch"] and 0.6 < p["lane_efficiency"] < 1.0


def test_the_round_is_faster_than_the_last(line):
    last = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench_R512.json")))
    assert line["config"]["replicas_per_gpu"] == last["config"]["replicas_per_gpu"] == 512
    assert line["value"] > 1.15 * last["value"]
    assert line["roofline"]["usec_per_launch"] < 0.8 * last["roofline"]["usec_per_launch"]
