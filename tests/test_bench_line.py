"""The ONE JSON line of bench.py's contract stays under 6 KB and is the LAST stdout line.  (Round 3's line had grown to
24 KB; the driver keeps about 8 KB of stdout, so its record of that round started in the middle of `extra.cfg4` and
could not be parsed.)  Canned input: the full result of a default run (profiles/r03z_default_bench.json, every prose
field included) with the time_to_target / exact_order legs in their current shape."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r03z_default_bench.json")))
    t2t = {"train_samples": 1000000, "held_out_samples": 200000, "problem": "labels from a planted degree-2 FM " + "x" * 200,
           "held_out_loss_at_start": 0.693169, "planted_model_held_out_loss": 0.403512, "targets_are": "y" * 300,
           "sequential": [{"epochs": e, "seconds": 0.7747 * e, "held_out_loss": 0.69 - 0.01 * e, "gap_closed": 0.1234 * i}
                          for i, e in enumerate((1, 3, 10), 1)],
           "minibatch": [{"batch": b, "touch_cap": 16.0, "epochs_run": 40, "held_out_loss": 0.612345, "seconds_per_epoch": 0.01789,
                          "targets": [{"seq_epochs": e, "target": 0.69 - 0.01 * e, "reached": b != 32768 or e < 10, "epochs": 3 * e,
                                       "seconds": 0.0531 * e, "speedup": 43.64 if (b != 32768 or e < 10) else None} for e in (1, 3, 10)]}
                         for b in (32768, 8192, 2048)],
           "best_batch": 2048, "note": "z" * 200}
    for leg in [full] + list(full["extra"].values()):
        leg["time_to_target"] = t2t
        leg["exact_order"]["bit_equal"] = True
        leg["exact_order"]["bit_equal_sample"] = "w" * 120
    return full


def test_contract_line_is_short_and_complete():
    sys.path.insert(0, ROOT)
    import bench

    full = canned()
    assert len(json.dumps(full)) > 20000  # the detail is as verbose as round 3's line was
    line = bench.contract_line(full)
    assert "\n" not in line and len(line) < 6000, len(line)
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])
    assert out["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(out["cpu_baseline"])
    assert "workload" in out["config"] and "model" not in out["config"] and len(out["config"]["update_rule"]) <= 120
    assert out["exact_order"]["bit_equal"] is True
    assert [h["seq_epochs"] for h in out["time_to_target"]["targets"]] == [1, 3, 10]
    assert out["time_to_target"]["batch"] == 2048
    assert sorted(out["extra"]) == ["cfg2", "cfg3", "cfg4", "cfg5"]
    for e in out["extra"].values():
        assert e["frac"] is not None and e["value"] > 0 and e["reached"] is True and len(e["t2t_speedup"]) == 3


def test_contract_line_sheds_optional_parts_rather_than_overflow():
    sys.path.insert(0, ROOT)
    import bench

    full = canned()
    full["extra"] = {("cfg%d" % i): v for i in range(40) for v in [full["extra"]["cfg2"]]}  # an absurd number of legs
    line = bench.contract_line(full)
    assert len(line) < 6000
    out = json.loads(line)
    assert out["roofline"]["frac"] == full["roofline"]["frac"] and out["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def test_last_8000_bytes_of_stdout_parse(tmp_path):
    """what the driver does: keep the tail of stdout, parse its last line"""
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench, test_bench_line as t; "
            "print('RCCL version : banner on stdout'); bench.ROOT = %r; bench.emit(t.canned())" % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    tail = r.stdout[-8000:].decode()
    last = tail.strip().splitlines()[-1]
    out = json.loads(last)
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0
    detail = json.load(open(os.path.join(str(tmp_path), "gpurun_out", "bench_detail.json")))
    assert detail["extra"]["cfg4"]["time_to_target"]["best_batch"] == 2048  # nothing is lost: the detail keeps everything
    assert len(r.stderr) > 20000


def test_contract_line_of_a_multi_gpu_run():
    """N > 1: the line carries the exchange's statistics (and the quoted progress per epoch), extra holds configs[2] only"""
    sys.path.insert(0, ROOT)
    import bench

    full = canned()
    full["n_gpus"] = 8
    full["cpu_baseline"] = None
    full["exact_order"] = None
    full["time_to_target"] = None
    full["dp"] = {"combine": "mean", "sync_period": 128, "world": 8, "collectives_per_step": 10.0, "bytes_per_step_per_rank": 5200000640.0,
                  "note": "n" * 300, "progress_per_epoch": 1.12, "progress_source": "profiles/r03_dp_convergence.txt"}
    full["extra"] = {"cfg3": full["extra"]["cfg3"]}
    line = bench.contract_line(full)
    assert len(line) < 6000
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["dp"]["world"] == 8 and out["dp"]["progress_per_epoch"] == 1.12 and "note" not in out["dp"]
    assert out["cpu_baseline"] is None and out["exact_order"] is None and list(out["extra"]) == ["cfg3"]
    assert out["roofline"]["frac"] == full["roofline"]["frac"]


@pytest.mark.parametrize("record,batch,frac", [("r05d", 65536, 0.5), ("r05f", 131072, 0.6), ("r05g", 131072, 0.6), ("r05h", 131072, 0.6), ("r05i", 262144, 0.68), ("r05j", 262144, 0.68), ("r05k", 262144, 0.68), ("r05l", 262144, 0.68)])
def test_round5_detail_record_gives_a_complete_line(record, batch, frac):
    """this round's own detail records (profiles/r05d_default_bench_detail.json: headline batch 65536; r05f: 131072, the batch
    bench.py quotes now) through contract_line: under 6 KB, and the fields round 5 added are there -- the headline's value at
    the old batch, the best Hogwild setting and the ratio against it, the exact order's flavours and how far the one-term
    window is from the one-workgroup kernel"""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "%s_default_bench_detail.json" % record)))
    line = bench.contract_line(full)
    assert "\n" not in line and len(line) < 6000, len(line)
    out = json.loads(line)
    assert out["config"]["batch"] == batch and out["value_batch_8192"] and out["value_batch_8192"] < out["value"]
    assert out["roofline"]["frac"] > frac
    if record == "r05l":  # the wording says which of the two rates `value` is
        assert "fixed order (value_shuffled: a fresh order per epoch)" in out["config"]["workload"] and len(line) < 5400
    if record in ("r05k", "r05l"):  # AdaGrad without the stopping criterion's sum beside the figure that tracks it (the figure quoted)
        assert out["extra"]["cfg3"]["value_no_viol"] > 1.15 * out["extra"]["cfg3"]["value"] and "value_no_viol" not in out["extra"]["cfg2"]
    if record in ("r05j", "r05k", "r05l"):  # cfg4 at batch 65536; the roofline leg's figures are those of the median of five epochs
        c4 = out["extra"]["cfg4"]
        assert c4["batch"] == 65536 == bench.WORKLOADS["cfg4"]["batch"] and c4["frac"] > 0.5 and c4["reached"] is True and min(c4["t2t_speedup"]) > 25
        assert "median of 5 epochs" in out["roofline"]["avg_over"] and len(full["roofline"]["pair_ms_epochs"]) == 5
    if record in ("r05h", "r05i"):  # field-aware AdaGrad at batch 32768 with the batch's gradient cross products in g_norm (nfm_opt_set_ada_cross)
        c4 = out["extra"]["cfg4"]
        assert c4["batch"] == 32768 and c4["frac"] > 0.4 and c4["reached"] is True and min(c4["t2t_speedup"]) > 20
        assert bench.WORKLOADS["cfg4"]["ada_cross"] == 0.1
    if record in ("r05g", "r05h", "r05i", "r05j", "r05k", "r05l"):  # (the line says what time_to_target's speed-ups are against)
        assert out["time_to_target"]["speedup_is_vs"] == "gpu exact order" and len(out["time_to_target"]["vs_cpu_port_1_thread"]) == 3
        assert out["value_shuffled"] > 0.86 * out["value"]  # positions of a device-drawn order computed from its key
    if record in ("r05f", "r05g", "r05h"):
        assert "r05f_headline_B131072_pmc_traffic.json" in full["roofline"]["traffic_source"]  # traffic of THIS batch
    if record in ("r05i", "r05j", "r05k", "r05l"):  # the workload table as it stands: the headline at 262144 / cap 32, cfg2 at 65536 / cap 32, PMC of THAT batch
        assert bench.WORKLOADS["headline"]["batch"] == batch and bench.WORKLOADS["headline"]["touch_cap"] == 32.0
        assert bench.WORKLOADS["cfg2"]["batch"] == 65536 == out["extra"]["cfg2"]["batch"] and out["extra"]["cfg2"]["frac"] > 0.65
        assert "r05i_headline_B262144_pmc_traffic.json" in full["roofline"]["traffic_source"] and out["config"]["touch_cap"] == 32.0
        assert [t["epochs"] for t in out["time_to_target"]["targets"]] == [1, 3, 7]
    hog = out["cpu_baseline"]["hogwild"]
    sweep = full["cpu_baseline"]["hogwild"]["sweep"]
    assert hog["value"] == max(e["value"] for e in sweep) and len(sweep) >= 4  # the BEST of the sweep is what the line carries
    assert abs(out["vs_best_cpu"] - out["value"] / max(hog["value"], out["cpu_baseline"]["value"])) < 0.1
    x = out["exact_order"]
    assert x["bit_equal"] is True and 0 < x["max_rel_diff"] < 1e-6 and x["term_by_term"] < x["value"]
    for e in out["extra"].values():
        assert e["exact_bit_equal"] is True and e["exact_max_rel_diff"] < 1e-6
