"""bench.py's contract with the driver: one JSON line with the agreed keys on a GPU box, a loud failure without a HIP device."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_hip_device():
    """no CPU fallback in the measured path: on a box without a GPU the bench exits non-zero and says why"""
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--no-secondary"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "maps/s" and j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "maze32" in j["config"]["workload"] and j["config"]["batch_per_gpu"] == 4096
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(j["value"] - 4096 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]  # value = units of the K steps / their wall time
    im = j["issue_model"]  # pipe model: what the launch costs the VALU / LDS pipes, and what its longest chain alone costs
    assert 0.0 < im["valu_busy_frac"] < 1.0 and 0.0 < im["lds_busy_frac"] < 1.0 and 0.0 < im["frac_of_serial_floor"] < 1.5
    assert im["instructions_per_step"] == 71 and im["serial_floor_us"] > 0
    # the fraction is quoted on the bytes the VanillaAstar call must move (24 B/cell), the 28 B/cell figure of SURVEY 8(d) beside it; fields read
    # from committed profile files say so in their names
    assert rf["algorithmic_bytes_per_launch"] == 24 * 32 * 32 * 4096 and abs(rf["frac_28B_per_cell"] - rf["frac"] * 28 / 24) < 1e-9
    assert "committed_kernel_profile_us" in rf and "traffic_source" in rf
    # WHICH figure is the top line: what a caller of the reference's forward() signature gets -- VanillaAstar.forward() per step
    # with the default same-call verdict, on never-searched batches that carry their loader's placement hint, the hint's counting sort INSIDE the
    # timed step.  The bare C-ABI launch (order precomputed outside the clock: the headline of rounds 1-5) and the hint-free call stand beside it.
    assert j["config"]["placement"].startswith("dataset") and "INSIDE" in j["config"]["placement"] and j["config"]["every_timed_step_is_a_first_visit"] is True
    assert "VanillaAstar.forward" in j["config"]["step"] and "inside the step" in j["config"]["step"] and "forward() signature" in j["value_is"]
    assert j["config"]["host_lane"].startswith("native")  # lib/_nastar_fastlane.so was built and loaded
    no, bare = j["natural_order"], j["bare_launch"]
    assert no["value"] > 0 and j["value_natural_order"] == no["value"]
    assert bare["dataset_order"]["value"] > 0 and bare["natural_order"]["value"] > 0 and j["value_bare_launch"] == bare["dataset_order"]["value"]
    assert j["value"] <= bare["dataset_order"]["value"] * 1.10  # (the API call cannot beat the bare launch it contains; 10 % for clock noise over 4 steps)
    assert rf["launch_ms_avg"] * 1e-3 <= (j["ms_per_step"] * 1e-3) * 1.25 and 0.0 < rf["frac_of_whole_step"] <= rf["frac"] * 1.25
    ce = j["contract_exact_no_prewarm"]  # the W + K protocol run first, before the untimed pre-warm launches
    assert ce["value"] > 0 and abs(ce["value"] - 4096 * 4 / (ce["ms_per_step"] * 4e-3)) < 1e-6 * ce["value"]
    assert j["config"]["distributed"] == {"initialized": False}


@pytest.mark.gpu
def test_bench_line_survives_its_secondary_experiments():
    """the secondary experiments run in a CHILD process (bench_extras.py) with a time limit: cut off after a few seconds here, the line is still
    printed, says so, and carries the sections the child had finished"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--extras-timeout", "25"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["value"] > 0 and "roofline" in j and "extras_note" in j
    assert "limit" in j["extras_note"] or "rc 0" in j["extras_note"]
    tm = j.get("through_module")  # the first section of the child: done within the limit on any box
    assert tm is None or tm["check_solvable_default_sync"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["warcraft"])
def test_train_bench_prints_one_json_line(config):
    """`bench.py --mode train` (BASELINE config 5; `--config maze` is the other one): one full training step per bench step"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--config", config, "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["unit"] == "maps/s" and j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak" and j["data"] == "synthetic"
    assert j["config"]["batch_per_gpu"] == 100 and config in j["config"]["workload"]
    assert j["roofline"]["bound"] == "mfma" and 0.0 < j["roofline"]["frac"] < 1.0
    assert abs(j["value"] - 100 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    import math
    assert math.isfinite(j["final_loss"])


@pytest.mark.gpu
def test_sharded_step_collates_per_bucket_over_rccl_in_a_one_rank_group():
    """The N > 1 step of bench.py (`--force-collate`: a 1-rank RCCL group on this GPU): the API call writes its bit-packed masks into the
    collation bucket's slot (planner.astar.packed_sink -> packed_out of nastar_forward_ex), one all-gather per `--collate-bucket` steps on a side
    stream, the tail bucket flushed inside the timed region (parallel.BucketedCollator; the world-2 content check runs over gloo on the CPU:
    tests/test_host_logic.py).  5 steps with buckets of 2 = three collectives incl. a partly filled one."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collate", "--collate-bucket", "2", "--steps", "5", "--warmup", "2",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    note = j["config"]["collate"]
    assert "FAILED" not in note and "ONE all-gather per 2 steps" in note and "emitted by the search launch itself" in note, note
    assert j["config"]["distributed"]["backend"] == "nccl" and j["config"]["distributed"]["world_size"] == 1
    assert j["value"] > 0 and j["config"]["host_lane"].startswith("native")
