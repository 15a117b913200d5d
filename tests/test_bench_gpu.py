"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1 and the self-launched N = 2 path (rehearsed with
two ranks sharing the one card over gloo -- RCCL cannot put two ranks on one GPU; the driver's scaling run uses nccl)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--batch", "2", "--clip_len", "16384", "--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--no_fp32", "--no_graph"]


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run(SMALL)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["unit"] == "utterances/s" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("configs[1]") and d["roofline"]["bound"] in ("mfma", "hbm")
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_hip_graph_leg_runs_in_a_child_process_and_reports():
    """The secondary `hip_graph` object: the same step with forward + backward replayed from a captured graph, measured in a
    child process so that a failure there cannot take the headline line down."""
    d = _run([a for a in SMALL if a != "--no_graph"])
    g = d["hip_graph"]
    assert "error" not in g, g
    assert g["unit"] == "utterances/s" and g["value"] > 0 and g["loss"] == g["loss"]


def test_two_ranks_started_by_bench_itself():
    d = _run(["--gpus", "2"] + SMALL, env={"SMT_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["rehearsal_shared_gpu"] and d["backend"] == "gloo"
    assert len(d["ms_per_step_per_rank"]) == 2 and d["grad_sync_exposed_ms_per_step"] is not None
    assert abs(d["ms_per_step"] - max(d["ms_per_step_per_rank"])) < 1e-6          # MAX over ranks
    assert abs(d["value"] - 4 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]    # whole-job utterances/s
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"


def test_transformer_lm_workload_line_and_two_rank_rehearsal():
    """`--workload transformer_lm`: same line format under its own metric; data-parallel ranks keep the
    gradients in the flat all-reduce buffer (two ranks on the one card over gloo)."""
    small = ["--workload", "transformer_lm", "--lm_batch", "2", "--lm_len", "66", "--steps", "2", "--warmup", "1", "--no_cpu_baseline"]
    d = _run(small)
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 66 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert {"lm_attention:fwd", "lm_attention:bwd", "lm_add_ln:fwd", "lm_add_ln:bwd"} <= {k["name"] for k in d["kernels"]}
    d2 = _run(["--gpus", "2"] + small, env={"SMT_BENCH_REHEARSAL": "1"})
    assert d2["n_gpus"] == 2 and d2["rehearsal_shared_gpu"] and d2["config"]["global_batch"] == 4
    assert abs(d2["value"] - 4 * 66 * 1000.0 / d2["ms_per_step"]) < 1e-6 * d2["value"]
    d3 = _run(small + ["--lm_graph"])                       # forward + backward replayed as one captured hipGraph
    assert d3["hip_graph"] and d3["value"] > 0 and d3["loss"] == d3["loss"]


def test_aux_workload_reports_the_f_rows():
    """`--workload aux`: encode-only pass, STFT.inverse and maximum_path, each with a value, a unit and (where it has a
    byte count) a roofline object."""
    d = _run(["--workload", "aux", "--batch", "2", "--clip_len", "16384", "--steps", "2", "--warmup", "1", "--no_cpu_baseline"])
    assert d["encode_only"]["unit"] == "utterances/s" and d["encode_only"]["value"] > 0
    assert d["stft_inverse"]["roofline"]["bound"] == "hbm" and 0 < d["stft_inverse"]["roofline"]["frac"] < 1
    assert d["maximum_path"]["unit"] == "cells/s" and d["maximum_path"]["value"] > 0 and d["vs_baseline"] is None
