"""`python bench.py --gpus N` starts its own rank processes: launcher logic on CPU."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = ("import os, json, sys; e = os.environ; "
         "print(json.dumps({k: e[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}), "
         "file=open(os.path.join(sys.argv[1], 'r' + e['RANK'] + '.json'), 'w'))")


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    from smt_amd import launcher
    rc = launcher.spawn_ranks(3, [sys.executable, "-c", PROBE, str(tmp_path)])
    assert rc == 0
    seen = [json.load(open(tmp_path / f"r{r}.json")) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] == [s["LOCAL_RANK"] for s in seen]
    assert {s["WORLD_SIZE"] for s in seen} == {"3"} and {s["MASTER_ADDR"] for s in seen} == {"127.0.0.1"}
    assert len({s["MASTER_PORT"] for s in seen}) == 1


def test_a_failing_rank_stops_the_others_and_sets_the_exit_code():
    from smt_amd import launcher
    code = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"
    import time
    t0 = time.time()
    rc = launcher.spawn_ranks(2, [sys.executable, "-c", code], grace_s=5.0)
    assert rc == 7 and time.time() - t0 < 30


def test_gloo_ranks_started_by_the_launcher_rendezvous(tmp_path):
    """The children really form a process group from the environment the launcher hands them (gloo on CPU)."""
    from smt_amd import launcher
    code = ("import os, sys, torch, torch.distributed as dist; dist.init_process_group('gloo', init_method='env://'); "
            "t = torch.tensor([float(dist.get_rank() + 1)]); dist.all_reduce(t); "
            "open(os.path.join(sys.argv[1], 'w%d' % dist.get_rank()), 'w').write('%d %g' % (dist.get_world_size(), t.item())); "
            "dist.destroy_process_group()")
    assert launcher.spawn_ranks(2, [sys.executable, "-c", code, str(tmp_path)]) == 0
    assert open(tmp_path / "w0").read() == "2 3" == open(tmp_path / "w1").read()


def test_bench_refuses_more_ranks_than_visible_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr


def test_bench_does_not_relaunch_under_torchrun(monkeypatch):
    import bench
    from smt_amd import launcher
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2")
    assert launcher.under_launcher()
    assert bench.launch_or_none(bench.parse(["--gpus", "2"]), ["--gpus", "2"]) is None
    monkeypatch.delenv("RANK"); monkeypatch.delenv("WORLD_SIZE")
    assert bench.launch_or_none(bench.parse(["--gpus", "1"]), ["--gpus", "1"]) is None
