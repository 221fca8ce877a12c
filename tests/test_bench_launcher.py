"""bench.py --gpus N without a launcher: the parent's argument / environment plumbing (CPU only, no GPU process).
The reference is single-device (yolo_toy_regressor_torch.py:375); the N > 1 launch contract is the build's own."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_environments_are_what_a_launcher_exports():
    envs = bench.rank_environments(4, 29871, base={"PATH": "/bin", "GPU_MAX_HW_QUEUES": "4"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29871" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)


def test_wants_parent_only_without_a_launcher(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench._wants_parent(["--gpus", "8", "--steps", "3"]) == 8
    assert bench._wants_parent(["--gpus=2"]) == 2
    assert bench._wants_parent(["--steps", "3"]) == 0
    assert bench._wants_parent(["--gpus", "1"]) == 0
    monkeypatch.setenv("WORLD_SIZE", "8")          # torch.distributed.run already made the ranks
    assert bench._wants_parent(["--gpus", "8"]) == 0


def _run_parent(tmp_path, body, n=3):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(body))
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(%d, ['--gpus', '%d', '--steps', '2'], child=[sys.executable, %r], watchdog_s=20))"
            % (ROOT, n, n, str(child)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_parent_relays_rank_zero_and_passes_the_arguments(tmp_path):
    r = _run_parent(tmp_path, """
        import json, os, sys
        rec = {"rank": os.environ["RANK"], "world": os.environ["WORLD_SIZE"], "argv": sys.argv[1:],
               "addr": os.environ["MASTER_ADDR"], "port": os.environ["MASTER_PORT"]}
        print(json.dumps(rec))        # only rank 0's stdout reaches the parent's stdout
    """)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["rank"] == "0" and rec["world"] == "3" and rec["addr"] == "127.0.0.1" and int(rec["port"]) > 0
    assert rec["argv"] == ["--gpus", "3", "--steps", "2"]


def test_parent_fails_when_a_rank_fails_and_stops_the_others(tmp_path):
    r = _run_parent(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                 # a rank stuck in a collective its peer never entered
    """)
    assert r.returncode == 7
    assert "rank 1 exited with 7" in r.stderr


def test_pmc_fields_come_from_the_committed_summaries():
    """roofline.traffic / mfma_busy_pmc are read from the newest profiles/rNN summary (its "#json" line) and carry the
    file's git blob hash, so that a number typed in by hand or a stale file shows in the record."""
    vals, src = bench.committed_pmc("pmc_traffic_hbm_kernels.txt")
    assert set(vals) == {"nms", "loss", "assign"} and all(v > 1e6 for v in vals.values())
    assert src["file"].startswith("profiles/r") and len(src["git_blob"]) == 40
    if os.path.isdir(os.path.join(ROOT, ".git")):
        out = subprocess.run(["git", "hash-object", src["file"]], cwd=ROOT, capture_output=True, text=True)
        if out.returncode == 0:
            assert out.stdout.strip() == src["git_blob"]
    kern, ksrc = bench.committed_pmc("conv_pmc_summary.txt")
    assert any(k.startswith("igemm8_kernel") for k in kern) and all(0.0 < v["mfma_busy"] < 1.0 for v in kern.values())
    assert bench.committed_pmc("no_such_summary.txt") == (None, None)
