"""bench.py's own N-rank launch (no GPU needed): `python bench.py --gpus N` without a launcher must start N ranks with
the torch.distributed.run environment, relay rank 0's JSON line, and fail when a rank fails or the line reports another
world size.  The parent imports neither torch nor numpy (it may never initialise the GPU)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

STUB = textwrap.dedent("""
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
    mode = sys.argv[1]
    with open(os.path.join(sys.argv[2], "seen%d" % rank), "w") as f:
        f.write(" ".join(sys.argv[3:]))
    if mode == "fail" and rank == 1:
        for i in range(60):
            sys.stderr.write("rank1 diagnostic line %d\\n" % i)
        sys.stderr.flush()
        sys.exit(3)
    if mode == "hang":
        time.sleep(60)   # e.g. a rank stuck in RCCL initialisation: the parent's timeout must end the run
    if mode == "fail":
        time.sleep(60)   # a rank that would hang on the dead peer's collective: the parent must terminate it
    if rank == 0:
        print("some log line")
        print(json.dumps({"metric": "stub", "n_gpus": (world if mode != "lie" else 1), "value": 1.0}))
""")


def _run(tmp_path, mode, n, capsys, timeout=120):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    rc = bench.spawn_ranks(n, ["--gpus", str(n), "--steps", "2"], worker=[sys.executable, str(stub), mode, str(tmp_path)], timeout=timeout)
    return rc, capsys.readouterr()


def test_spawn_relays_rank0_line(tmp_path, capsys):
    rc, io = _run(tmp_path, "ok", 3, capsys)
    assert rc == 0
    lines = [l for l in io.out.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly ONE JSON line reaches the driver
    assert json.loads(lines[0])["n_gpus"] == 3
    for r in range(3):                     # every rank got the same argv
        assert (tmp_path / ("seen%d" % r)).read_text() == "--gpus 3 --steps 2"


def test_spawn_fails_when_a_rank_fails(tmp_path, capsys):
    rc, io = _run(tmp_path, "fail", 2, capsys)
    assert rc != 0 and io.out.strip() == ""
    assert "rank 1 failed" in io.err


def test_spawn_relays_the_failing_ranks_last_stderr_lines(tmp_path, capsys):
    rc, io = _run(tmp_path, "fail", 2, capsys)
    assert rc != 0
    relayed = [l for l in io.err.splitlines() if l.startswith("  [rank 1] ")]
    assert len(relayed) == 40                                   # the LAST 40 of the 60 lines the rank wrote
    assert relayed[0].endswith("diagnostic line 20") and relayed[-1].endswith("diagnostic line 59")


def test_spawn_times_out_instead_of_hanging(tmp_path, capsys):
    import time
    t0 = time.time()
    rc, io = _run(tmp_path, "hang", 2, capsys, timeout=3)
    assert rc != 0 and io.out.strip() == ""
    assert time.time() - t0 < 30                                # the ranks were terminated by PID, not waited for
    assert "no exit within 3 s" in io.err and "ranks still running: [0, 1]" in io.err


def test_spawn_has_a_default_timeout():
    import inspect
    assert inspect.signature(bench.spawn_ranks).parameters["timeout"].default == bench.DEFAULT_RANK_TIMEOUT_S == 900.0
    assert bench.parse_args(["--gpus", "2"]).rank_timeout == 900.0
    assert bench.parse_args(["--gpus", "2", "--rank-timeout", "60"]).rank_timeout == 60.0


def test_spawn_rejects_a_line_with_another_world_size(tmp_path, capsys):
    rc, io = _run(tmp_path, "lie", 2, capsys)
    assert rc != 0 and io.out.strip() == ""


def test_parent_does_not_import_torch():
    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; "
            "bench.spawn_ranks = lambda n, argv, **k: (print('torch' in sys.modules, 'numpy' in sys.modules, n), 0)[1]; bench.main()")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["False", "False", "2"]


def test_worker_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
