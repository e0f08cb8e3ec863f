"""`python bench.py --gpus N` with no launcher around it: the parent starts N ranks itself. The
children here are a stub (no GPU, no torch): what is under test is the launcher — ranks, environment, one line on stdout,
exit codes, and that a dead rank takes the waiting ones with it."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / "stub_main.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launcher_starts_n_ranks_and_relays_rank0_line(tmp_path):
    import bench

    cmd = _stub(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        open(os.path.join(os.path.dirname(__file__), f"rank{rank}.json"), "w").write(json.dumps(
            {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                            "HSA_ENABLE_IPC_MODE_LEGACY")} | {"argv": sys.argv[1:]}))
        print(json.dumps({"metric": "stub", "rank": rank, "n_gpus": int(os.environ["WORLD_SIZE"])}))
        print("noise on stderr", file=sys.stderr)
    """)
    out = io.StringIO()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"], child_cmd=cmd, env=env, stdout=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1  # ONE JSON line: rank 0's; rank 1's stdout goes nowhere
    assert json.loads(lines[0]) == {"metric": "stub", "rank": 0, "n_gpus": 2}
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(2)]
    for r, e in enumerate(seen):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "2"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert e["argv"] == ["--gpus", "2", "--steps", "3"]
    assert seen[0]["MASTER_PORT"] == seen[1]["MASTER_PORT"] and int(seen[0]["MASTER_PORT"]) > 0


def test_launcher_reports_a_failed_rank_and_stops_the_others(tmp_path):
    import bench

    cmd = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)  # rank 0 "waits in a collective" for the dead rank
    """)
    out = io.StringIO()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    rc = bench.launch_ranks(2, [], child_cmd=cmd, env=env, stdout=out)
    assert rc == 3 and time.time() - t0 < 60 and out.getvalue().strip() == ""


def test_bench_cli_takes_the_launcher_only_without_world_size(tmp_path):
    """Under torchrun (WORLD_SIZE set) the command behaves as before: no children. Without it, --gpus 2 goes through
    launch_ranks — checked on the real command line with the child command replaced by a stub through a wrapper."""
    wrapper = tmp_path / "run_bench_main.py"
    wrapper.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        calls = []
        bench.launch_ranks = lambda n, argv, **kw: calls.append((n, argv)) or 0
        sys.argv = ["bench.py"] + sys.argv[1:]
        try:
            bench.main()
        except SystemExit as e:
            print("EXIT", e.code, calls)
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(wrapper), "--gpus", "2", "--steps", "5"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert "EXIT 0 [(2, ['--gpus', '2', '--steps', '5'])]" in r.stdout, r.stdout + r.stderr
