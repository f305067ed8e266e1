"""`python bench.py --gpus N` outside a launcher starts its own ranks: the parent makes no GPU call, starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process and relays rank 0's line.
Here (no GPU): the dry run prints the command and the environment, and the relay path is run end to end against a stand-in child."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dry_run_shows_a_child_torchrun_with_the_same_arguments():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-run-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "2"]            # the ranks see the same arguments, minus the dry-run switch
    assert d["ranks"] == 8 and d["parent_touched_gpu"] is False
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_under_a_launcher_the_script_does_not_launch_again():
    """WORLD_SIZE set (torch.distributed.run's environment): no second launcher; a world size that contradicts --gpus is an error"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-launch"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE is 2" in out.stderr and "launch" not in out.stdout


def test_relay_passes_the_line_and_the_exit_code(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "fake_python.sh"
    fake.write_text("#!/bin/sh\necho '{\"metric\": \"x\", \"n_gpus\": 2}'\nexit 0\n")
    fake.chmod(0o755)
    monkeypatch.setattr(sys, "executable", str(fake))
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    fake.write_text("#!/bin/sh\necho rank 1 died >&2\nexit 3\n")
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 3
    fake.write_text("#!/bin/sh\nexit 0\n")                                    # clean exit without a line is still a failure
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 1


def test_a_killed_parent_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to `python bench.py --gpus N` (the driver's timeout) must not leave the torchrun child and its ranks behind.
    A stand-in child that writes its pid and sleeps; the parent is terminated; the child must be gone within the grace period."""
    import signal
    import time
    pidfile = tmp_path / "child.pid"
    fake = tmp_path / "fake_python.sh"
    fake.write_text(f"#!/bin/sh\necho $$ > {pidfile}\nsleep 300\n")
    fake.chmod(0o755)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.executable = {str(fake)!r}; "
            "raise SystemExit(bench.launch_ranks(2, ['--gpus', '2']))")
    parent = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for _ in range(600):
        if pidfile.exists() and pidfile.read_text().strip():
            break
        time.sleep(0.1)
    child_pid = int(pidfile.read_text())
    os.kill(child_pid, 0)                                   # alive
    parent.send_signal(signal.SIGTERM)
    parent.wait(timeout=60)
    assert parent.returncode == 128 + signal.SIGTERM
    for _ in range(150):
        try:
            os.kill(child_pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.kill(child_pid, signal.SIGKILL)
        raise AssertionError("the launcher's child outlived the parent")


def test_counter_summaries_are_attached_only_for_the_source_and_workload_they_were_measured_on(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` / `pmc` / `pmc_source` are not measured in the run (counters need rocprofv3 passes of their own): they are read
    from profiles/rNN/*.json and attached ONLY while the kernel source is byte-identical (SHA-1) to the one the passes ran on and the workload is
    BASELINE's (batch 64, 10 frames, 512 audio tokens, ViT-B, bf16) - otherwise null, never a stale number."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    head = argparse.Namespace(batch=64, frames=10, audio_tokens=512, model="vit_base", fp8=False, recompute=None)
    sha = bench._sha1(os.path.join(ROOT, "avsiam_amd", "csrc", "gemm.hip"))
    rounds = [r for r in ("r06", "r05", "r04", "r03", "r02") if os.path.exists(os.path.join(ROOT, "profiles", r, "traffic.json"))]      # newest first, as bench.py looks
    match = [r for r in rounds if json.load(open(os.path.join(ROOT, "profiles", r, "traffic.json")))["source_sha1"]["gemm.hip"] == sha]
    if match:
        # a committed summary belongs to the kernel source as it is: the line carries the NEWEST such one, with its provenance
        t = bench.pmc_traffic(head)
        assert t is not None and 6e8 < t < 1.2e9                                   # ~0.81 GB per forward/dgrad GEMM launch (558 MB algorithmic; 0.87 / 613 in round 5)
        src = bench.pmc_source(head)
        assert set(src) == {"traffic.json", "pmc_busy.json"}
        assert all(v["kernel_source_sha1"] == sha and v["file"].startswith("profiles/%s/" % match[0]) for v in src.values())
    else:
        # gemm.hip was edited after the passes ran: the line must carry null until tools/round6_profile.sh pmc is re-run
        assert bench.pmc_traffic(head) is None and not bench.pmc_source(head)
    # another workload: nothing attached
    for kw in ({"batch": 4}, {"frames": 1}, {"model": "vit_large"}, {"fp8": True}, {"recompute": "auto"}):
        other = argparse.Namespace(**{**vars(head), **kw})
        assert bench.pmc_traffic(other) is None and not bench.pmc_source(other)
    # another kernel source: nothing attached (a copy of the tree's profiles with a foreign SHA-1)
    fake_root = tmp_path / "repo"
    (fake_root / "profiles" / "r05").mkdir(parents=True)
    (fake_root / "avsiam_amd" / "csrc").mkdir(parents=True)
    (fake_root / "avsiam_amd" / "csrc" / "gemm.hip").write_text("// edited kernel\n")
    for name in ("traffic.json", "pmc_busy.json"):
        (fake_root / "profiles" / "r05" / name).write_text(open(os.path.join(ROOT, "profiles", "r05", name)).read())
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    assert bench.pmc_traffic(head) is None and not bench.pmc_source(head)
