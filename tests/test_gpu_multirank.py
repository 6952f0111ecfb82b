"""The HIP path under more than one rank, rehearsed on ONE GPU: two fresh child processes share cuda:0 (backend gloo,
DCLL_DIST_BACKEND — RCCL refuses two ranks on one device), each runs the real network on its shard_range of the batch.
What 8(e) promises is asserted: sharding changes no per-sample result, the all-reduced tallies equal the single-process
tallies, and a local-learning run with the per-timestep gradient all-reduce equals the full-batch run.
Also: `bench.py --gpus 2` started plainly launches its own ranks and prints one JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(out_dir, world, B, T, timeout=900):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        r = subprocess.run([sys.executable, WORKER, out_dir, str(B), str(T)], env=base, timeout=timeout,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        return
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DCLL_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, WORKER, out_dir, str(B), str(T)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, se[-3000:])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_one_gpu_equal_the_single_process_run(tmp_path, world):
    """Two / four child ranks sharing cuda:0 (gloo rehearsal; with the test process itself that is 3 / 5 processes on the
    card, inside the box's limit of 6): evaluation results concatenate to the single-process run bit for bit, the all-reduced
    tallies are the single-process tallies on every rank, and a 7-step learning run with the per-timestep gradient slabs
    equals the full-batch run on every rank."""
    B, T = 1024, 128
    out = str(tmp_path)
    _run_ranks(out, 1, B, T)
    _run_ranks(out, world, B, T)
    full = np.load(os.path.join(out, "rank_0_of_1.npz"))
    parts = [np.load(os.path.join(out, "rank_%d_of_%d.npz" % (r, world))) for r in range(world)]
    per = B // world
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == [(r * per, (r + 1) * per) for r in range(world)]
    # evaluation: concatenated per-rank results == the single-process run, bit for bit; tallies equal on every rank
    for i in range(3):
        for key in ("clout%d", "host_clout%d"):
            assert np.array_equal(np.concatenate([p[key % i] for p in parts], axis=1), full[key % i]), key % i
        assert np.array_equal(np.concatenate([p["vote%d" % i] for p in parts]), full["vote%d" % i])
    assert np.array_equal(np.concatenate([p["o"] for p in parts], axis=1).view(np.uint32), full["o"].view(np.uint32))
    for p in parts:
        assert np.array_equal(p["tallies"], full["tallies"])
    assert int(full["tallies"][0, -1]) == B and full["tallies"].shape == (3, 24 * 24 + 2)
    # learning: all ranks hold the same parameters, and they equal the full-batch run (sums in another order)
    for key in ["w0", "w1", "w2", "b0", "b1", "b2", "ow", "ob"]:
        for p in parts[1:]:
            assert np.array_equal(parts[0][key], p[key]), key
    for key in ["g0_w0", "g0_w1", "g0_w2", "g0_b0", "g0_b1", "g0_b2", "g0_ow"]:
        for p in parts[1:]:
            assert np.array_equal(parts[0][key], p[key]), key
        ref = full[key]
        np.testing.assert_allclose(parts[0][key], ref, rtol=2e-3, atol=1e-5 * np.abs(ref).max(), err_msg=key)
    for key in ["w0", "w1", "w2", "b0", "b1", "b2", "ow", "ob"]:
        ref = full[key]
        np.testing.assert_allclose(parts[0][key], ref, rtol=0, atol=2e-3 * np.abs(ref).max(), err_msg=key)


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent spawns two ranks before touching the GPU and
    relays rank 0's single JSON line (on this 1-GPU box the ranks share the device and fall to gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "256"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 512
    assert out["value"] > 0 and out["scaling"] == "weak" and "roofline" in out
    assert "cpu_baseline" not in out            # rank 0 at N = 1 only


@pytest.mark.timeout(1200)
def test_bench_strong_scaling_flag_fixes_the_global_batch():
    """`bench.py --gpus N --global-batch G` (north_star: "batch 4096 ... at 1/2/4/8 GPUs" read as a FIXED global batch): the
    ranks run contiguous shards of the SAME G windows, the line says "scaling": "strong", value = global windows / time,
    the upload-inclusive value sits beside the device-resident one — and the all-reduced tallies (correct / total per
    layer, digest of the confusion matrices) of a FOUR-rank run (four processes sharing cuda:0, gloo rehearsal; ragged
    shards of 257/257/257/256 windows) equal the single-process run's.  The N > 1 line carries what makes a real 8-GPU run
    diagnosable: per-rank step time (min / max), the tally all-reduce's time, device-busy time, ranks seen, backend."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    lines = {}
    for n in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1",
                            "--global-batch", "1027", "--cpu-windows", "0", "--per-step", "0", "--config5", "0",
                            "--live-traffic", "0", "--batch-sweep", "0", "--plane128", "0", "--trained", "0", "--t1024", "0"],
                           env=env, capture_output=True, text=True, timeout=1000)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(out) == 1, r.stdout
        lines[n] = json.loads(out[0])
    one, four = lines[1], lines[4]
    assert four["n_gpus"] == 4 and four["scaling"] == "strong" == one["scaling"]
    assert four["config"]["global_batch"] == 1027 and four["config"]["batch_per_gpu"] == 257
    assert abs(four["value"] - 1027 / (four["ms_per_step"] * 1e-3)) < 1e-6 * four["value"]
    assert 0 < four["value_incl_upload"] <= four["value"] * 1.5 and four["ms_per_step_incl_upload"] > 0
    assert four["tallies"] == one["tallies"], (four["tallies"], one["tallies"])
    assert [ct[1] for ct in four["tallies"]["correct_total"]] == [1027] * 3
    assert four["vote_accuracy_vs_random_labels"] == one["vote_accuracy_vs_random_labels"]
    m = four["multi_gpu"]
    assert "multi_gpu" not in one and m["ranks_seen"] == 4 and m["backend"] == "gloo"
    assert 0 < m["per_rank_ms_per_step"]["min"] <= m["per_rank_ms_per_step"]["max"] <= four["ms_per_step"] * (1 + 1e-9)
    assert 0 < m["allreduce_ms_per_step"]["min"] <= m["allreduce_ms_per_step"]["max"]
    assert 0 < m["device_busy_ms_per_step"]["min"] <= m["device_busy_ms_per_step"]["max"]
    assert len(m["rank_devices"]) == 4 and m["distinct_devices"] == 1         # the rehearsal: four ranks, one GPU


def _no_process_carries(mark):
    """No live process of ours still has DCLL_TEST_MARK=<mark> in its environment (the ranks and their launcher are gone)."""
    left = []
    for pid in os.listdir("/proc"):
        if not pid.isdigit() or int(pid) == os.getpid():
            continue
        try:
            with open("/proc/%s/environ" % pid, "rb") as f:
                if ("DCLL_TEST_MARK=%s" % mark).encode() in f.read():
                    left.append(int(pid))
        except OSError:
            continue
    return left


@pytest.mark.timeout(900)
def test_bench_under_the_drivers_torchrun_command():
    """The driver's multi-GPU command, verbatim — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — rehearsed on the one GPU with N = 4 (the box admits six
    processes on the card: four ranks + this test; the EIGHT-rank form of the same command runs on CPU in
    tests/test_parallel_gloo.py::test_eight_ranks_under_the_drivers_exact_command_on_cpu), DCLL_DIST_BACKEND=gloo:
      * the default (weak-scaling) line: one JSON line on stdout, n_gpus 4, all four ranks seen on ONE distinct device and the
        line says REHEARSAL, `roofline` from rank 0's kernel events, every N = 1-only extra absent, exit code 0, no rank or
        launcher process left behind;
      * north_star's literal "batch 4096 ... at 1/2/4/8 GPUs": `--global-batch 4096` over the four ranks (1 024 windows each)
        -> the all-reduced tallies equal the single-process run's digest."""
    import uuid
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    bench = os.path.join(ROOT, "bench.py")
    quiet = ["--cpu-windows", "0", "--per-step", "0", "--config5", "0", "--live-traffic", "0", "--batch-sweep", "0", "--plane128", "0",
             "--trained", "0", "--t1024", "0"]

    def torchrun(n, extra):
        mark = uuid.uuid4().hex
        env = dict(base, DCLL_DIST_BACKEND="gloo", DCLL_TEST_MARK=mark, OMP_NUM_THREADS="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), bench, "--gpus", str(n)] + extra,
                           env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert len(out) == 1, r.stdout
        assert _no_process_carries(mark) == []
        return json.loads(out[0])
    line = torchrun(4, ["--steps", "1", "--warmup", "0", "--batch", "64", "--live-traffic", "0"])
    m = line["multi_gpu"]
    assert line["n_gpus"] == 4 and line["steps"] == 1 and line["warmup"] == 0 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 256 and line["config"]["batch_per_gpu"] == 64
    assert m["ranks_seen"] == 4 and m["backend"] == "gloo" and m["distinct_devices"] == 1 and len(m["rank_devices"]) == 4
    assert "REHEARSAL" in line["config"]["parallelism"]
    assert line["value"] > 0 and abs(line["value"] - 256 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    roof = line["roofline"]
    assert roof["kernel"] == "k_lif_seq_c32d" and roof["launches"] == 2 and 0 < roof["frac"] < 1.2 and roof["avg_launch_ms"] > 0
    for extra in ("cpu_baseline", "speedup_vs_cpu_baseline", "per_step_paths", "config5", "t1024", "batch_sweep", "trained_top1"):
        assert extra not in line, extra
    four = torchrun(4, ["--steps", "1", "--warmup", "0", "--global-batch", "4096"] + quiet)
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "1", "--warmup", "0", "--global-batch", "4096"] + quiet,
                       env=base, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    one = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert four["scaling"] == "strong" and four["config"]["batch_per_gpu"] == 1024 and four["config"]["global_batch"] == 4096
    assert four["tallies"] == one["tallies"] and [ct[1] for ct in four["tallies"]["correct_total"]] == [4096] * 3


@pytest.mark.timeout(900)
def test_entry_points_shard_over_ranks(tmp_path):
    """`test_radio_ml.py --gpus 2` and `train.py --gpus 2` started plainly: each launches two ranks (sharing cuda:0 here,
    gloo), shards every batch and reduces — the per-SNR accuracies and confusion matrices equal the single-process
    evaluation exactly, the trained parameters equal the single-process training within the G6 tolerance."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    common = ["--I_resolution", "16", "--Q_resolution", "16", "--arp", "1.0", "--burnin", "4", "--batch_size_test", "24",
              "--n_test_samples", "48", "--synthetic", "48", "--n_iters_test", "24", "--min_snr", "10", "--max_snr", "12"]
    outs = {}
    for n in (1, 2):
        out = tmp_path / ("eval%d" % n)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "test_radio_ml.py"), "--gpus", str(n), "--out_dir", str(out)]
                           + common, env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[n] = (np.load(out / "snr_evaluation_accs.npy"), np.load(out / "confusion_matrix_snr_10.npy"),
                   [l for l in r.stdout.splitlines() if l.startswith("SNR")])
    assert outs[1][0].shape == (2, 3) and np.array_equal(outs[1][0], outs[2][0])
    assert np.array_equal(outs[1][1], outs[2][1]) and outs[1][1].sum() == 48
    assert len(outs[2][2]) == 2                                   # rank 0 alone reports
    params = {}
    for n in (1, 2):
        res = tmp_path / ("train%d" % n)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--gpus", str(n), "--output", str(res),
                            "--batch_size", "24", "--n_steps", "2", "--n_iters", "12", "--n_test_interval", "1",
                            "--learning_rates", "1e-7"] + common, env=env, capture_output=True, text=True, timeout=400,
                           cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        runs = list((res / "RadioML").iterdir())
        assert len(runs) == 1
        params[n] = torch.load(runs[0] / "parameters_1.pth")
    for k, v in params[1].items():
        a, b = v.numpy(), params[2][k].numpy()
        if "i2o" in k or "alpha" in k or "tau" in k:
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(b, a, rtol=0, atol=2e-3 * np.abs(a).max(), err_msg=k)


@pytest.mark.timeout(1200)
def test_rccl_code_path_on_a_one_rank_group_equals_the_plain_run(tmp_path):
    """The RCCL branch of parallel.py on the 1-GPU box: ONE fresh child with WORLD_SIZE=1, DCLL_FORCE_DIST=1 and the
    default backend (nccl == RCCL) forms a communicator with `device_id` and sends every exchange of the evaluation and
    of a local-learning run through it on device tensors — int64 tallies, the fp32 gradient bucket per timestep, barrier,
    destroy_process_group.  Everything it computes equals the plain single-process run BIT FOR BIT.  Then `bench.py
    --gpus 1` under the same switch: additionally the fp64 ReduceOp.MAX of the step time; same votes as the plain run."""
    B, T = 256, 64
    plain, forced_dir = tmp_path / "plain", tmp_path / "rccl"
    plain.mkdir()
    forced_dir.mkdir()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DCLL_DIST_BACKEND",
                                                              "DCLL_FORCE_DIST")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rccl = dict(base, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                DCLL_FORCE_DIST="1")
    for env, out in ((base, plain), (rccl, forced_dir)):
        r = subprocess.run([sys.executable, WORKER, str(out), str(B), str(T)], env=env, timeout=900, capture_output=True,
                           text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(plain / "rank_0_of_1.npz"), np.load(forced_dir / "rank_0_of_1.npz")
    assert str(a["backend"]) == "none" and str(b["backend"]) == "nccl"
    assert sorted(a.files) == sorted(b.files)
    for key in a.files:
        if key == "backend":
            continue
        x, y = a[key], b[key]
        assert x.shape == y.shape and x.dtype == y.dtype and x.tobytes() == y.tobytes(), key
    assert int(b["tallies"][0, -1]) == B
    lines = {}
    for name, env in (("plain", base), ("rccl", dict(rccl, MASTER_PORT=str(_free_port())))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                            "--batch", "256", "--cpu-windows", "0", "--per-step", "0", "--config5", "0", "--live-traffic", "0"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(out) == 1, r.stdout
        lines[name] = json.loads(out[0])
    assert "backend nccl" in lines["rccl"]["config"]["parallelism"] and "backend none" in lines["plain"]["config"]["parallelism"]
    assert lines["rccl"]["n_gpus"] == 1 and lines["rccl"]["value"] > 0
    assert lines["rccl"]["vote_accuracy_vs_random_labels"] == lines["plain"]["vote_accuracy_vs_random_labels"]


@pytest.mark.gpu
def test_launcher_counts_the_gpus_without_a_runtime():
    """spawn_local_ranks decides "rehearsal: ranks share devices" from parallel.visible_gpu_count(), which must not bring
    the HIP runtime up in the launcher (DRM render nodes + *_VISIBLE_DEVICES): it has to agree with what a rank sees."""
    from snn_modulation_classification_amd import parallel
    assert parallel.visible_gpu_count() == torch.cuda.device_count()
