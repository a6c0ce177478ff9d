"""CPU: the host-side driver logic added around the training step -- the N-rank launch of
bench.py, per-rank latent streams, the reference's label draw (new_betavaegan.py:89-90) and epoch
bookkeeping (:196-201), the global-batch divisor a ragged data-parallel tail needs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------ bench.py --gpus N starts N ranks
def test_bench_gpus_flag_launches_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["ranks_counted_by_all_reduce"] == 2
    assert res["config"]["global_batch"] == 256            # weak scaling: 128 per rank


def test_bench_rejects_a_world_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2" in (out.stderr + out.stdout)


def test_bench_reads_the_committed_profile_figures():
    """The two figures bench.py takes from committed profiles rather than from the run itself: the HBM bytes per launch
    of the dominant kernel (PMC passes) and the on-box MFMA rate on random operands (scripts/mfma_peak.hip)."""
    sys.path.insert(0, ROOT)
    import bench
    rate, src = bench.measured_mfma_rate()
    assert src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src))
    assert 1000.0 < rate < bench.PEAK_BF16_MFMA_TFLOPS          # a measured rate, below the nominal dense peak
    traffic, tsrc = bench.pmc_traffic(("conv_fwd", 128, 128, 32, 32, 256, 2), "bf16x6")
    assert traffic >= 103.9e6 and os.path.exists(os.path.join(ROOT, tsrc))      # at least the algorithmic bytes (x + y + w)


# ------------------------------------------------------------------ per-rank latent streams
def _latent_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
    tr = BetaVAEGANTrainer(device="cpu", seed=999)
    assert tr.world == world and tr.rank == rank and tr.dp
    w = torch.cat([p.detach().flatten()[:64] for p in list(tr.netEG.parameters()) + list(tr.netD.parameters())])
    perm = torch.randperm(1000)                  # the CPU generator stays shared: the loader's permutation
    # numpy: pickled by value (torch tensors travel as shared-memory handles that die with the sender)
    q.put((rank, w.numpy(), tr.draw_latents(4).numpy(), tr.draw_latents(4).numpy(), perm.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_share_weights_and_permutation_but_not_latents():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_latent_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, [torch.from_numpy(x) for x in rest]) for r, *rest in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (w0, a0, b0, perm0), (w1, a1, b1, perm1) = got[0], got[1]
    assert torch.equal(w0, w1)                               # replicas start identical (no broadcast needed)
    assert torch.equal(perm0, perm1)
    assert not torch.equal(a0, a1) and not torch.equal(b0, b1)    # each image of the global batch its own draw
    assert not torch.equal(a0, b0)                           # noise / eps2 / eps3 are successive draws
    # reproducible: the stream is a function of (seed, rank) only
    from disentangle_mlp_amd.trainer import _latent_generator
    g = _latent_generator(torch.device("cpu"), 999, 1)
    assert torch.equal(torch.randn(4, 128, generator=g), a1)


# ------------------------------------------------------------------ labels (new_betavaegan.py:89-90)
def test_sample_labels_follows_the_reference_draw():
    from disentangle_mlp_amd.trainer import sample_labels
    rs, ref = np.random.RandomState(7), np.random.RandomState(7)
    for _ in range(200):
        real, fake = sample_labels(rs)
        fake_ref = ref.choice(a=[0.1, 0.9], p=[0.95, 0.05])         # the reference draws the fake label first
        real_ref = ref.choice(a=[0.1, 0.9], p=[0.05, 0.95])
        assert (real, fake) == (real_ref, fake_ref)
    np.random.seed(11)
    a = [sample_labels() for _ in range(5)]                          # default: NumPy's global stream
    np.random.seed(11)
    b = [(lambda f, r: (r, f))(np.random.choice(a=[0.1, 0.9], p=[0.95, 0.05]),
                               np.random.choice(a=[0.1, 0.9], p=[0.05, 0.95])) for _ in range(5)]
    assert a == b
    rs = np.random.RandomState(3)
    draws = np.array([sample_labels(rs) for _ in range(20000)])
    assert abs((draws[:, 0] == 0.1).mean() - 0.05) < 0.01            # 5 % flipped real labels
    assert abs((draws[:, 1] == 0.9).mean() - 0.05) < 0.01            # 5 % flipped fake labels
    assert set(np.unique(draws)) == {0.1, 0.9}


def _label_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer, GANTrainer
    np.random.seed(1000 + rank)                  # whatever the ranks' global NumPy streams hold must not matter
    seen = {}
    for name, tr in (("vaegan", BetaVAEGANTrainer(device="cpu", seed=999)), ("gan", GANTrainer(device="cpu", seed=999))):
        calls = []

        def fake_step(data, real_label=None, fake_label=None, global_batch=None, _calls=calls, **kw):
            _calls.append((real_label, fake_label, global_batch))
            z = torch.tensor(0.0)
            return {"mse_enc": z, "D_x_sum": z, "errG": z, "errD_real": z, "errD_fake": z}
        tr.step = fake_step
        loader = _Loader([torch.zeros(4, 3, 64, 64)] * 200, n=800 * world)
        loader.last_global_batch = None
        tr.train_epoch(loader)
        seen[name] = calls
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_draw_the_same_labels_in_train_epoch():
    """One label pair per GLOBAL batch (the reference is one process: new_betavaegan.py:89-90 / new_gan.py:68-69):
    without an explicit ``label_rng`` every rank of a data-parallel run must draw the same soft / flipped labels."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_label_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for name in ("vaegan", "gan"):
        a, b = got[0][name], got[1][name]
        assert len(a) == len(b) == 200 and a == b
        labels = {(r, f) for r, f, _ in a}
        assert len(labels) > 1                                   # 200 draws at 5 % flips: not all (0.9, 0.1)
        assert all(gb == 4 for _, _, gb in a)                    # _Loader publishes its own global batch


# ------------------------------------------------------------------ epoch bookkeeping (:196-201)
class _Loader:
    def __init__(self, batches, n):
        self.batches, self.dataset = batches, list(range(n))
        self.last_global_batch = None

    def __iter__(self):
        for b in self.batches:
            self.last_global_batch = b.size(0)
            yield b, None


def test_train_epoch_returns_the_reference_averages():
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
    tr = BetaVAEGANTrainer(device="cpu")
    calls = []

    def fake_step(data, real_label=None, fake_label=None, global_batch=None, **kw):
        i = len(calls)
        calls.append((data.size(0), real_label, fake_label, global_batch))
        return {"mse_enc": torch.tensor(100.0 * (i + 1)), "D_x_sum": torch.tensor(0.5 * data.size(0))}
    tr.step = fake_step
    loader = _Loader([torch.zeros(8, 3, 64, 64), torch.zeros(8, 3, 64, 64), torch.zeros(5, 3, 64, 64)], n=21)
    rs = np.random.RandomState(5)
    enc, dec, dis, dx = tr.train_epoch(loader, label_rng=rs)
    # sums of the encoder-phase MSE / of the per-batch MEAN of D(x), both over len(dataset)
    assert enc == dec == pytest.approx((100 + 200 + 300) / 21)
    assert dis == dx == pytest.approx(3 * 0.5 / 21)
    ref = np.random.RandomState(5)
    for (b, real, fake, gb) in calls:
        f = float(ref.choice(a=[0.1, 0.9], p=[0.95, 0.05]))
        r = float(ref.choice(a=[0.1, 0.9], p=[0.05, 0.95]))
        assert (real, fake) == (r, f) and gb == b
    assert [c[0] for c in calls] == [8, 8, 5]


def test_gan_train_epoch_keeps_the_reference_bookkeeping():
    from disentangle_mlp_amd.trainer import GANTrainer
    tr = GANTrainer(device="cpu")
    assert tr.flat_g is None and tr.flat_d is None              # single process: no exchange
    n_calls = [0]

    def fake_step(data, real_label=None, fake_label=None, global_batch=None, **kw):
        n_calls[0] += 1
        return {"errG": torch.tensor(2.0), "errD_real": torch.tensor(0.25), "errD_fake": torch.tensor(0.5)}
    tr.step = fake_step
    loader = _Loader([torch.zeros(4, 3, 64, 64)] * 3, n=12)
    g, d = tr.train_epoch(loader, label_rng=np.random.RandomState(0))
    assert g == pytest.approx(6.0 / 12) and d == pytest.approx(6.0 / 12 / 12)     # new_gan.py:137-138
    assert tr.last_epoch_sums == {"errG": pytest.approx(6.0), "errD": pytest.approx(2.25)}
    dp = GANTrainer(device="cpu", data_parallel=True)             # both networks get an exchange (new_gan.py:51-53)
    assert dp.flat_g is not None and dp.flat_d is not None


# ------------------------------------------------------------------ ragged data-parallel tail
def test_loader_publishes_the_global_batch_of_a_ragged_tail():
    from disentangle_mlp_amd.data import DeviceLoader

    class DS:
        device = "cpu"

        def __len__(self):
            return 37
    order = torch.arange(37)
    seen = {}
    for rank in range(4):
        ld = DeviceLoader(DS(), 16, rank=rank, world_size=4)
        seen[rank] = [(c.tolist(), ld.last_global_batch) for c in ld.index_batches(order)]
    # 16 + 16 + 5: the tail of 5 is scattered 2/2/1/0 -> too short for every rank, dropped on all
    assert all(len(v) == 2 and all(gb == 16 for _, gb in v) for v in seen.values())
    for rank in range(2):
        ld = DeviceLoader(DS(), 16, rank=rank, world_size=2)
        got = [(c.tolist(), ld.last_global_batch) for c in ld.index_batches(order)]
        assert [gb for _, gb in got] == [16, 16, 5]          # the BCE mean of the tail runs over 5 images
        assert len(got[2][0]) == (3 if rank == 0 else 2)     # DataParallel's scatter: ceil(5/2), rest


# ------------------------------------------------------------------ __main__ of the experiment script (:211-267)
def test_fit_and_evaluate_follow_the_reference_main(tmp_path, monkeypatch, capsys):
    """fit: per epoch train_epoch -> model_{epoch+1}.tar -> FID -> the printed line -> the logger row; evaluate: per
    checkpoint the reference's epoch renumbering, FID samples, nrow=1 reconstructions, five samples named after
    start_epoch.  Kernels stubbed (CPU): the control flow, file names and arguments are what is checked."""
    from disentangle_mlp_amd import image_io
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
    tr = BetaVAEGANTrainer(device="cpu")
    calls = []
    tr.train_epoch = lambda loader, label_rng=None, max_iterations=None: (calls.append(("train", loader)) or (3.0, 3.0, 0.5, 0.5))
    monkeypatch.setattr(image_io, "generate_fid_samples", lambda fn, epoch, n, nh, path, device="cuda": calls.append(("fid_samples", epoch, n, nh, path)))
    monkeypatch.setattr(image_io, "gen_reconstructions", lambda fn, dl, epoch, path, nrow=8, path_for_originals="", device="cuda": calls.append(("recons", epoch, path, nrow, path_for_originals)))
    monkeypatch.setattr(image_io, "generate_samples", lambda fn, epoch, n, nh, path, nrow=8, device="cuda": calls.append(("samples", epoch, n, path, nrow)))
    logged = []
    rows = tr.fit("LOADER", epochs=3, start_epoch=1, model_path=str(tmp_path), calc_fid=True, n_samples=7,
                  fid_path_recons="FIDDIR", fid_path_pretrained="PRE", get_fid=lambda a, b: 12.5 if (a, b) == ("FIDDIR", "PRE") else None,
                  log=logged.append)
    assert [r["Epoch"] for r in rows] == [1, 2] and logged == [{k: v for k, v in r.items() if k != "Dx"} for r in rows]
    assert logged[0] == {"Epoch": 1, "Avg Eec Loss": 3.0, "Avg Dnc Loss": 3.0, "Avg Dis Loss": 0.5, "FID": 12.5}
    assert sorted(os.listdir(tmp_path)) == ["model_2.tar", "model_3.tar"]          # {model_path}/model_{epoch+1}.tar
    ck = torch.load(tmp_path / "model_3.tar", weights_only=False)
    assert ck["epoch"] == 3 and all(k.startswith("module.") for k in ck["discriminator_model"])
    assert calls == [("train", "LOADER"), ("fid_samples", 1, 7, 128, "FIDDIR"), ("train", "LOADER"), ("fid_samples", 2, 7, 128, "FIDDIR")]
    assert "====> Epoch: 2 Avg Encoder Loss: 3.0000 Avg Decoder Loss: 3.0000 Avg Discriminator Loss: 0.5000 FID: 12.5 Dx: 0.5000" in capsys.readouterr().out
    # evaluate: checkpoints of epochs 2, 3 and 3 again -> 2, 3, 4 (the reference's "quick fix" renumbering)
    calls.clear()
    paths = [str(tmp_path / "model_2.tar"), str(tmp_path / "model_3.tar"), str(tmp_path / "model_3.tar")]
    res = tr.evaluate(paths, test_loader="TEST", start_epoch=0, calc_fid=True, n_samples=4, fid_path_samples="S", fid_path_pretrained="PRE",
                      get_fid=lambda a, b: 1.0, test_recons=True, test_results_path_recons="R", test_results_path_originals="O",
                      test_samples=True, test_results_path_samples="T")
    assert [r["epoch"] for r in res] == [2, 3, 4] and all(r["FID"] == 1.0 for r in res)
    assert calls[:3] == [("fid_samples", 2, 4, 128, "S"), ("recons", 2, "R", 1, "O"), ("samples", 0, 5, "T", 1)]
    assert [c for c in calls if c[0] == "recons"] == [("recons", e, "R", 1, "O") for e in (2, 3, 4)]

