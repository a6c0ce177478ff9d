"""Measure the vendor-GEMM algorithm table of disentangle_mlp_amd/tuned_gemms.py on this GPU.
  python scripts/tune_gemms.py tune [out.csv]    PyTorch TunableOp tunes every GEMM shape of the beta-VAE-GAN iteration
                                                 (per-GPU batch 128), of `new_vae` at batch 16 and of the DCGAN step
  python scripts/tune_gemms.py check [table.csv] look-up mode with that table: two trainers from the same seed run three
                                                 iterations each -- losses and final weights must agree bit for bit
                                                 (an algorithm that adds split-K partials with atomics would not)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
path = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "disentangle_mlp_amd", "tuned", "gfx950_gemm.csv")
import torch
import torch.cuda.tunable as tunable

if mode == "tune":
    os.environ["VG_TUNED_GEMMS"] = "0"
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_iterations(100)
    tunable.set_filename(path + ".raw", insert_device_ordinal=False)
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer, VAETrainer, GANTrainer
    g = torch.Generator().manual_seed(0)
    for B, make in ((128, lambda: BetaVAEGANTrainer(graph=False)), (16, lambda: VAETrainer(graph=False)),
                    (128, lambda: GANTrainer(graph=False))):
        tr = make()
        x = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda()
        for _ in range(2):
            tr.step(x)
        del tr
    torch.cuda.synchronize()
    res = tunable.get_results()
    with open(path, "w") as f:
        for k, v in tunable.get_validators():
            f.write(f"Validator,{k},{v}\n")
        for op, key, sol, t in res:
            f.write(f"{op},{key},{sol},{t}\n")
            print(f"{op} {key} -> {sol} {float(t) * 1e3:.1f} us", flush=True)
    print("wrote", path, len(res), "entries")
else:
    from disentangle_mlp_amd import tuned_gemms
    assert tuned_gemms.enable(path), "table not accepted (validators?)"
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(128, 3, 64, 64, generator=g) * 2 - 1).cuda()
    lat = [torch.randn(128, 128, generator=g).cuda() for _ in range(3)]
    runs = []
    for r in range(2):
        tr = BetaVAEGANTrainer(graph=False)
        losses = [{k: v.clone() for k, v in tr.step(x, *lat).items()} for _ in range(3)]
        runs.append((losses, {k: v.clone() for n in (tr.netEG, tr.netD) for k, v in n.state_dict().items()}))
    ok = all(torch.equal(a[k], b[k]) for a, b in zip(runs[0][0], runs[1][0]) for k in a) and \
        all(torch.equal(v, runs[1][1][k]) for k, v in runs[0][1].items())
    print("reproducible:", ok, {k: float(v) for k, v in runs[0][0][-1].items()})
    sys.exit(0 if ok else 1)
