"""CPU: host-side logic of the product package that needs no device -- module zoo layout /
seed recipe / checkpoint format against the oracle, and the data-parallel gradient exchange
(flat buckets + all-reduce SUM) with 2 gloo ranks against an in-process 2-replica emulation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import steps as osteps


def test_module_zoo_matches_reference_layout_and_seed_recipe():
    from disentangle_mlp_amd import model as M
    from disentangle_mlp_amd.trainer import ModelOpt
    torch.manual_seed(999)
    a, d = M.VAE(ModelOpt()), M.Discriminator_celeba(ModelOpt())
    a.apply(M.weights_init), d.apply(M.weights_init)
    b, e, _, _ = osteps.build_nets()
    for x, y in ((a, b), (d, e)):
        sx, sy = x.state_dict(), y.state_dict()
        assert list(sx) == list(sy)
        assert all(torch.equal(sx[k], sy[k]) for k in sx)       # bit-identical initial weights
    y.load_state_dict(sx)
    x.load_state_dict(sy)
    g1, g2 = M.Generator_celeba(ModelOpt()), oracle.Generator_celeba(oracle.OracleOpt())
    e1, e2 = M.Encoder_celeba(ModelOpt()), oracle.Encoder_celeba(oracle.OracleOpt())
    assert list(g1.state_dict()) == list(g2.state_dict())
    assert list(e1.state_dict()) == list(e2.state_dict())


def test_num_batches_tracked_is_flushed_into_state_dict():
    from disentangle_mlp_amd.model import HipBatchNorm2d
    bn = HipBatchNorm2d(4)
    bn._nbt_pending = 3
    assert int(bn.state_dict()["num_batches_tracked"]) == 3
    assert bn._nbt_pending == 0


def test_transposed_conv_rejects_foreign_output_size():
    from disentangle_mlp_amd.model import HipConvTranspose2d
    m = HipConvTranspose2d(4, 4, 2)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 8, 8), output_size=(1, 4, 15, 15))   # the reference's Generator at n_z!=8x8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _replica_grads(rank, world, batch, phase_out):
    """Oracle phase-1 (discriminator) gradients of one replica on its shard, BCE divided by
    the global batch."""
    eg, d, oeg, od = osteps.build_nets()
    b = osteps.synthetic_batch(batch)
    lo, hi = rank * batch // world, (rank + 1) * batch // world
    sh = {k: v[lo:hi] for k, v in b.items()}
    grads = {}

    class Stop(Exception):
        pass

    def hook(ph, net):
        grads[ph] = [p.grad.detach().clone() for p in net.parameters()]
        raise Stop
    try:
        osteps.betavaegan_step(eg, d, oeg, od, sh["data"], sh["noise"], sh["eps2"], sh["eps3"],
                               bce_divisor=batch, grad_hook=hook)
    except Stop:
        pass
    return d, grads["D"]


def _dp_worker(rank, world, port, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from disentangle_mlp_amd.trainer import FlatGrads
    d, g = _replica_grads(rank, world, batch, None)
    flat = FlatGrads(d.parameters())
    flat.zero_and_attach()
    for p, gi in zip(d.parameters(), g):     # what autograd's AccumulateGrad does in the trainer:
        if p.grad is None:
            p.grad = gi.clone()              # large tensors: the produced gradient is installed
        else:
            p.grad.add_(gi)                  # small tensors: accumulated into the flat views
    flat.finish()                            # no hook fired (no backward here): every bucket reduced now
    if rank == 0:
        q.put(flat.gathered().numpy())       # numpy: pickled through the pipe (a torch tensor travels as a shared-
                                             # memory handle that dies with this process)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_two_replica_emulation():
    world, batch = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.set_num_threads(2)
    want = None
    for r in range(world):
        _, g = _replica_grads(r, world, batch, None)
        flat = torch.cat([t.flatten() for t in g])
        want = flat if want is None else want + flat
    assert got.shape == want.shape == (36122945,)
    err = float((got - want).norm() / want.norm())
    assert err <= 1e-6, err


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disentangle_mlp_amd.trainer import FlatGrads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 8))
    unused = torch.nn.Parameter(torch.zeros(5))            # never receives a gradient: finish() must cover it
    params = list(net.parameters()) + [unused]
    flat = FlatGrads(params, bucket_bytes=512 * 4, direct_bytes=64 * 64 * 4, overlap=True)
    assert len(flat.buckets) >= 2 and sum(flat.direct) == 1      # the 64x64 weight goes direct
    res = []
    for it in range(2):                                    # two phases reuse the same buffers
        flat.zero_and_attach()
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(16, 40, generator=g)
        net(x).pow(2).sum().backward()
        fired_early = sum(flat._launched) + len(flat._direct_done)
        flat.finish()
        res.append((flat.gathered().numpy().copy(), fired_early))   # numpy: see _dp_worker
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_overlapped_allreduce_two_ranks(world):
    """The bucketed, hook-driven gradient exchange (what runs over RCCL on the GPUs) with 2 and with 4 gloo
    ranks equals the sum of the ranks' gradients; buckets fire during backward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for it, (got, fired_early) in enumerate(res):
        got = torch.from_numpy(got)
        assert fired_early >= 2                            # overlap: launched from the hooks
        want = None
        for r in range(world):
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64),
                                      torch.nn.ReLU(), torch.nn.Linear(64, 8))
            g = torch.Generator().manual_seed(100 * it + r)
            net(torch.randn(16, 40, generator=g)).pow(2).sum().backward()
            flat = torch.cat([p.grad.flatten() for p in net.parameters()] + [torch.zeros(5)])
            want = flat if want is None else want + flat
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)


def test_deferred_linear_weight_gradient_is_the_sum_over_passes(monkeypatch):
    """functional.deferred_wgrad(): a Linear layer applied twice before one backward gets ONE weight-gradient GEMM over
    the concatenated passes (by whichever backward node runs last), the same gradients as autograd's two GEMMs + add;
    a forward without its backward inside the context is an error."""
    from disentangle_mlp_amd import functional as HF, ops
    monkeypatch.setattr(HF, "DEFER_MIN_WEIGHTS", 1)
    monkeypatch.setattr(HF, "DEFER_WGRAD", True)
    torch.manual_seed(0)
    w, b = torch.randn(6, 5, requires_grad=True), torch.randn(6, requires_grad=True)
    x1, x2 = torch.randn(4, 5, requires_grad=True), torch.randn(3, 5)
    ref = torch.nn.functional.linear(x1, w, b).pow(2).sum() + torch.nn.functional.linear(x2, w, b).sin().sum()
    gw, gb, gx = torch.autograd.grad(ref, (w, b, x1))
    with HF.deferred_wgrad():
        y = HF.linear(x1, w, b).pow(2).sum() + HF.linear(x2, w, b).sin().sum()
        y.backward()
    assert torch.allclose(w.grad, gw, rtol=1e-5, atol=1e-5) and torch.allclose(b.grad, gb) and torch.allclose(x1.grad, gx)
    with pytest.raises(RuntimeError, match="got no backward"):
        with HF.deferred_wgrad():
            HF.linear(x1, w, b)
    # a backward that runs after its context has closed computes its own weight gradient (no shared state to trip over)
    w.grad = None
    with pytest.raises(RuntimeError, match="got no backward"):
        with HF.deferred_wgrad():
            late = HF.linear(x2, w, b).sum()
    late.backward()
    assert torch.allclose(w.grad, torch.ones(3, 6).t() @ x2, rtol=1e-5, atol=1e-5)
    # contexts nest and do not share state
    w.grad = None
    with HF.deferred_wgrad() as outer:
        y1 = HF.linear(x2, w, b).sum()
        with HF.deferred_wgrad() as inner:
            y2 = HF.linear(x2, w, b).sum()
            assert sum(inner.pending.values()) == 1 and sum(outer.pending.values()) == 1
            y2.backward()
        y1.backward()
    assert torch.allclose(w.grad, 2 * torch.ones(3, 6).t() @ x2, rtol=1e-5, atol=1e-5)


def test_fused_chain_falls_back_to_module_calls_when_hooked():
    """model.FusedChain / the decoder chain call the kernels of several modules at once; with a forward hook on any of
    them (a feature tap, a profiler) the chain must run module by module so that the hook fires."""
    from disentangle_mlp_amd import model as M
    from disentangle_mlp_amd.trainer import ModelOpt
    d = M.Discriminator_celeba(ModelOpt())
    mods = list(d.convs)
    assert not M._has_hooks(mods)
    h = d.convs[3].register_forward_hook(lambda m, i, o: None)
    assert M._has_hooks(mods)
    h.remove()
    assert not M._has_hooks(mods)
    h = d.convs[1].register_forward_pre_hook(lambda m, i: None)
    assert M._has_hooks(mods)
    h.remove()


def test_tuned_gemm_table_is_well_formed_and_inert_without_a_gpu():
    """tuned_gemms: the shipped vendor-GEMM algorithm table parses (validators first, then one row per GEMM shape with a
    solution name and a positive time), covers the three 16384 <-> 2048 GEMMs of the benchmark batch, and enable() is a
    no-op on a box without a GPU (and when VG_TUNED_GEMMS=0)."""
    from disentangle_mlp_amd import tuned_gemms
    rows = [l.strip().split(",") for l in open(tuned_gemms.TABLE) if l.strip()]
    vals = [r for r in rows if r[0] == "Validator"]
    ops = [r for r in rows if r[0] != "Validator"]
    assert {v[1] for v in vals} >= {"PT_VERSION", "GCN_ARCH_NAME", "HIPBLASLT_VERSION", "ROCBLAS_VERSION"}
    assert any(v[1] == "GCN_ARCH_NAME" and v[2].startswith("gfx950") for v in vals)
    assert ops and all(len(r) == 4 and r[0].startswith("Gemm") and float(r[3]) > 0 for r in ops)
    keys = {r[1] for r in ops}
    assert {"tn_2048_128_16384_ld_16384_16384_2048", "nn_16384_128_2048_ld_16384_2048_16384",
            "nt_16384_2048_128_ld_16384_2048_16384"} <= keys          # forward, data gradient, weight gradient at M = 128
    if not torch.cuda.is_available():
        assert tuned_gemms.enable() is False


def test_accumulate_param_grads_adds_inside_the_second_pass(monkeypatch):
    """functional.accumulate_param_grads(): a layer applied twice before one backward hands autograd ONE gradient tensor;
    the second pass adds into it (here the Linear Function: addmm_) -- same gradients as autograd's own sum; reset()
    separates two backwards; outside a context nothing changes."""
    from disentangle_mlp_amd import functional as HF
    torch.manual_seed(1)
    w, b = torch.randn(6, 5, requires_grad=True), torch.randn(6, requires_grad=True)
    x1, x2 = torch.randn(4, 5), torch.randn(3, 5)

    def loss():
        return HF.linear(x1, w, b).pow(2).sum() + HF.linear(x2, w, b).sin().sum()
    ref = torch.nn.functional.linear(x1, w, b).pow(2).sum() + torch.nn.functional.linear(x2, w, b).sin().sum()
    gw, gb = torch.autograd.grad(ref, (w, b))
    calls = []
    orig = torch.Tensor.addmm_
    monkeypatch.setattr(torch.Tensor, "addmm_", lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1])
    with HF.accumulate_param_grads() as acc:
        loss().backward()
        assert len(calls) == 1 and len(acc.acc) == 1           # the second pass added in place
        assert torch.allclose(w.grad, gw, rtol=1e-6, atol=1e-6) and torch.allclose(b.grad, gb, rtol=1e-6, atol=1e-6)
        acc.reset()
        w.grad = b.grad = None
        loss().backward()                                       # a new backward starts from nothing
        assert torch.allclose(w.grad, gw, rtol=1e-6, atol=1e-6) and len(calls) == 2
    w.grad = b.grad = None
    loss().backward()                                           # no context: autograd's own accumulation
    assert torch.allclose(w.grad, gw, rtol=1e-6, atol=1e-6) and len(calls) == 2



def test_captured_graph_keeps_the_buffers_it_points_at_alive():
    """ADVICE (round 3, medium): a captured HIP graph holds raw device pointers into the pack cache and the scratch
    buffers of `ops`; `packed_filter_scope.__enter__` rebuilds the cache once it exceeds `_PACK_CACHE_MAX` entries, and a
    workspace is replaced when it grows.  `_CapturedIteration` therefore keeps `ops.buffers_in_use()`: the tensors outlive
    the cache entries that pointed at them.  Host logic only (CPU tensors stand in for the device buffers)."""
    import gc
    import weakref
    from disentangle_mlp_amd import ops
    saved = dict(ops._pack_cache), dict(ops._workspaces), dict(ops._pack_scratch)
    try:
        ops._pack_cache.clear(), ops._workspaces.clear(), ops._pack_scratch.clear()
        for i in range(ops._PACK_CACHE_MAX + 2):                       # more entries than the cache tolerates
            ops._pack_cache[(i, 2, 2, (1,), 3)] = [True, 0, torch.zeros(8)]
        ops._workspaces[("cpu", None, 0)] = torch.zeros(16, dtype=torch.uint8)
        held = ops.buffers_in_use()                                      # what a capture keeps
        refs = [weakref.ref(t) for t in held]
        assert len(held) == ops._PACK_CACHE_MAX + 3
        with ops.packed_filter_scope():                                  # the next trainer's scope: the cache is rebuilt
            assert len(ops._pack_cache) == 0
        ops._workspaces[("cpu", None, 0)] = torch.zeros(32, dtype=torch.uint8)      # ... and the workspace regrown
        gc.collect()
        assert all(r() is not None for r in refs)                        # alive: the "graph" still owns them
        del held
        gc.collect()
        assert all(r() is None for r in refs)                            # and gone with it
    finally:
        ops._pack_cache.clear(), ops._workspaces.clear(), ops._pack_scratch.clear()
        ops._pack_cache.update(saved[0]), ops._workspaces.update(saved[1]), ops._pack_scratch.update(saved[2])


def test_rccl_unique_id_survives_the_broadcast_box():
    """rccl.Communicator hands the 128 bytes of ncclGetUniqueId to the other ranks through a Python `bytes`; a c_char
    array field reads back truncated at its first NUL (the first version of the binding: ncclCommInitRank then timed
    out on a mangled id), so the struct is c_ubyte and is copied with string_at / memmove."""
    import ctypes
    from disentangle_mlp_amd import rccl
    uid = rccl._UniqueId()
    assert ctypes.sizeof(uid) == 128
    pattern = bytes([0, 7, 0, 0, 255, 1] + [i % 251 for i in range(122)])          # NULs early on
    ctypes.memmove(ctypes.byref(uid), pattern, 128)
    box = ctypes.string_at(ctypes.byref(uid), 128)
    assert box == pattern
    back = rccl._UniqueId()
    ctypes.memmove(ctypes.byref(back), box, 128)
    assert bytes(back.internal) == pattern


def test_fp16x3_is_the_default_arithmetic_and_planes_codes_match_the_header():
    import re
    from disentangle_mlp_amd import ops
    assert ops.CONV_ARITH == __import__("os").environ.get("VG_CONV_ARITH", "fp16x3")
    hdr = open(__import__("os").path.join(__import__("os").path.dirname(ops.__file__), "..", "include", "vaegan_hip.h")).read()
    flag = int(re.search(r"#define VG_PLANES_F16 (0x[0-9a-fA-F]+)", hdr).group(1), 16)
    assert flag == ops.PLANES_F16
    prev = ops.CONV_ARITH
    try:
        for mode, planes in (("fp32", 0), ("bf16x3", 2), ("bf16x6", 3), ("fp16x3", 2 | flag)):
            ops.CONV_ARITH = mode
            assert ops._planes() == planes and ops._thin_planes() == (3 if mode == "fp16x3" else planes)
    finally:
        ops.CONV_ARITH = prev


def test_linear_layers_dispatch_and_the_adam_tensor_struct_match_the_header():
    """Which Linear GEMMs leave the vendor library (ops.linear_split_ok), and that the ctypes mirror of VgAdamTensor has the
    header's fields in the header's order (the struct grew by `amax` in ABI 5)."""
    import os, re
    from disentangle_mlp_amd import ops, optim
    prev = ops.CONV_ARITH, ops.LINEAR_SPLIT
    try:
        ops.CONV_ARITH, ops.LINEAR_SPLIT = "fp16x3", True
        assert ops.linear_split_ok(16384, 16384 * 2048) and ops.linear_split_ok(128, 128 * 16384)
        assert not ops.linear_split_ok(2048, 2048 * 128)          # the 2048 -> 128 heads: below 2^20 weights
        assert not ops.linear_split_ok(100, 16384 * 2048)         # reduction not a multiple of 32 (odd batches' weight gradient)
        ops.CONV_ARITH = "bf16x6"
        assert not ops.linear_split_ok(16384, 16384 * 2048)       # opt-in arithmetics: vendor GEMMs
        ops.CONV_ARITH, ops.LINEAR_SPLIT = "fp16x3", False
        assert not ops.linear_split_ok(16384, 16384 * 2048)
    finally:
        ops.CONV_ARITH, ops.LINEAR_SPLIT = prev
    hdr = open(os.path.join(os.path.dirname(ops.__file__), "..", "include", "vaegan_hip.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} VgAdamTensor;", hdr).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.findall(r"(\w+)\s*;", line)[0] for line in body.split("\n") if ";" in line]
    assert fields == [f[0] for f in optim._AdamTensor._fields_] == ["p", "g", "m", "v", "n", "amax"]


def test_emitted_weight_bounds_belong_to_one_tensor_object_and_one_version():
    """ops.set_weight_bound / weight_bound (the bound HipAdam's step emits for a big Linear weight): a hit needs the SAME
    tensor object at the SAME version -- not merely the same address, shape or id."""
    import gc
    import torch
    from disentangle_mlp_amd import ops
    w, b = torch.zeros(4, 4), torch.ones(1)
    ops.set_weight_bound(w, b)
    assert ops.weight_bound(w) is b
    key = id(w)
    w.add_(1.0)                                   # version bump: stale
    ent = ops._wbound_emitted[key]
    assert ent[0]() is w and ent[1] != w._version
    ops.set_weight_bound(w, b)
    assert ops.weight_bound(w) is b
    del w
    gc.collect()
    assert ops._wbound_emitted[key][0]() is None   # a new tensor that re-uses the id can never match the dead reference
