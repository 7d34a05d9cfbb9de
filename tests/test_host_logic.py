"""CPU-only checks: the C-ABI library loads and exports every symbol include/dlrm_hip.h declares, the
drop-in module has the reference's surface, host bookkeeping is bit-exact, and the product path refuses
to run without a GPU instead of falling back."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden, params_with_prefix


def header_functions():
    src = open(os.path.join(ROOT, "include", "dlrm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dlrm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dlrm_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.dlrm_hip_abi_version() == _lib.EXPECTED_ABI == 17
    assert b"gfx950" in lib.dlrm_hip_build_info()


def test_no_cpu_fallback():
    from dlrm_amd import ops
    W = [torch.zeros(4, 4)]
    with pytest.raises(RuntimeError, match="GPU"):
        ops.BagBatch([torch.zeros(2, dtype=torch.int64)], [torch.zeros(2, dtype=torch.int64)])
    with pytest.raises(RuntimeError, match="GPU"):
        ops.linear_fwd(torch.zeros(2, 2), torch.zeros(2, 2), None, 0, torch.zeros(2, 2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dlrm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("# oracle", ""), f


def test_module_surface_and_rng_order_match_reference():
    """same numpy seed -> bit-identical initial parameters and the reference's state_dict keys"""
    import dlrm_amd
    d, meta = load_golden("config1_b128")
    np.random.seed(123)
    m = dlrm_amd.DLRM_Net(meta["m_spa"], np.asarray(meta["ln_emb"]), np.asarray(meta["ln_bot"]), np.asarray(meta["ln_top"]),
                          arch_interaction_op="dot", sigmoid_top=meta["sigmoid_top"], loss_function="bce")
    init = params_with_prefix(d, "init")
    sd = m.state_dict()
    assert list(sd.keys()) == list(init.keys())
    for k, v in init.items():
        assert np.array_equal(sd[k].numpy(), v), k
    for attr in ("emb_l", "v_W_l", "bot_l", "top_l", "loss_fn", "ndevices", "loss_threshold", "weighted_pooling",
                 "quantize_emb", "apply_mlp", "apply_emb", "interact_features", "create_emb", "create_mlp",
                 "sequential_forward", "distributed_forward"):
        assert hasattr(m, attr), attr
    assert isinstance(m.top_l, torch.nn.Sequential) and isinstance(m.top_l[0], torch.nn.Linear) and len(m.top_l) == 6
    assert all(p.requires_grad for p in m.parameters())
    empty = dlrm_amd.DLRM_Net()
    assert len(list(empty.parameters())) == 0


def test_ext_dist_partition_bit_exact():
    from dlrm_amd import ext_dist
    with open(os.path.join(GOLDEN, "bookkeeping.json")) as f:
        book = json.load(f)
    saved = (ext_dist.my_rank, ext_dist.my_size)
    try:
        for case in book["partition"]:
            for rank, want in enumerate(case["ranks"]):
                ext_dist.my_rank, ext_dist.my_size = rank, case["size"]
                sl = ext_dist.get_my_slice(case["n"])
                assert [sl.start, sl.stop, sl.step] == want["slice"]
                mine, splits = ext_dist.get_split_lengths(case["n"])
                assert mine == want["my_len"] and splits == want["splits"]
    finally:
        ext_dist.my_rank, ext_dist.my_size = saved


def test_error_strings_match_reference():
    import dlrm_amd
    np.random.seed(0)
    with pytest.raises(SystemExit, match="--loss-function=huber is not supported"):
        dlrm_amd.DLRM_Net(2, np.asarray([4, 3]), np.asarray([4, 2]), np.asarray([5, 1]), "dot", loss_function="huber")
    m = dlrm_amd.DLRM_Net(2, np.asarray([4, 3]), np.asarray([4, 2]), np.asarray([5, 1]), "foo")
    with pytest.raises(SystemExit, match="--arch-interaction-op=foo is not supported"):
        m.interact_features(torch.zeros(1, 2), [torch.zeros(1, 2)])


REFERENCE = os.environ.get("DLRM_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "dlrm_s_pytorch.py")),
                    reason="reference checkout not present (it is absent on the GPU box by design)")
def test_launcher_swaps_the_hot_path_under_the_unmodified_reference_run(tmp_path):
    """python -m dlrm_amd.launch: the reference's own run() builds OUR DLRM_Net from its CLI (construction only here —
    zero epochs — because the product path has no CPU fallback); with one epoch it must fail loudly, not fall back."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    probe = ("import sys, numpy as np; sys.argv=['x']; from dlrm_amd import launch; import dlrm_amd; "
             "ref = launch.load_reference(%r); assert ref.DLRM_Net is dlrm_amd.DLRM_Net; "
             "assert ref.ext_dist is dlrm_amd.ext_dist; print('swapped')" % REFERENCE)
    r = subprocess.run([sys.executable, "-c", probe], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "swapped" in r.stdout, r.stderr[-2000:]
    cli = ["--arch-sparse-feature-size=16", "--arch-embedding-size=1000-1000-1000", "--arch-mlp-bot=13-512-16",
           "--arch-mlp-top=22-256-1", "--mini-batch-size=128", "--data-generation=random", "--num-batches=2"]
    base = [sys.executable, "-m", "dlrm_amd.launch", "--reference", REFERENCE, "--"]
    r = subprocess.run(base + cli + ["--nepochs=0"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(base + cli + ["--nepochs=1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_embedding_update_plan_for_sgd_and_rowwise_adagrad():
    """The optimizer-step pre-hook's plan: SGD -> ("sgd", lr); RWSAdagrad (ours, and the reference's own class when the
    reference checkout is present) -> clr = lr / (1 + (step-1)*lr_decay) (optim/rwsadagrad.py:113-115), row-wise state
    created lazily in optimizer.state[p]["momentum"] with initial_accumulator_value, step counted per table."""
    import sys
    from dlrm_amd.dlrm_net import _embedding_update_plan
    from dlrm_amd.optim import FusedRWSAdagrad
    tables = [torch.nn.Parameter(torch.zeros(7, 4)), torch.nn.Parameter(torch.zeros(3, 4))]
    dense = torch.nn.Parameter(torch.zeros(5))
    sgd = torch.optim.SGD(tables + [dense], lr=0.25)
    assert _embedding_update_plan(sgd, tables) == ("sgd", 0.25)
    assert _embedding_update_plan(torch.optim.SGD([dense], lr=0.1), tables) is None      # does not own the tables
    # optimizers the fused kernels do not implement take the reference's own route: the sparse COO gradient is
    # materialised (dlrm_emb_bwd_coo) and the optimizer's step consumes it
    assert _embedding_update_plan(torch.optim.SGD(tables, lr=0.1, momentum=0.9), tables) == ("coo",)
    assert _embedding_update_plan(torch.optim.Adagrad(tables, lr=0.1), tables) == ("coo",)
    with pytest.raises(SystemExit):          # gradient accumulation + the non-linear fused row-wise update: refused, not approximated
        _embedding_update_plan(FusedRWSAdagrad(tables, lr=0.1), tables, count=2)
    classes = [FusedRWSAdagrad]
    if os.path.isfile(os.path.join(REFERENCE, "optim", "rwsadagrad.py")):
        sys.path.insert(0, os.path.join(REFERENCE, "optim"))
        import rwsadagrad
        classes.append(rwsadagrad.RWSAdagrad)
    for cls in classes:
        opt = cls(tables + [dense], lr=0.5, lr_decay=0.1, initial_accumulator_value=0.25, eps=1e-7)
        kind, clr, eps, states = _embedding_update_plan(opt, tables)
        assert kind == "rwsadagrad" and clr == 0.5 and eps == 1e-7
        assert [tuple(s_.shape) for s_ in states] == [(7,), (3,)] and all(float(s_[0]) == 0.25 for s_ in states)
        assert states[0] is opt.state[tables[0]]["momentum"] and opt.state[tables[1]]["step"] == 1
        _, clr2, _, states2 = _embedding_update_plan(opt, tables)
        assert abs(clr2 - 0.5 / 1.1) < 1e-12 and states2[0] is states[0] and opt.state[tables[0]]["step"] == 2


def test_accumulation_guard_compares_tables_not_argument_tuples():
    """every forward call hands EmbeddingBagsFunction a FRESH tuple of the same tables, so two parked backward
    passes must be recognised by table identity; with tuple identity the guard of the non-linear row-wise update never fired."""
    import dlrm_amd
    from dlrm_amd.optim import FusedRWSAdagrad
    tables = [torch.nn.Parameter(torch.zeros(7, 4)), torch.nn.Parameter(torch.zeros(3, 4))]
    opt = FusedRWSAdagrad(tables, lr=0.1)
    shell = dlrm_amd.DLRM_Net()
    pending = [(tuple(tables), None, None), (tuple(tables), None, None)]        # two distinct tuples, same tables
    assert pending[0][0] is not pending[1][0]
    with pytest.raises(SystemExit) as e:
        shell._apply_pending(pending, opt, None)
    assert "gradient accumulation" in str(e.value)
    assert opt.state[tables[0]].get("step", 0) == 0                             # refused before any state was advanced
    # tables of ANOTHER model parked beside them do not count
    other = [torch.nn.Parameter(torch.zeros(5, 4))]
    opt2 = FusedRWSAdagrad(tables + other, lr=0.1)
    from dlrm_amd.dlrm_net import _embedding_update_plan
    key = tuple(id(w) for w in tables)
    assert sum(1 for p_ in [(tuple(tables),), (tuple(other),)] if tuple(id(w) for w in p_[0]) == key) == 1
    assert _embedding_update_plan(opt2, tables, count=1)[0] == "rwsadagrad"


def test_flat_gradient_slots_live_on_the_parameter_and_die_with_the_wrapper():
    """the flat-buffer slot of a FlatDDP-wrapped parameter is an attribute of the Parameter (validated at use), not an
    entry of a process-global table keyed by its address; releasing / collecting the wrapper removes it."""
    import gc
    from dlrm_amd import ext_dist, functional
    tower = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU())
    w = ext_dist.FlatDDP(tower, broadcast=False)
    p = tower[0].weight
    v = functional._grad_out(p)
    assert v.data_ptr() == w.flat.data_ptr() and v.shape == p.shape
    functional.ARENA_BUSY.clear()
    assert not hasattr(functional, "GRAD_ARENAS")
    p._dlrm_grad_arena = (p._dlrm_grad_arena[0], 10 ** 6)                       # a slot that no longer fits is an error, not a stray write
    with pytest.raises(RuntimeError):
        functional._grad_out(p)
    del w
    gc.collect()
    assert not hasattr(p, "_dlrm_grad_arena")
    assert functional._grad_out(p).data_ptr() != 0 and not functional.ARENA_BUSY


def test_oracle_ref_recipe_builds_an_importable_sourceless_reference(tmp_path):
    """`make -C oracle ref` (oracle/build_ref.py): the reference's modules compiled where they lie into oracle/_ref/*.pyc — no
    source text in the tree — import and run on a box without a checkout (here: a subprocess that cannot see /root/reference)."""
    import subprocess
    import sys
    from oracle import build_ref
    if os.path.isfile(os.path.join(REFERENCE, "dlrm_s_pytorch.py")):
        assert build_ref.build(REFERENCE, quiet=True) == build_ref.OUT
    d = build_ref.ref_dir()
    if d is None:
        pytest.skip("no reference checkout and no prebuilt oracle/_ref")
    assert not [f for _, _, fs in os.walk(d) for f in fs if f.endswith(".py")], "reference SOURCES must never be copied"
    manifest = json.load(open(os.path.join(d, "MANIFEST.json")))
    assert set(manifest["modules"]) >= {"dlrm_s_pytorch.py", "extend_distributed.py", "optim/rwsadagrad.py"}
    probe = ("import sys, types; tb = types.ModuleType('torch.utils.tensorboard'); tb.SummaryWriter = object; import torch.utils; "
             "sys.modules['torch.utils.tensorboard'] = tb; sys.path[:] = [p for p in sys.path if 'reference' not in p]; "
             "sys.path.insert(0, %r); import dlrm_s_pytorch as r, optim.rwsadagrad as o; "
             "assert r.__file__.endswith('.pyc') and hasattr(r, 'DLRM_Net') and hasattr(o, 'RWSAdagrad'); print('ok')" % d)
    r = subprocess.run([sys.executable, "-c", probe], cwd=tmp_path, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    # the launcher accepts the compiled tree as --reference
    base = [sys.executable, "-m", "dlrm_amd.launch", "--reference", d, "--", "--arch-sparse-feature-size=16",
            "--arch-embedding-size=100-100", "--arch-mlp-bot=13-32-16", "--arch-mlp-top=19-8-1", "--mini-batch-size=16",
            "--data-generation=random", "--num-batches=2", "--nepochs=0"]
    r = subprocess.run(base, cwd=tmp_path, env=dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_fused_rwsadagrad_hyperparameter_checks():
    from dlrm_amd.optim import FusedRWSAdagrad
    p = [torch.nn.Parameter(torch.zeros(3))]
    for bad in (dict(lr=-1.0), dict(lr_decay=-0.1), dict(eps=-1e-3), dict(initial_accumulator_value=-1.0), dict(weight_decay=0.1)):
        with pytest.raises(ValueError):
            FusedRWSAdagrad(p, **bad)
    opt = FusedRWSAdagrad(p, lr=0.1)
    assert opt.state[p[0]]["step"] == 0 and opt.defaults["eps"] == 1e-10 and opt.defaults["lr_decay"] == 0.0


def test_graph_input_helpers():
    """Static-input bookkeeping of the whole-step HIP graph: shared tensors stay shared, stale shapes are refused."""
    from dlrm_amd.graph import _clone_struct, _copy_struct
    off = torch.arange(5)
    lst = [off, off, torch.arange(5) * 2]
    st = _clone_struct(lst)
    assert st[0] is st[1] and st[0] is not off and torch.equal(st[2], lst[2])
    _copy_struct(st, [off + 1, off + 1, off * 3])
    assert torch.equal(st[0], off + 1) and torch.equal(st[2], off * 3)
    t = _clone_struct(torch.ones(2, 3))
    _copy_struct(t, torch.zeros(2, 3))
    assert float(t.sum()) == 0
    with pytest.raises(RuntimeError, match="shape"):
        _copy_struct(t, torch.zeros(3, 3))
    with pytest.raises(RuntimeError, match="shape"):
        _copy_struct(st, [off, off, torch.arange(6)])
    with pytest.raises(RuntimeError, match="structure"):
        _copy_struct(st, [off, off])


def test_graph_input_pairs_for_the_raw_replay():
    """The raw replay path (dlrm_graph_replay: wait + input copies + hipGraphLaunch in one C call) collects (static, caller) tensor pairs
    through the same checks as the copy_ path: one pair per DISTINCT static tensor, none where the caller already passed the static
    buffer, the shape / structure errors unchanged."""
    from dlrm_amd.graph import _clone_struct, _copy_struct
    off = torch.arange(5)
    st = _clone_struct([off, off, torch.arange(5) * 2])
    pairs = []
    put = lambda d, s_: pairs.append((d, s_))           # noqa: E731
    a, b = off + 1, off * 3
    _copy_struct(st, [a, a, b], put)
    assert len(pairs) == 2 and pairs[0][0] is st[0] and pairs[0][1] is a and pairs[1][0] is st[2] and pairs[1][1] is b
    assert torch.equal(st[0], off)                      # nothing was copied: the pairs are for the C call
    pairs.clear()
    _copy_struct(st, [st[0], st[1], b], put)            # the caller handed the static buffers back: only the third tensor moves
    assert len(pairs) == 1 and pairs[0][1] is b
    with pytest.raises(RuntimeError, match="shape"):
        _copy_struct(st, [a, a, torch.arange(6)], put)
    t = _clone_struct(torch.ones(2, 3))
    pairs.clear()
    _copy_struct(t, t, put)
    assert pairs == []


def test_small_batch_tower_path_is_chosen_by_rows_widths_and_weight_traffic(monkeypatch):
    """functional._tower_applies: the whole-tower kernels (csrc/tower.hip) take native-fp32 towers of small batches only — at most
    TOWER_ROWS rows, widths the kernels hold in LDS (<= 512), at most 8 layers, and (M / 16) x parameters x 4 bytes of weight streaming
    under TOWER_L2_BYTES: Criteo-Kaggle's towers at 2048 rows yes (bench/dlrm_s_criteo_kaggle.sh:24), the Criteo-Terabyte towers never
    (1024-wide layers), any tower at the headline batch no."""
    from dlrm_amd import functional, ops

    class T:                                             # the attributes _tower_applies reads of a tensor
        def __init__(self, *shape, cuda=True):
            self.shape, self.is_cuda = shape, cuda

        def size(self, i):
            return self.shape[i]

    def tower(widths):
        ps = []
        for k, n in zip(widths[:-1], widths[1:]):
            ps += [T(n, k), T(n)]
        return ps

    f32, bf16 = ops.arith_code("f32"), ops.arith_code("bf16")
    monkeypatch.setattr(functional, "TOWER_ROWS", 4096)
    monkeypatch.setattr(functional, "TOWER_L2_BYTES", 384 << 20)
    kag_bot, kag_top = [13, 512, 256, 64, 16], [367, 512, 256, 1]
    tb_bot, tb_top = [13, 512, 256, 128], [479, 1024, 1024, 512, 256, 1]
    ok = lambda M, w, xw=None, arith=f32, cuda=True: functional._tower_applies(T(M, xw or w[0], cuda=cuda), arith, tower(w), len(w) - 1)   # noqa: E731
    assert ok(2048, kag_bot) and ok(2048, kag_top, xw=368) and ok(128, kag_bot) and ok(1, [4, 4])
    assert not ok(2048, kag_top, xw=370)                 # an input that is neither the true nor the padded width
    assert not ok(2048, tb_top, xw=480)                  # 1024-wide layers do not fit the kernels' LDS buffers
    assert ok(2048, tb_bot) and not ok(8192, tb_bot)     # rows
    assert not ok(65536, kag_bot) and not ok(0, kag_bot)
    assert not ok(2048, kag_bot, arith=bf16) and not ok(2048, kag_bot, cuda=False)
    assert not ok(64, [16] * 10)                         # more layers than a launch takes
    wide = [512] * 9                                     # 8 layers of 512 x 512: 8 MB of weights per 16-row workgroup
    assert ok(256, wide) and not ok(4096, wide)          # 16 x 8 MB = 128 MB of streaming yes, 256 x 8 MB = 2 GB no
    monkeypatch.setattr(functional, "TOWER_ROWS", 0)
    assert not ok(2048, kag_bot)


def test_data_front_ends_refuse_cpu_and_malformed_input(tmp_path):
    from dlrm_amd.criteo_bin import CriteoBinBatches, batch_byte_range, num_batches
    from dlrm_amd.datagen import UniformBatchGenerator
    from dlrm_amd.evaluate import inference
    with pytest.raises(RuntimeError, match="GPU"):
        UniformBatchGenerator(13, [10, 20], device="cpu")
    g = object.__new__(UniformBatchGenerator)
    g.seed = 5
    seeds = {g._seed(b, s_) for b in range(50) for s_ in (1, 2, 16, 48)}
    assert len(seeds) == 200 and all(0 <= x < 2 ** 64 for x in seeds)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\0" * 161)
    with pytest.raises(RuntimeError, match="160-byte"):
        CriteoBinBatches(str(bad), 4)
    ok = tmp_path / "ok.bin"
    ok.write_bytes(b"\0" * 160 * 10)
    with pytest.raises(RuntimeError, match="GPU"):
        CriteoBinBatches(str(ok), 4, device="cpu")
    assert num_batches(1600, 4) == 3 and batch_byte_range(1600, 4, 2) == (1280, 1600)
    with pytest.raises(RuntimeError, match="no test batch"):
        inference(None, [])


def test_chunk_pack_permutation_and_gradient():
    """Row bookkeeping of the pipelined all-to-all (functional.ChunkPackFunction): chunk c holds, per destination rank r,
    rows r*Bl + c*Bc .. of the global batch; backward routes every gradient row back to its global position."""
    from dlrm_amd.functional import ChunkPackFunction
    N, C, Bc, W = 3, 4, 2, 5
    B = N * C * Bc
    E = torch.arange(B * W, dtype=torch.float32).view(B, W).requires_grad_()
    outs = ChunkPackFunction.apply(E, N, C)
    assert len(outs) == C and all(o.shape == (N * Bc, W) and o.is_contiguous() for o in outs)
    for c in range(C):
        for r in range(N):
            assert torch.equal(outs[c][r * Bc:(r + 1) * Bc], E.detach()[r * C * Bc + c * Bc:r * C * Bc + (c + 1) * Bc])
    sum((o * (i + 1)).sum() for i, o in enumerate(outs)).backward()
    want = torch.tensor([(b % (C * Bc)) // Bc + 1.0 for b in range(B)]).view(B, 1).expand(B, W)
    assert torch.equal(E.grad, want)
    with pytest.raises(RuntimeError, match="split"):
        ChunkPackFunction.apply(torch.zeros(10, 2), 3, 2)


def test_dense_sync_environment_switch_resolves_ddp_to_the_flat_wrapper():
    """DLRM_DENSE_SYNC=flat: the reference's run() calls ext_dist.DDP(tower, device_ids=[...]) and gets the flat-buffer wrapper
    (same constructor surface); unset, ext_dist.DDP is torch's DistributedDataParallel, re-exported as the reference does."""
    import subprocess
    import sys
    code = ("from dlrm_amd import ext_dist; import inspect; "
            "print(ext_dist.DDP.__name__, 'device_ids' in inspect.signature(ext_dist.FlatDDP.__init__).parameters, ext_dist.TorchDDP.__name__)")
    for env, want in (({"DLRM_DENSE_SYNC": "flat"}, "FlatDDP True DistributedDataParallel"), ({}, "DistributedDataParallel True DistributedDataParallel")):
        e = {k: v for k, v in os.environ.items() if k != "DLRM_DENSE_SYNC"}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, (r.stdout, r.stderr[-500:])


def test_fused_lookup_defaults_and_pmc_categories(monkeypatch):
    """The fused lookup + interaction path is the product default (DLRM_FUSE_EMB_INTERACT=0 and bench.py --no-fuse turn it off), and the
    PMC folding tool books the <NI, true> instantiations of the LDS-DMA interaction kernels under emb_interact_*, never under interact_*."""
    import importlib
    import sys as _sys
    import numpy as _np
    import dlrm_amd
    ln = _np.asarray([10, 20])
    kw = dict(m_spa=4, ln_emb=ln, ln_bot=_np.asarray([3, 4]), ln_top=_np.asarray([4 + 3, 2, 1]), arch_interaction_op="dot")
    _np.random.seed(0)
    assert dlrm_amd.DLRM_Net(**kw).fuse_emb_interact is True
    monkeypatch.setenv("DLRM_FUSE_EMB_INTERACT", "0")
    assert dlrm_amd.DLRM_Net(**kw).fuse_emb_interact is False
    monkeypatch.delenv("DLRM_FUSE_EMB_INTERACT")

    tools = os.path.join(ROOT, "tools")
    _sys.path.insert(0, tools)
    try:
        p2j = importlib.import_module("pmc_to_json")
    finally:
        _sys.path.remove(tools)
    fused, plain = "interact_fwd_dma_kernel<14, true>", "interact_fwd_dma_kernel<14, false>"
    assert p2j.in_cat("emb_interact_fwd", p2j.CATS["emb_interact_fwd"], fused) and not p2j.in_cat("interact_fwd", p2j.CATS["interact_fwd"], fused)
    assert p2j.in_cat("interact_fwd", p2j.CATS["interact_fwd"], plain) and not p2j.in_cat("emb_interact_fwd", p2j.CATS["emb_interact_fwd"], plain)
    assert p2j.in_cat("interact_bwd", p2j.CATS["interact_bwd"], "interact_bwd_kernel<2>") and p2j.in_cat("emb_fwd", p2j.CATS["emb_fwd"], "emb_fwd_kernel<4, 32, 1, long long, 2>")

    monkeypatch.setattr(_sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    assert bench.parse().fuse is True
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--no-fuse"])
    assert bench.parse().fuse is False


def test_bench_gpus_n_launches_n_ranks_or_refuses(monkeypatch):
    """VERDICT r3 missing-3: `python bench.py --gpus 8` (no launcher around it) must never print an `n_gpus: 1` line.  With WORLD_SIZE
    unset and N > 1 bench.py re-executes itself under torch.distributed.run with one rank per GPU (the reference's own launch
    pattern, README.md:345-346); DLRM_BENCH_NO_SELF_LAUNCH=1 makes it exit with the command line instead; a launcher whose
    WORLD_SIZE disagrees with --gpus is refused with exit code 2."""
    import importlib
    import sys as _sys
    import types
    bench = importlib.import_module("bench")
    ns = lambda n: types.SimpleNamespace(gpus=n)     # noqa: E731
    assert bench.resolve_world(ns(1), [], {}) == ("run", 1)
    assert bench.resolve_world(ns(1), ["--gpus", "1"], {"WORLD_SIZE": "1"}) == ("run", 1)
    assert bench.resolve_world(ns(8), ["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "3"}) == ("run", 8)
    what, cmd = bench.resolve_world(ns(8), ["--gpus", "8", "--steps", "7", "--warmup", "2"], {})
    assert what == "exec" and cmd[0] == _sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(ns(4), ["--gpus", "4"], {"DLRM_BENCH_NO_SELF_LAUNCH": "1"})
    assert "torch.distributed.run" in str(e.value.code) and "--nproc-per-node=4" in str(e.value.code)
    for env in ({"WORLD_SIZE": "2"}, {"WORLD_SIZE": "1"}):
        with pytest.raises(SystemExit) as e:
            bench.resolve_world(ns(8), ["--gpus", "8"], env)
        assert e.value.code == 2
    # end to end without a GPU: the re-executed command line is what actually runs (a stub "python" records its argv)
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, "-c",
                        "import sys, os; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']; os.environ.pop('WORLD_SIZE', None)\n"
                        "import bench\n"
                        "os.execv = lambda exe, argv: (print('EXEC', ' '.join(argv)), sys.exit(0))\n"
                        "bench.main()"], cwd=root, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode == 0 and "EXEC" in r.stdout and "--nproc-per-node=2" in r.stdout and "n_gpus" not in r.stdout, (r.stdout, r.stderr[-2000:])


def test_product_library_has_no_work_skipping_switches():
    """DLRM_GEMM_DEBUG / DLRM_INTERACT_DEBUG / DLRM_SEG_DEBUG (timing-only bits that make kernels skip work) exist
    only in a tuning build (`make TUNING=1`): the shipped library does not contain their names, so it cannot read them.
    VERDICT r5 #6 (round 6): the same now holds for EVERY environment variable the library ever read — launch-plan and schedule knobs
    (DLRM_GEMM_TM, DLRM_WGRAD_WGS, DLRM_SORT, DLRM_BF16_PHASED, ...) fold to their defaults at compile time (csrc/common.h DLRM_TUNE_ENV): no
    string of the shipped binary starts with DLRM_, and no source calls getenv outside the DLRM_TUNING block of common.h ("no process-wide
    mutable state": behaviour travels with the arguments of the call)."""
    import re
    from dlrm_amd import _lib
    _lib.load()
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"DLRM_GEMM_DEBUG", b"DLRM_INTERACT_DEBUG", b"DLRM_SEG_DEBUG"):
        assert name not in blob, name
    names = sorted(set(re.findall(rb"DLRM_[A-Z0-9_]{3,}", blob)))
    assert not names, "environment-style names in the product library: %r" % names
    for f in sorted(os.listdir(_lib.CSRC)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(_lib.CSRC, f)).read()
            if f == "common.h":
                head, _, tail = src.partition("#ifdef DLRM_TUNING")
                body, _, rest = tail.partition("#else")
                assert "getenv" not in head and "getenv" not in rest, f
            else:
                assert "getenv" not in src, f
    mk = open(os.path.join(_lib.CSRC, "Makefile")).read()
    assert "DLRM_TUNING" in mk and "TUNING" in mk


@pytest.mark.parametrize("ln,B,arith", [([13, 512, 256, 128], 4096, "bf16x6"), ([479, 1024, 1024, 512, 256, 1], 4096, "bf16x6"),
                                        ([480, 1024, 512, 256], 2048, "bf16x6"), ([96, 64, 32], 512, "bf16x6"), ([256, 320, 192, 64], 1000, "bf16x6"),
                                        ([13, 512, 256, 128], 4096, "bf16"), ([479, 1024, 1024, 512, 256, 1], 4096, "bf16")])
def test_mlp_storage_plan_hands_every_consumer_what_it_reads(monkeypatch, ln, B, arith):
    """MLPFunction's storage plan (functional.py: which layer keeps fp32, which keeps only the reduced-width copy — bf16, or the three planes
    of arith "bf16x6" — and which kernel form every forward / data-gradient / weight-gradient product takes) with the device operators
    replaced by plain torch on the CPU: every operator asserts that the operands it is handed exist in the form it reads, and the tower's
    output and all gradients must equal torch autograd's.  The kernels themselves are tested on the GPU; this is the HOST logic around
    them — the towers of the Terabyte model (13 -> 16 padded first layer, a 128-wide and a 1-wide layer between reduced-width layers), a
    tower the planes kernel refuses entirely, and a batch that is not a multiple of 64 (fp32 weight gradients beside planes)."""
    from dlrm_amd import functional, ops
    from dlrm_amd.functional import MLPFunction
    masks = {}

    def act(v, a):
        return torch.relu(v) if a == ops.ACT_RELU else torch.sigmoid(v) if a == ops.ACT_SIGMOID else v

    def linear_fwd(X, W, b, a, Y, arith_, relu_bits=None):
        assert Y is not None and X is not None and X.dtype == torch.float32
        y = act(X[:, :W.size(1)] @ W.t() + (b if b is not None else 0), a)
        Y.copy_(y)
        if relu_bits is not None:
            masks[relu_bits.data_ptr()] = y > 0

    def act_bwd(dY, Y, a, dZ, db):
        dZ.copy_(dY * (Y > 0) if a == ops.ACT_RELU else dY * Y * (1 - Y) if a == ops.ACT_SIGMOID else dY)

    def linear_bwd_weight(dZ, X, dW, db, accumulate=False, arith=None, **_):
        assert dZ is not None and X is not None
        dW.copy_((dZ.t() @ X)[:, :dW.size(1)])
        if db is not None:
            db.copy_(dZ.sum(0))

    def linear_bwd_data(dZ, W, Xprev, mask_act, dprev, arith_, relu_bits=None):
        assert dZ is not None
        d = dZ @ W
        if mask_act == ops.ACT_RELU:
            d = d * (masks[relu_bits.data_ptr()] if relu_bits is not None else (Xprev > 0))
        dprev.copy_(d[:, :dprev.size(1)])

    def reduced(shape3):                       # stand-in for a reduced-width tensor: right shape / dtype, the fp32 values ride along
        return torch.zeros(shape3, dtype=torch.bfloat16)

    def cast_like(planes):
        def cast(src, Npad=None, category=None):
            M, N = src.shape
            Npad = Npad or N
            t = reduced((3, M, Npad) if planes else (M, Npad))
            t.f32 = torch.zeros((M, Npad)); t.f32[:, :N] = src
            return t
        return cast

    def cast_t_like(planes):
        return lambda src, Rpad=None, category=None: cast_like(planes)(src.t().contiguous(), Rpad)

    def gemm_like(planes):
        def gemm(A, Bm, bias, a, Cf, Cr, relu_bits_out=None, relu_bits_in=None, category=None, **_):
            assert A.dim() == (3 if planes else 2) and Bm.dim() == A.dim() and A.size(-1) == Bm.size(-1), (A.shape, Bm.shape)
            if planes:
                assert ops.gemm_bf16x6_ok(A.size(1), Bm.size(1), A.size(2)), (A.shape, Bm.shape)      # no other kernel reads planes
            y = act(A.f32 @ Bm.f32.t() + (bias if bias is not None else 0), a)
            if relu_bits_out is not None:
                masks[relu_bits_out.data_ptr()] = y > 0
            if relu_bits_in is not None:
                y = y * masks[relu_bits_in.data_ptr()]
            assert Cf is not None or Cr is not None
            if Cf is not None:
                Cf.copy_(y[:, :Cf.size(1)])
            if Cr is not None:
                assert tuple(Cr.shape[-2:]) == (A.size(-2), Bm.size(-2))
                Cr.f32 = y
        return gemm

    def wgrad_like(dZr, Xr, dW, db, accumulate=False):
        assert dZr.size(-2) == Xr.size(-2)
        dW.copy_((dZr.f32.t() @ Xr.f32)[:, :dW.size(1)])
        if db is not None:
            db.copy_(dZr.f32.sum(0))

    def pad_cols(src, Kp):
        t = torch.zeros((src.size(0), Kp)); t[:, :src.size(1)] = src
        return t

    for name, fn in (("linear_fwd", linear_fwd), ("act_bwd", act_bwd), ("linear_bwd_weight", linear_bwd_weight), ("linear_bwd_data", linear_bwd_data),
                     ("pad_cols", pad_cols)):
        monkeypatch.setattr(ops, name, fn)
    def cast_multi_like(items, category=None):           # (round 5: all weight copies of a bf16 tower in one launch)
        return [(cast_like(False)(s_, cp) if cp else None, cast_t_like(False)(s_, rp) if rp else None) for s_, cp, rp in items]
    monkeypatch.setattr(functional._Bf16Store, "cast_multi", staticmethod(cast_multi_like))
    for store, planes in ((functional._PlaneStore, True), (functional._Bf16Store, False)):
        monkeypatch.setattr(store, "cast", staticmethod(cast_like(planes)))
        monkeypatch.setattr(store, "cast_t", staticmethod(cast_t_like(planes)))
        monkeypatch.setattr(store, "gemm", staticmethod(gemm_like(planes)))
        monkeypatch.setattr(store, "wgrad", staticmethod(wgrad_like))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)

    rng = np.random.default_rng(0)
    L = len(ln) - 1
    params = []
    for i in range(L):
        params += [torch.tensor((rng.standard_normal((ln[i + 1], ln[i])) * np.sqrt(2 / (ln[i] + ln[i + 1]))).astype(np.float32), requires_grad=True),
                   torch.tensor((rng.standard_normal(ln[i + 1]) * 0.1).astype(np.float32), requires_grad=True)]
    acts = tuple([ops.ACT_RELU] * (L - 1) + [ops.ACT_SIGMOID if ln[-1] == 1 else ops.ACT_RELU])
    x = torch.tensor(rng.random((B, ln[0])).astype(np.float32), requires_grad=True)
    dy = torch.tensor(rng.standard_normal((B, ln[-1])).astype(np.float32))
    y = MLPFunction.apply(x, acts, None, ops.arith_code(arith), *params)
    y.backward(dy)
    got = [y.detach(), x.grad] + [p.grad for p in params]
    ps = [p.detach().clone().requires_grad_(True) for p in params]
    xr = x.detach().clone().requires_grad_(True)
    h = xr
    for i in range(L):
        h = act(h @ ps[2 * i].t() + ps[2 * i + 1], acts[i])
    h.backward(dy)
    ref = [h.detach(), xr.grad] + [p.grad for p in ps]
    for k, (a, b) in enumerate(zip(got, ref)):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-3, atol=2e-3 * float(b.abs().max()), err_msg=str((ln, k)))


def test_bench_node_state_never_raises_and_reads_the_sysfs_format(tmp_path, monkeypatch):
    """bench.py's `box.node` diagnostics: on a machine without GPUs (here) it returns error notes instead of raising; the shader-clock parser
    reads the `pp_dpm_sclk` format the GPU boxes expose (captured in profiles/round4/visit_slow_box/sysfs_clocks.txt)."""
    import glob as _glob
    import bench
    out = bench.node_state()
    assert isinstance(out, dict)
    d = tmp_path / "card0" / "device"
    d.mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2393Mhz *\n2: 2400Mhz\n")
    d2 = tmp_path / "card8" / "device"
    d2.mkdir(parents=True)
    (d2 / "pp_dpm_sclk").write_text("S: 94Mhz *\n0: 500Mhz\n1: 2400Mhz\n")
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: sorted(real(str(tmp_path / "card*" / "device" / "pp_dpm_sclk"))) if "pp_dpm_sclk" in pat else real(pat))
    out = bench.node_state()
    assert out["sclk_mhz_all_cards"] == [2393, 94] and out["cards_at_high_clock"] == 1


def test_planes_kernel_preconditions_are_host_side():
    """dlrm_gemm_bf16x6_supported answers without touching a GPU (the host mirror asks it for every layer of every tower)."""
    from dlrm_amd import ops
    assert ops.round_x6_k(479) == 480 and ops.round_x6_k(16) == 16 and ops.round_x6_k(13) == 16
    assert ops.gemm_bf16x6_ok(65536, 1024, 480) and ops.gemm_bf16x6_ok(65536, 512, 16) and ops.gemm_bf16x6_ok(256, 192, 64)
    assert not ops.gemm_bf16x6_ok(65536, 128, 256)          # N < 192
    assert not ops.gemm_bf16x6_ok(128, 512, 256)            # M < 256
    assert not ops.gemm_bf16x6_ok(65536, 512, 24)           # K % 16
    assert not ops.gemm_bf16x6_ok(65536, 1, 256)


def test_bench_calibration_failures_do_not_cost_the_headline(monkeypatch, capsys):
    """measure_box_or_none swallows any probe failure (reported on stderr); merge_box tolerates a probe present on one side only."""
    import bench

    def boom(device, quick=False):
        raise RuntimeError("HIP out of memory")
    monkeypatch.setattr(bench, "measure_box", boom)
    assert bench.measure_box_or_none(torch.device("cpu")) is None
    assert "box calibration failed" in capsys.readouterr().err
    b0 = {"cu_count": 256, "mfma_f32_tflops": 156.0, "hbm_copy_gbps": 6200.0, "hbm_gather_gbps": 6100.0}
    b1 = {"cu_count": 256, "mfma_f32_tflops": 158.0, "hbm_copy_gbps": 6300.0, "hbm_gather_error": "OOM"}
    box = bench.merge_box(b0, b1)
    assert box["mfma_f32_tflops"] == 157.0 and box["hbm_copy_gbps"] == 6250.0 and box["hbm_gather_gbps"] == 6100.0
    assert box["hbm_gather_error"] == "OOM" and box["before"]["hbm_gather_gbps"] == 6100.0 and "hbm_gather_gbps" not in box["after"]


def test_bench_help_renders():
    """`python bench.py --help` (argparse expands %-formats in help strings: a bare "1.7 % (" in one of them made --help raise for two rounds)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--mlp-arith" in r.stdout, r.stderr[-500:]
