"""CPU-side checks: column layout, seed-parity of construction, ABI surface, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from satrans_amd import DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features, get_feature_names
from satrans_amd import native
from tests.helpers import ALL_CASES, Case, build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_feature_index_layout():
    cols = [SparseFeat("a", 10, 8), DenseFeat("p", 2), SparseFeat("b", 5, 8),
            VarLenSparseFeat(SparseFeat("h", 7, 8), maxlen=3, length_name="h_len")]
    idx = build_input_features(cols + cols)          # main.py passes linear + dnn columns: duplicates are skipped
    assert list(idx.items()) == [("a", (0, 1)), ("p", (1, 3)), ("b", (3, 4)), ("h", (4, 7)), ("h_len", (7, 8))]
    assert get_feature_names(cols) == ["a", "p", "b", "h", "h_len"]
    assert SparseFeat("z", 16, "auto").embedding_dim == 12


@pytest.mark.parametrize("name", ALL_CASES)
def test_construction_is_bit_identical_to_reference(name):
    """Same seed -> same state_dict as the reference (same generator draws in the same order)."""
    c = Case(name)
    model = build_model(c, "cpu")
    mine, want = model.state_dict(), c.tensors("param")
    assert set(mine) == set(want)
    for k in want:
        assert torch.equal(mine[k], want[k]), k
    # storage: tables are views of one arena, trainables views of one flat buffer, and they stay so after a move
    assert model.embedding_dict[c.meta["fields"][0]].weight.data_ptr() == model.embedding_arena.data_ptr()
    model.to("cpu")
    assert model.embedding_dict[c.meta["fields"][0]].weight.data_ptr() == model.embedding_arena.data_ptr()
    total = sum(v for v in c.meta["vocab"])
    assert model.embedding_arena.shape == (total, c.meta["D"])


@pytest.mark.parametrize("name", ["aliccp_sota", "alimama_sota_pos", "small_k", "small_none", "small_onlyemb", "small_gate",
                                  "small_bilinear", "small_multidomain"])
def test_trainable_set_matches_reference_gradients(name):
    """The tensors the fused Adam steps are exactly the ones that receive a gradient in the reference."""
    c = Case(name)
    model = build_model(c, "cpu")
    mine = set(model._trainable_flat()) | {f"embedding_dict.{n}.weight" for n in model._table_order()}
    assert mine == set(c.arrays("grad"))


def test_forward_on_cpu_fails_loudly():
    c = Case("small_q")
    model = build_model(c, "cpu")
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        model(c.X)


def test_bad_arguments_raise_like_the_reference():
    cols = [SparseFeat("a", 10, 8), SparseFeat("d", 4, 8)]
    from satrans_amd import SATrans
    with pytest.raises(ValueError, match="integer multiple of head_num"):
        SATrans(cols, cols, ["d"], [3], domain_att_layer_num=1, att_head_num=3, flag="sota")
    with pytest.raises(ValueError, match="head_num must be"):
        SATrans(cols, cols, ["d"], [3], domain_att_layer_num=1, att_head_num=0, flag="sota")
    mixed = [SparseFeat("a", 10, 8), SparseFeat("d", 4, 4)]
    with pytest.raises(ValueError, match="must be same"):
        SATrans(mixed, mixed, ["d"], [3], domain_att_layer_num=1, att_head_num=2, flag="sota")


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed) and exports exactly what include/satrans_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "satrans_hip.h")).read()
    declared = set(re.findall(r"\b(satrans_[a-z0-9_]+)\s*\(", header))
    declared -= {"satrans_layer_desc", "satrans_adam_hparams"}
    assert os.path.exists(native.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(native.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(handle, sym), f"{sym} declared in the header but not exported"
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    lib = native.lib()
    assert lib.satrans_abi_version() == native.ABI_VERSION
    # argument validation works without touching a device
    assert lib.satrans_gather_fwd(None, None, None, None, 0, 0, 1, 1, 32, None, None, None, None) == -1
    assert b"null pointer" in lib.satrans_last_error()
    assert ctypes.sizeof(native.LayerDesc) == 8 * 4 + 4 + 4 + 4 + 4 + 8 + 18 * 8   # mirrors satrans_layer_desc


def test_dropout_mask_statistics():
    """The counter-based mask keeps ~90 % and decorrelates sites, layers and steps."""
    from oracle.satrans_oracle import dropout_keep
    b = np.arange(512, dtype=np.uint32)[:, None]
    e = np.arange(608, dtype=np.uint32)[None, :]
    k0 = dropout_keep(1021, 1, 0, 0, b, e, 0.1)
    k1 = dropout_keep(1021, 1, 0, 1, b, e, 0.1)
    k2 = dropout_keep(1021, 2, 0, 0, b, e, 0.1)
    assert abs(k0.mean() - 0.9) < 0.005
    for other in (k1, k2):
        assert abs((k0 & other).mean() - 0.81) < 0.01
    # element by element: every one of the 608 positions keeps 90 % of 16,384 samples (sigma 0.0023), neighbours inside a
    # block of four (one LCG step apart), across blocks and 38 apart (the same field position of the next head) are independent
    b = np.arange(16384, dtype=np.uint32)[:, None]
    k = dropout_keep(7, 3, 2, 2, b, e, 0.1)
    rate = k.mean(0)
    assert np.abs(rate - 0.9).max() < 5 * 0.0023, float(np.abs(rate - 0.9).max())
    for lag in (1, 2, 3, 4, 8, 38):
        both = (k[:, :-lag] & k[:, lag:]).mean(0)
        assert np.abs(both - 0.81).max() < 5 * 0.0031, (lag, float(np.abs(both - 0.81).max()))
        drop2 = (~k[:, :-lag] & ~k[:, lag:]).mean(0)
        assert np.abs(drop2 - 0.01).max() < 5.5 * 0.00078, (lag, float(np.abs(drop2 - 0.01).max()))
    # ... and so are consecutive samples at the same element (the per-sample key is a finalised hash of the sample index)
    both = (k[:-1] & k[1:]).mean(0)
    assert np.abs(both - 0.81).max() < 5 * 0.0031


def test_division_by_a_constant_through_its_double_reciprocal_is_the_fp32_quotient():
    """embed_adam.hip: adam_core evaluates sqrt(v) / bc2_sqrt as float32(float64(s) * (1 / float64(c))).  The header gives the
    argument why that is the correctly rounded fp32 quotient; this checks it on 40 M random pairs over the ranges the
    optimizer sees (s = sqrt(v) from 1e-22 to 1e3, c = sqrt(1 - beta2^t) in (0.03, 1]) plus adversarial neighbours."""
    rng = np.random.RandomState(0)
    for _ in range(4):
        s = np.exp(rng.uniform(np.log(1e-22), np.log(1e3), size=10_000_000)).astype(np.float32)
        c = rng.uniform(0.03, 1.0, size=10_000_000).astype(np.float32)
        want = s / c                                                     # IEEE fp32 division
        got = (s.astype(np.float64) * (1.0 / c.astype(np.float64))).astype(np.float32)
        assert np.array_equal(want, got)
    # quotients that land next to a rounding boundary: s = nextafter(q * c) for q with a long run of ones / zeros
    q = (np.float32(1.0) + np.arange(1, 200001, dtype=np.float32) * np.float32(2.0 ** -23))
    c = rng.uniform(0.03, 1.0, size=q.size).astype(np.float32)
    for s in (q * c, np.nextafter(q * c, np.float32(0)), np.nextafter(q * c, np.float32(10))):
        s = s.astype(np.float32)
        assert np.array_equal(s / c, (s.astype(np.float64) * (1.0 / c.astype(np.float64))).astype(np.float32))


def test_device_metrics_equal_sklearn():
    """satrans_amd/device_metrics.py (the per-step train metrics of fit, kept on the device) against the sklearn functions the
    reference calls, on float32 probabilities with many ties and with saturated values."""
    from sklearn.metrics import accuracy_score, log_loss, mean_squared_error, roc_auc_score
    from satrans_amd import device_metrics as DM
    rng = np.random.RandomState(3)
    for n, levels in ((8192, None), (4096, 17), (300, 3)):
        p = rng.rand(n).astype(np.float32)
        if levels:
            p = (np.floor(p * levels) / levels).astype(np.float32)           # heavy ties
        p[:5] = [0.0, 1.0, 1e-30, 1 - 1e-7, 0.5]
        y = (rng.rand(n) < 0.3).astype(np.float32)
        yt, pt = torch.from_numpy(y), torch.from_numpy(p)
        p64 = p.astype("float64")
        assert float(DM.log_loss(yt, pt)) == pytest.approx(log_loss(y, p64), rel=1e-12)
        assert float(DM.roc_auc(yt, pt)) == pytest.approx(roc_auc_score(y, p64), rel=1e-12, abs=1e-15)
        assert float(DM.mse(yt, pt)) == pytest.approx(mean_squared_error(y, p64), rel=1e-12)
        assert float(DM.accuracy(yt, pt)) == pytest.approx(accuracy_score(y, np.where(p64 > 0.5, 1, 0)), rel=1e-12)
        # the per-scenario report of reference main.py:355-374
        dom = rng.randint(1, 4, size=len(y))
        auc, per, loss = DM.per_domain_auc(yt, pt, torch.from_numpy(dom))
        assert auc == pytest.approx(roc_auc_score(y, p64), rel=1e-12)
        assert sorted(per) == [1, 2, 3]
        for i in per:
            assert per[i] == pytest.approx(roc_auc_score(y[dom == i], p64[dom == i]), rel=1e-12, abs=1e-15)
        want = torch.nn.functional.binary_cross_entropy(torch.tensor(p64), torch.tensor(y).double()).item()
        assert loss == pytest.approx(want, rel=1e-12)


def test_host_batch_feeder_yields_every_row_once_in_order(tmp_path):
    """satrans_amd/pipeline.py on the CPU device (no pinning, same staging logic): batches of a memory-mapped column set, in
    order and along a permutation, with a ragged last batch, an integer id matrix and a dense block."""
    from satrans_amd.inputs import PackedInput
    from satrans_amd.pipeline import HostBatchFeeder, load_npy_columns
    rng = np.random.RandomState(1)
    N, C = 1003, 5
    cols = {f"c{i}": rng.randint(0, 1 << 30, size=N).astype(np.int64) for i in range(C)}
    for k, v in cols.items():
        np.save(tmp_path / f"{k}.npy", v)
    mm = load_npy_columns(str(tmp_path), list(cols))
    ids = np.stack([mm[k] for k in cols], axis=1)
    y = rng.rand(N).astype(np.float32)
    dense = rng.rand(N, 2).astype(np.float32)
    for order in (None, rng.permutation(N)):
        got_x, got_y, got_d = [], [], []
        feeder = HostBatchFeeder(ids, y, 128, "cpu", order, dense)
        assert len(feeder) == 8
        for xb, yb in feeder:
            assert isinstance(xb, PackedInput) and xb.ids.dtype == torch.int64
            got_x.append(xb.ids.clone()); got_d.append(xb.dense.clone()); got_y.append(yb.clone())
        idx = np.arange(N) if order is None else order
        assert torch.equal(torch.cat(got_x), torch.from_numpy(ids[idx]))
        assert torch.equal(torch.cat(got_y), torch.from_numpy(y[idx]))
        assert torch.equal(torch.cat(got_d), torch.from_numpy(dense[idx]))
    with pytest.raises(FileNotFoundError):
        from satrans_amd.pipeline import load_h5_columns
        load_h5_columns("/nonexistent.h5", "ctr_train", ["101"])


def test_hdf5_columns_without_h5py(tmp_path):
    """The reference reads its datasets from HDF5 with h5py (utils.py:22-30, 266-278); this image has none, so
    satrans_amd/h5lite.py parses the subset `h5py.File(path, 'w')` + `f[name] = array` writes.  Pinned on a file that
    tests/h5_fixture.py writes byte by byte from the format specification (groups `ctr_train` / `ctr_test` with 22 datasets each
    - three symbol-table nodes under the group's B-tree -, root-level datasets as in alimama.h5, int64 / int32 / float64 /
    two-dimensional / empty datasets, object headers with continuation blocks), and fed to the streaming input pipeline."""
    from tests.h5_fixture import write_h5
    from satrans_amd.h5lite import H5File
    from satrans_amd.pipeline import HostBatchFeeder, load_h5_columns
    rng = np.random.RandomState(3)
    cols = ['101', '121', '122', '124', '125', '126', '127', '128', '129', '205', '206', '207', '210', '216', '508', '509', '702',
            '853', '301', 'click', 'purchase']
    train = {c: rng.randint(0, 1 << 40 if c == '205' else 1000, size=777).astype(np.int64) for c in cols}
    test = {c: rng.randint(0, 1000, size=130).astype(np.int32) for c in cols}
    train['10914_3'] = rng.randint(0, 50, size=(777, 3)).astype(np.int32)      # the history columns are [N, k] (dataset_processing:237)
    price = rng.rand(55)
    path = str(tmp_path / "alicpp.h5")
    write_h5(path, {"ctr_train": train, "ctr_test": test, "price": price, "empty": np.zeros(0, np.int64)})
    f = H5File(path)
    assert f.keys("/") == ["ctr_test", "ctr_train", "empty", "price"]
    assert f.keys("ctr_train") == sorted(train)
    got = load_h5_columns(path, "ctr_train", cols + ['10914_3'])
    for c, want in train.items():
        assert got[c].dtype == want.dtype and np.array_equal(got[c], want), c
        assert isinstance(got[c], np.memmap)                                   # nothing copied until a batch is gathered
    got = load_h5_columns(path, "ctr_test")                                    # every member, as utils.loadh52df does
    assert sorted(got) == sorted(test) and all(np.array_equal(got[c], test[c]) for c in test)
    root = load_h5_columns(path, None, ["price", "empty"], mmap=False)
    assert root["price"].dtype == np.float64 and np.array_equal(root["price"], price) and root["empty"].shape == (0,)
    with pytest.raises(KeyError):
        load_h5_columns(path, "ctr_train", ["no_such_column"])
    # the columns feed the double-buffered host pipeline as they are (memory-mapped int64 ids beyond 2**24)
    tr = load_h5_columns(path, "ctr_train", cols[:19])
    ids = np.stack([tr[c] for c in cols[:19]], axis=1)
    seen = torch.cat([xb.clone() for xb, _ in HostBatchFeeder(ids, None, 256, "cpu")])
    assert torch.equal(seen, torch.from_numpy(ids))
    # outside the subset: a chunked dataset must be named as such, not misread
    blob = bytearray(open(path, "rb").read())
    at = blob.find(bytes([3, 1]) + (blob.find(train['101'].tobytes())).to_bytes(8, "little"))
    assert at > 0
    blob[at + 1] = 2                                                           # layout class 2 = chunked
    bad = str(tmp_path / "chunked.h5")
    open(bad, "wb").write(bytes(blob))
    with pytest.raises(NotImplementedError, match="chunked"):
        load_h5_columns(bad, "ctr_train", ["101"])


def test_hdf5_reader_on_a_file_written_by_libhdf5():
    """A file written by the real HDF5 library, where this image happens to ship one: scipy's MATLAB v7.3 test file (an HDF5
    file behind a 512-byte user block, written by HDF5 1.6: version-0 superblock at offset 512, non-zero base address, old
    style root group, version-1 data layout message)."""
    import os
    import scipy.io
    path = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if not os.path.exists(path):
        pytest.skip("scipy's test data are not installed")
    from satrans_amd.h5lite import H5File
    f = H5File(path)
    assert f.base == 512 and f.keys("/") == ["testdouble"]
    a = f.dataset("/testdouble")
    assert a.dtype == np.float64 and np.allclose(np.asarray(a).ravel(), np.pi / 4 * np.arange(9))


def test_speculative_epoch_order_is_the_order_the_loader_draws():
    """fit() draws the NEXT epoch's permutation while the device drains the current one (basemodel._speculate_epoch_order).  The
    speculation must leave the global generator untouched, must be what `_epoch_order` would have drawn (= what the reference's
    DataLoader(shuffle=True) draws: one base-seed draw, one sampler-seed draw, randperm on that seed), and must be dropped when
    something consumes or reseeds the generator in between (a callback)."""
    model = build_model(Case("small_qkv"), "cpu")
    n = 1000

    def loader_order():                                    # torch.utils.data.DataLoader(range(n), shuffle=True) order, first epoch
        import torch.utils.data as tud
        return torch.tensor(list(iter(tud.DataLoader(range(n), batch_size=n, shuffle=True)))[0])

    torch.manual_seed(77)
    want1 = loader_order()
    want2 = loader_order()
    torch.manual_seed(77)
    got1 = model._epoch_order(n, True, on_host=True)
    state = torch.get_rng_state()
    model._speculate_epoch_order(n)
    assert torch.equal(torch.get_rng_state(), state), "the speculation must leave the global generator as it was"
    spec_perm = model._order_spec[2]
    got2 = model._epoch_order(n, True, on_host=True)
    assert got2 is spec_perm, "the speculated permutation is the one used when nothing touched the generator"
    assert torch.equal(got1, want1) and torch.equal(got2, want2)
    # a callback reseeds between the epochs: the speculation misses and the order is the reseeded one
    model._speculate_epoch_order(n)
    torch.manual_seed(5)
    want3 = loader_order()
    torch.manual_seed(5)
    got3 = model._epoch_order(n, True, on_host=True)
    assert torch.equal(got3, want3) and model._order_spec is None


def test_host_randperm_is_torch_randperm_and_its_head_is_final_before_its_tail():
    """satrans_host_randperm (csrc/host_sampler.hip) must return torch.randperm's permutation bit for bit - it replaces the draw
    of the reference's RandomSampler - for seeds above 2**32 (torch seeds its Mersenne Twister with the low 32 bits), around the
    block size of its look-ahead, and for the trivial sizes; sampler.AsyncOrder must hand out the same rows, head first, and
    through the loader-order draws of `_epoch_order(lazy=True)` the order of DataLoader(shuffle=True)."""
    from satrans_amd import sampler
    assert sampler.native_ok(), "the library's pass no longer reproduces torch.randperm on this torch version"
    for seed in (0, 1, 5489, 2 ** 31 + 7, 2 ** 40 + 3, 2 ** 63 - 1):
        for n in (0, 1, 2, 3, 63, 64, 65, 66, 127, 128, 129, 130, 1000, 65537, 200001):
            gen = torch.Generator()
            gen.manual_seed(seed)
            want = torch.randperm(n, generator=gen)
            got = sampler.randperm(seed, n)
            assert got.dtype == torch.int64 and torch.equal(got, want), (seed, n)
    n = 300001
    gen = torch.Generator()
    gen.manual_seed(2 ** 35 + 11)
    want = torch.randperm(n, generator=gen)
    order = sampler.AsyncOrder(2 ** 35 + 11, n, "cpu")
    head = order.rows(0, 4096)                      # (possibly while the worker is still drawing the tail)
    assert torch.equal(head, want[:4096])
    assert torch.equal(order.rows(4096, 70000), want[4096:70000])
    assert torch.equal(order.full(), want) and torch.equal(order.rows(250000, n), want[250000:])
    assert torch.equal(sampler.AsyncOrder(3, n, "cpu", ready=want).rows(5, 9), want[5:9])     # a permutation drawn ahead
    # through the model: the two generator draws of the loader, then the permutation
    import torch.utils.data as tud
    model = build_model(Case("small_qkv"), "cpu")
    torch.manual_seed(123)
    loader = torch.tensor(list(iter(tud.DataLoader(range(5000), batch_size=5000, shuffle=True)))[0])
    torch.manual_seed(123)
    lazy = model._epoch_order(5000, True, on_host=True, lazy=True)
    assert torch.equal(lazy.rows(0, 100), loader[:100]) and torch.equal(lazy.full(), loader)
