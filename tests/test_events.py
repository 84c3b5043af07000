"""Event-window builder.  CPU: the oracle against fixtures produced by the reference's own
ERPCParser.__getitem__ (oracle/make_golden_events.py).  GPU: ev2h_event_window_* against the fixtures and the oracle,
bit-exact (indices, counts AND the float32 per-pixel sums, which np.add.at accumulates in stream order)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import event_window_oracle as EW

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "events_[0-9]*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_oracle_matches_reference_fixture(path):
    g = np.load(path)
    for w in range(int(g["nwin"])):
        data, table, idx = EW.build_window(g[f"raw{w}"], g[f"idx{w}"].astype(np.int64))
        assert np.array_equal(table, g[f"table{w}"])
        assert np.array_equal(data.numpy(), g[f"data{w}"])
        assert data.shape == (5, 2048) and float(data[:3].abs().max()) <= 1.0


def test_oracle_draws_like_reference():
    g = np.load(FIX[0])
    np.random.seed(100)
    data, _, idx = EW.build_window(g["raw0"])
    assert np.array_equal(idx, g["idx0"]) and np.array_equal(data.numpy(), g["data0"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_gpu_builder_matches_reference_fixture(path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.events import EventWindowBuilder
    g = np.load(path)
    nw = int(g["nwin"])
    wins = [g[f"raw{w}"] for w in range(nw)]
    bld = EventWindowBuilder("cuda:0")
    table, counts = bld.accumulate(wins)
    idx = np.stack([g[f"idx{w}"] for w in range(nw)])
    out = bld.sample(table, counts, idx).cpu().numpy()
    for w in range(nw):
        ref_t = g[f"table{w}"]
        assert int(counts[w]) == ref_t.shape[0]
        got_t = table[w, :ref_t.shape[0], :5].cpu().numpy()
        assert np.array_equal(got_t, ref_t.astype(np.float32)), f"window {w}: unique-pixel table differs"
        assert np.array_equal(out[w], g[f"data{w}"]), f"window {w}: normalised tensor differs"


@pytest.mark.gpu
@pytest.mark.parametrize("n_ev,seed", [(2048, 3), (9000, 4), (32768, 5), (1, 6)])
def test_gpu_builder_matches_oracle_dense_windows(n_ev, seed):
    """Heavier windows: many events per pixel (order-sensitive float32 sums), maximum size, a single event."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.events import EventWindowBuilder
    wins = []
    for k in range(3):
        s = EW.synth_event_stream(n_ev, seed * 10 + k).astype(np.float64)
        s[:, 2] *= 1e-3                                        # us -> ms, as EvalutaionStream.get_event does
        if k == 1:
            s[:, 0] = np.floor(s[:, 0] / 16) + 100             # squeeze into few pixels: long runs per pixel
            s[:, 1] = np.floor(s[:, 1] / 16) + 100
        wins.append(s)
    bld = EventWindowBuilder("cuda:0", n_events=512)
    table, counts = bld.accumulate(wins)
    rng = np.random.RandomState(seed)
    for w, raw in enumerate(wins):
        xi, yi, t_avg, p_evn, n_evn = EW.accumulate_pixels(raw)
        M = xi.shape[0]
        assert int(counts[w]) == M
        got = table[w, :M, :5].cpu().numpy()
        ref = np.stack([xi, yi, t_avg, p_evn, n_evn], 1).astype(np.float32)
        assert np.array_equal(got, ref)
    idx = np.stack([rng.randint(0, int(counts[w]), 512) for w in range(3)])
    out = bld.sample(table, counts, idx).cpu().numpy()
    for w, raw in enumerate(wins):
        if int(counts[w]) < 2:
            continue                                            # a single pixel normalises to 0/0 in the reference too
        ref, _, _ = EW.build_window(raw, idx[w], n_events=512)
        assert np.array_equal(out[w], ref.numpy())


@pytest.mark.gpu
def test_gpu_builder_rejects_oversized_window():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.events import EventWindowBuilder
    s = EW.synth_event_stream(40000, 1).astype(np.float64)
    bld = EventWindowBuilder("cuda:0")
    table, counts = bld.accumulate([s])
    assert int(counts[0]) == -1
    with pytest.raises(RuntimeError):
        bld.sample(table, counts)


# ------------------------------------------------------------------------------------------------ Ev2Hands-S variant (erpc.py)
FIX_S = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "events_s_*.npz")))


@pytest.mark.parametrize("path", FIX_S, ids=[os.path.basename(p)[:-4] for p in FIX_S])
def test_oracle_s_matches_reference_fixture(path):
    """oracle build_window_s against outputs of the reference's own Ev2HandSDataset.__getitem__ (oracle/make_golden_events_s.py)."""
    g = np.load(path)
    sampling = bool(int(g["sampling"]))
    for w in range(int(g["nwin"])):
        ev, lab, table, table_lab, _ = EW.build_window_s(g[f"rows{w}"], g[f"idx{w}"].astype(np.int64), sampling=sampling)
        assert np.array_equal(ev.numpy(), g[f"events{w}"]) and np.array_equal(lab.numpy(), g[f"labels{w}"])
        assert np.array_equal(table, g[f"table{w}"]) and np.array_equal(table_lab, g[f"table_lab{w}"])
        assert ev.shape == (5, 2048) and float(table[0, 2]) == 0.0 and (np.diff(table[:, 2]) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIX_S, ids=[os.path.basename(p)[:-4] for p in FIX_S])
def test_gpu_builder_s_matches_reference_fixture(path):
    """ev2h_event_window_build(raw_time) + _timesort + _sample against the reference's own dataset items: bit-exact events,
    labels and time-ordered tables, with sampling on and off."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.events import EventWindowBuilderS
    g = np.load(path)
    sampling = bool(int(g["sampling"]))
    nw = int(g["nwin"])
    bld = EventWindowBuilderS("cuda:0")
    out = bld([g[f"rows{w}"] for w in range(nw)], sampling=sampling, sample_idx=[g[f"idx{w}"] for w in range(nw)] if not sampling
              else np.stack([g[f"idx{w}"] for w in range(nw)]))
    for w in range(nw):
        M = g[f"table{w}"].shape[0]
        assert np.array_equal(bld.table[w, :M, :5].cpu().numpy(), g[f"table{w}"]), f"window {w}: time-ordered table differs"
        assert np.array_equal(bld.table_labels[w, :M].cpu().numpy(), g[f"table_lab{w}"])
        assert np.array_equal(out["events"][w].cpu().numpy(), g[f"events{w}"]), f"window {w}: events differ"
        assert np.array_equal(out["class_logits"][w].cpu().numpy(), g[f"labels{w}"])


@pytest.mark.gpu
def test_gpu_builder_s_with_tied_times_keeps_pixel_order():
    """Real tables have several events per microsecond: per-pixel mean times tie exactly and np.argsort leaves their order
    undefined.  The kernel (like the oracle) keeps tied pixels in pixel order; everything else must still match."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.events import EventWindowBuilderS
    rows = EW.synth_s_rows(4096, 21)
    rows[:, 2] = np.floor(rows[:, 2] / 4000.0) * 4000.0 + 1e9            # 4 us resolution, 1e9 ns offset: many exact ties
    wins = [rows[:2048], rows[2048:]]
    idx = np.stack([np.arange(2048) % 1500, (np.arange(2048) * 7) % 1500])
    bld = EventWindowBuilderS("cuda:0")
    out = bld(wins, sampling=True, sample_idx=idx)
    for w in range(2):
        ev, lab, table, table_lab, _ = EW.build_window_s(wins[w], idx[w], sampling=True)
        assert len(np.unique(table[:, 2])) < table.shape[0] - 50        # the case really has ties
        assert np.array_equal(bld.table[w, :table.shape[0], :5].cpu().numpy(), table)
        assert np.array_equal(out["events"][w].cpu().numpy(), ev.numpy()) and np.array_equal(out["class_logits"][w].cpu().numpy(), lab.numpy())
