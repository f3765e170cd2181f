"""The signal-shaped frame generator (tests/signals.py, test infrastructure) is deterministic and produces what it says:
the class order of CNN.ipynb cell 2, the bundled frames' level, constant-envelope WBFM / GFSK, a one-sided AM-SSB
spectrum.  With the bundled trained nets these frames -- unlike N(0, sigma) noise, which the 3-filter net labels class 1
every single time -- land in all three classes with decisive margins, which is what the reduced-precision label bars
need (tests/test_signal_frames_gpu.py)."""
import numpy as np

from conftest import load_deployed_npz
from oracle import oracle_np as O
from signals import MODS, SNRS, modulated_frames


def test_generator_is_deterministic_and_shaped_like_the_bundled_frames():
    a, la, sa = modulated_frames(600, seed=5)
    b, lb, sb = modulated_frames(600, seed=5)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(la, lb)
    assert a.shape == (600, 2, 128) and a.dtype == np.float32 and la.dtype == np.int32
    assert MODS == ("WBFM", "AM-SSB", "GFSK") and set(np.unique(la)) == {0, 1, 2} and set(np.unique(sa)) <= set(SNRS)
    c = a[:, 0].astype(np.float64) + 1j * a[:, 1]
    rms = np.sqrt((np.abs(c) ** 2).mean(axis=1))
    assert 0.9 * 7.8e-3 * 0.999 <= rms.min() and rms.max() <= 1.1 * 7.8e-3 * 1.001      # the bundled bursts: |I + jQ| ~ 0.0077
    assert np.abs(a).max() < 0.05
    assert not np.array_equal(a, modulated_frames(600, seed=6)[0])


def test_modulations_have_their_signatures():
    x, lab, snr = modulated_frames(3000, seed=9, snrs=(18,))
    c = x[:, 0].astype(np.float64) + 1j * x[:, 1]
    env = np.abs(c).std(axis=1) / np.abs(c).mean(axis=1)
    assert np.median(env[lab == 0]) < 0.2 and np.median(env[lab == 2]) < 0.2      # FM / GFSK: constant envelope (+ noise at 18 dB)
    assert np.median(env[lab == 1]) > 0.3                                           # AM-SSB: the envelope carries the message
    spec = np.abs(np.fft.fft(c[lab == 1], axis=1)) ** 2
    # analytic (upper-sideband) signal: the energy sits on one side of its carrier; the carrier offset is at most 0.01
    # cycles per sample, i.e. +- 1.3 bins, so bins 3..63 against 65..125
    assert np.median(spec[:, 3:64].sum(axis=1) / spec[:, 65:126].sum(axis=1)) > 5.0


def test_bundled_nets_spread_these_frames_over_all_classes_with_decisive_margins():
    x, _, _ = modulated_frames(4096, seed=2016)
    noise = (np.random.default_rng(0).standard_normal((4096, 2, 128)) * 5e-3).astype(np.float32)
    for name in ("3convmodrecnets_CNN2_0.5", "convmodrecnets_CNN2_0.5", "5convmodrecnets_CNN2_0.5"):
        w = [a for p in load_deployed_npz(name) for a in p]
        r = O.forward_deployed(x, *w, dtype=np.float64)
        counts = np.bincount(r["labels"], minlength=3)
        assert counts.min() >= 40, (name, counts)
        srt = np.sort(r["dense"], axis=1)
        assert np.median(srt[:, -1] - srt[:, -2]) > 0.05, name
    w = [a for p in load_deployed_npz("3convmodrecnets_CNN2_0.5") for a in p]
    assert len(np.unique(O.forward_deployed(noise, *w, dtype=np.float64)["labels"])) == 1      # why noise frames are not enough
