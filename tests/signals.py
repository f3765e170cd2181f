"""Test infrastructure: signal-shaped I/Q frames (numpy only).

Every reduced-precision bar of rounds 1-2 was measured on N(0, sigma) noise frames, where the softmax is near-uniform
and margins are tiny.  The reference's frames are bursts of the three modulations of CNN.ipynb cell 2,
`mods_chosen = ['WBFM', 'AM-SSB', 'GFSK']` (class 0, 1, 2 in that order: cell 4 takes `mods_chosen.index`), cut from
RadioML2016.10a (128 complex samples per frame, 8 samples per symbol for the digital modes, SNR 2..18 dB in cell 2's
`snrs_chosen`); the 16 bundled Q6.12 frames are the only real ones (|x| <= 0.02).  This module synthesises frames of the
same KIND at the same scale -- it is not the dataset and makes no claim on accuracy: it gives the kernels inputs with
structure (constant-envelope phase modulations, a one-sided analytic spectrum) and, with the bundled trained weights,
decisive class margins.  Nothing here is product code.

    frames, labels, snrs = modulated_frames(n, seed)      # (n,2,128) float32, (n,) int, (n,) int dB
"""
import numpy as np

MODS = ("WBFM", "AM-SSB", "GFSK")            # class order of CNN.ipynb cell 2
SNRS = (2, 4, 6, 8, 10, 12, 14, 16, 18)      # snrs_chosen of the same cell
_L = 128


def _lowpass_noise(rng, n, length, cutoff):
    """Band-limited Gaussian 'audio': white noise through a brick-wall low-pass at `cutoff` (cycles per sample)."""
    pad = 4 * length
    spec = np.fft.rfft(rng.standard_normal((n, pad)), axis=1)
    f = np.fft.rfftfreq(pad)
    spec[:, f > cutoff] = 0.0
    m = np.fft.irfft(spec, n=pad, axis=1)[:, pad // 2: pad // 2 + length]
    return m / np.maximum(np.abs(m).max(axis=1, keepdims=True), 1e-12)


def _wbfm(rng, n):
    m = _lowpass_noise(rng, n, _L, 0.04)
    dev = rng.uniform(0.05, 0.2, (n, 1))                       # peak deviation, cycles per sample
    return np.exp(2j * np.pi * np.cumsum(dev * m, axis=1))


def _am_ssb(rng, n):
    m = _lowpass_noise(rng, n, _L, 0.08)
    spec = np.fft.fft(m, axis=1)
    spec[:, _L // 2 + 1:] = 0.0                                # analytic signal: upper sideband only
    spec[:, 1:_L // 2] *= 2.0
    x = np.fft.ifft(spec, axis=1)
    return x / np.maximum(np.abs(x).max(axis=1, keepdims=True), 1e-12)


def _gfsk(rng, n, sps=8, bt=0.35, h=0.5):
    nsym = _L // sps + 4
    bits = rng.integers(0, 2, (n, nsym)) * 2.0 - 1.0
    nrz = np.repeat(bits, sps, axis=1)
    t = np.arange(-2 * sps, 2 * sps + 1) / sps
    g = np.exp(-2 * (np.pi * bt * t) ** 2 / np.log(2))
    g /= g.sum()
    f = np.apply_along_axis(lambda r: np.convolve(r, g, mode="same"), 1, nrz)
    off = rng.integers(0, sps, n)                              # symbol timing offset
    idx = off[:, None] + np.arange(_L)[None, :] + sps
    f = np.take_along_axis(f, idx, axis=1)
    return np.exp(1j * np.pi * h * np.cumsum(f, axis=1) / sps)


def modulated_frames(n, seed=2016, rms_level=7.8e-3, snrs=SNRS):
    """n frames, classes and SNRs drawn uniformly: random carrier phase, a small carrier offset, complex AWGN at the
    frame's SNR, then every frame scaled to a complex rms of `rms_level` +- 10 % -- the level of the bundled frames
    (their constant-envelope bursts have |I + jQ| = 0.0076 .. 0.0078, their peaks reach 0.02; RadioML2016.10a
    normalises each vector's energy)."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, len(MODS), n)
    snr_db = rng.choice(np.asarray(snrs), n)
    x = np.empty((n, _L), np.complex128)
    for c, gen in enumerate((_wbfm, _am_ssb, _gfsk)):
        sel = np.nonzero(labels == c)[0]
        if len(sel):
            x[sel] = gen(rng, len(sel))
    cfo = rng.uniform(-0.01, 0.01, (n, 1))
    x *= np.exp(1j * (2 * np.pi * cfo * np.arange(_L)[None, :] + rng.uniform(0, 2 * np.pi, (n, 1))))
    p_sig = (np.abs(x) ** 2).mean(axis=1, keepdims=True)
    sigma = np.sqrt(p_sig / (2 * 10.0 ** (snr_db[:, None] / 10.0)))
    x = x + sigma * (rng.standard_normal((n, _L)) + 1j * rng.standard_normal((n, _L)))
    rms = np.sqrt((np.abs(x) ** 2).mean(axis=1, keepdims=True))
    x *= rms_level * rng.uniform(0.9, 1.1, (n, 1)) / rms
    frames = np.stack([x.real, x.imag], axis=1).astype(np.float32)
    return frames, labels.astype(np.int32), snr_db.astype(np.int32)
