"""Seeded synthetic I/Q: one carrier per receiver at its tune offset + 1 kHz,
log-uniform amplitudes in [2^18, 2^28], Gaussian noise sigma 2^16, quantised to int32-valued doubles like
the reference's UDP readers produce (quisk.c:3516-3518)."""
import numpy as np


def tune_offsets(n_rx, fs):
    """Receiver tune offsets: uniformly spaced over +-0.45 fs, integer Hz, never 0."""
    if n_rx == 1:
        return np.array([int(0.1 * fs)], dtype=np.int64)
    t = np.round(np.linspace(-0.45 * fs, 0.45 * fs, n_rx)).astype(np.int64)
    t[t == 0] = 1
    return t


def mode_mix(n_rx, mix="ssb"):
    """rx_mode_type per receiver: 'ssb' = all USB; 'mixed' = 50 % SSB (USB/LSB), 25 % AM, 25 % FM."""
    if mix == "ssb":
        return np.full(n_rx, 3, dtype=np.int64)
    pat = np.array([3, 4, 2, 5], dtype=np.int64)
    return pat[np.arange(n_rx) % 4]


def make_iq(n, fs, tunes, modes, seed, start=0, chunk=1 << 18):
    """n complex128 samples starting at sample index `start` (deterministic in (seed, start) for chunk-aligned starts)."""
    tunes = np.asarray(tunes, dtype=np.float64)
    modes = np.asarray(modes)
    rng0 = np.random.default_rng(seed)
    K = len(tunes)
    amp = 2.0 ** rng0.uniform(18, 28, K) / max(1.0, K / 8.0)
    ph0 = rng0.uniform(0, 2 * np.pi, K)
    out = np.empty(n, dtype=np.complex128)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        t = (start + c0 + np.arange(m, dtype=np.float64)) / fs
        acc = np.zeros(m, dtype=np.complex128)
        for j in range(K):
            car = np.exp(1j * (2 * np.pi * (tunes[j] + 1000.0) * t + ph0[j]))
            if modes[j] == 4:
                car = car * (1 + 0.5 * np.sin(2 * np.pi * 1000.0 * t))
            elif modes[j] in (5, 13):
                car = car * np.exp(1j * 3.0 * np.sin(2 * np.pi * 1000.0 * t))
            acc += amp[j] * car
        rng = np.random.default_rng([seed, (start + c0) // chunk])
        acc += 2.0 ** 16 * (rng.standard_normal(m) + 1j * rng.standard_normal(m)) / np.sqrt(2)
        out[c0:c0 + m] = np.clip(np.round(acc.real), -2147483647, 2147483647) + 1j * np.clip(np.round(acc.imag), -2147483647, 2147483647)
    return out


from .rxfilters import DEFAULT_BW  # noqa: E402,F401  (default filter widths, quisk.py:3579-3588)
