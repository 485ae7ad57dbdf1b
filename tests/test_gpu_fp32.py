"""The optional fp32 mode (BASELINE configs[4]; north_star: "within a stated fp32 tolerance for an optional fp32 mode").

What runs in float: the front end's half-band cascade (mixer product in double, then float samples, taps and accumulators),
the tiled decimating FIRs and the receive band-pass, and the display FFT (float butterflies and intermediate; window,
magnitudes and the average in double).  What stays double: the mixer phasor tables, the streams in HBM, the detectors,
the audio-rate interpolators and the AGC (quisk.c:2152 scales samples to 2^31: the AGC's thresholds need the range).

STATED TOLERANCE, measured against the fp64 oracle (the reference's arithmetic):
  audio in front of the AGC   RMS error per receiver <= 1e-5 x max(RMS of its own audio, 1e-3 x peak |sample| of the input
                              stream): 1e-5 relative for a receiver whose audio is within 60 dB of the stream's peak, and an
                              absolute floor of 1e-8 of that peak below (a float carries 24 bits: the wideband stream is
                              quantised to 6e-8 of its PEAK, whatever the receiver's own signal is)
  AGC                         bit-exact on identical input, as in fp64 mode (it runs in double)
  audio behind the AGC        reported, median over receivers <= 1e-5; a receiver whose gain ramp ends a sample apart from
                              the oracle's (the AGC ends ramps on a last-bit comparison, tests/_rxsetup.py 3.; float-sized
                              input differences decide it differently now and then) shows up to 5e-3 until the next ramp
  spectrum row                <= 0.01 dB on every pixel, S-meter <= 0.01 dB (measured: 1e-4 dB at 2^20 points)
"""
import numpy as np
import pytest

import _bind as B
from _rxsetup import PairRunner, rx_filter
from quisk_amd import Engine
from quisk_amd.synth import make_iq, tune_offsets, mode_mix

pytestmark = pytest.mark.gpu


def _rms_rel(got, want, floor=0.0):
    got = np.concatenate(got); want = np.concatenate(want)
    ref = max(float(np.sqrt(np.mean(np.abs(want) ** 2))), floor)
    return float(np.sqrt(np.mean(np.abs(got - want) ** 2)) / max(ref, 1e-300))


def _equal_carriers(n, fs, tunes, modes, seed, amp=2.0 ** 24):
    """make_iq's signal shapes with one amplitude for every carrier"""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / fs
    acc = np.zeros(n, dtype=np.complex128)
    for f, m in zip(tunes, modes):
        side = -1000.0 if m in (0, 2, 8) else 1000.0            # a tone inside the receiver's own sideband
        car = np.exp(1j * (2 * np.pi * (f + side) * t + rng.uniform(0, 2 * np.pi)))
        if m == 4:
            car = car * (1 + 0.5 * np.sin(2 * np.pi * 1000.0 * t))
        elif m in (5, 13):
            car = car * np.exp(1j * 3.0 * np.sin(2 * np.pi * 1000.0 * t))
        acc += amp * car
    acc += 2.0 ** 12 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return np.round(acc.real) + 1j * np.round(acc.imag)


def _run(fs, modes, tunes, x, sizes):
    eng = Engine(fs, n_receivers=len(modes), max_block=max(sizes), fp32=True)
    pr = PairRunner(eng, modes, tunes)
    pos = 0
    for n in sizes:
        pr.step(x[pos:pos + n])
        pos += n
    pr.close()
    assert pr.agc_exact                      # the AGC kernels run in double in both modes
    floor = 1e-3 * float(np.max(np.abs(x)))
    pre = [_rms_rel(lg["pg"], lg["po"], floor) for lg in pr.log]
    fin = [_rms_rel(lg["fg"], lg["fo"]) for lg in pr.log]
    return pre, fin, pr


def test_fp32_audio_equal_carriers_every_class():
    """CW / SSB / AM / FM / DGT-U receivers at 1.536 Msps (four fused half-bands + FIR98/2 in float), ragged blocks."""
    fs = 1536000
    modes = [3, 2, 4, 5, 1, 7]
    tunes = tune_offsets(len(modes), fs)
    sizes = [1 << 16, 50001, 1 << 16, 7, 1 << 16, 1 << 16]
    x = _equal_carriers(sum(sizes), fs, tunes, modes, 1005)
    pre, fin, pr = _run(fs, modes, tunes, x, sizes)
    print("fp32, equal carriers: pre-AGC relative RMS error per receiver", ["%.1e" % v for v in pre], "behind the AGC", ["%.1e" % v for v in fin])
    assert max(pre) <= 1e-5 and max(pre) > 1e-9          # within the stated tolerance -- and really computed in float
    assert np.median(fin) <= 1e-5 and max(fin) <= 5e-3   # behind the AGC: see the module docstring


def test_fp32_audio_config5_slice_dynamic_range():
    """The per-GPU share of configs[4] in miniature: 16 mixed receivers at 61.44 Msps (3 half-bands, FIR/3, 2 x FIR/5, FIR/2,
    24/25 resampler, fractional decimator), carriers spread over 60 dB as SURVEY.md 8(d) prescribes."""
    fs, nrx = 61440000, 16
    tunes, modes = tune_offsets(nrx, fs), mode_mix(nrx, "mixed")
    sizes = [1 << 20, 1 << 20, 400000]
    x = make_iq(sum(sizes), fs, tunes, modes, 1005)
    pre, fin, pr = _run(fs, modes, tunes, x, sizes)
    own = [float(np.sqrt(np.mean(np.abs(np.concatenate(lg["po"])) ** 2))) / float(np.max(np.abs(x))) for lg in pr.log]
    print("fp32, 60 dB of carriers: pre-AGC RMS error per receiver (own audio / stream peak)",
          ["%.1e (%.0e)" % (v, o) for v, o in zip(pre, own)], "behind the AGC", ["%.1e" % v for v in fin])
    assert max(pre) <= 1e-5


def test_fp32_config5_slice_at_bench_geometry():
    """The per-GPU share of configs[4] as bench.py's config5_slice leg runs it: 128 mixed receivers at 61.44 Msps in fp32 mode,
    blocks of 2^20 samples, the 2^20-point display FFT of every block beside them (the packed float front end's span partition
    depends on receiver count and block length both).  Audio to the stated tolerance, AGC bit-exact, the averaged row <= 0.01 dB."""
    fs, nrx, N, W = 61440000, 128, 1 << 20, 4096
    tunes, modes = tune_offsets(nrx, fs), mode_mix(nrx, "mixed")
    sizes = [N, N, N]
    x = make_iq(sum(sizes), fs, tunes, modes, 1006)       # a carrier for every receiver: an FM discriminator on noise alone amplifies any input difference
    eng = Engine(fs, n_receivers=nrx, max_block=N, fft_size=N, data_width=W, fp32=True)
    pr = PairRunner(eng, modes, tunes)
    og = B.OracleGraph(N, W, fs)
    _, _, bw0, off0 = rx_filter(fs, int(modes[0]))
    og.set_passband(int(tunes[0]), off0, bw0)           # the S-meter reads receiver 0's passband
    pos = 0
    for n in sizes:
        pr.step(x[pos:pos + n])
        pos += n
    og.add_frames(x)
    n_got, row, sm = eng.get_graph(1.0, 0.0)
    n_want, want, want_sm = og.release(1.0, 0.0)
    pr.close(); og.close()
    assert pr.agc_exact and len(pr.log) == nrx
    floor = 1e-3 * float(np.max(np.abs(x)))
    pre = [_rms_rel(lg["pg"], lg["po"], floor) for lg in pr.log]
    err = np.abs(row - want)
    print("fp32, 128 receivers + 2^20-point FFT: worst pre-AGC RMS error %.1e (median %.1e); row %.2e dB, S-meter %.2e dB"
          % (max(pre), float(np.median(pre)), err.max(), abs(sm - want_sm)))
    assert max(pre) <= 1e-5 and max(pre) > 1e-9
    assert n_got == n_want == len(sizes) and err.max() <= 0.01 and abs(sm - want_sm) <= 0.01


@pytest.mark.parametrize("N,W,frames", [(4096 * 4, 1024, 3), (1 << 20, 4096, 8)])
def test_fp32_spectrum(N, W, frames):
    fs = 61440000
    eng = Engine(fs, n_receivers=1, max_block=frames * N, fft_size=N, data_width=W, fp32=True)
    og = B.OracleGraph(N, W, fs)
    fi, fq, bw, off = rx_filter(fs, 3)
    tune = int(0.11 * fs)
    eng.set_mode(0, 3); eng.set_tune(0, tune); eng.set_filters(0, fi, fq, bw, off)
    og.set_passband(tune, off, bw)
    x = make_iq(frames * N, fs, [tune, -int(0.3 * fs), int(0.4 * fs)], [3, 4, 5], 1003)
    eng.process(x)
    og.add_frames(x)
    n, row, sm = eng.get_graph(1.0, 0.0)
    n2, want, want_sm = og.release(1.0, 0.0)
    assert n == n2 == frames
    err = np.abs(row - want)
    print("fp32 spectrum %d points: worst %.2e dB over the row (%.0f ... %.0f dB); S-meter %.2e dB"
          % (N, err.max(), want.min(), want.max(), abs(sm - want_sm)))
    assert err.max() <= 0.01 and abs(sm - want_sm) <= 0.01
    assert err.max() > 1e-9                               # float arithmetic did run
    eng.close(); og.close()
