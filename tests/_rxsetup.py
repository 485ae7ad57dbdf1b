"""Shared set-up for the parity tests: configure an oracle receiver and an engine receiver identically,
and compare them block by block.

Parity contract ("Parity"):
  1. everything in front of the AGC (mixer, decimators, band-pass, detector, audio interpolators): the
     engine's AGC input matches the oracle's within TOL = 1e-9 of the block peak (BASELINE.json north_star);
  2. the AGC kernels are BIT-EXACT against the oracle's process_agc restatement when both are fed the same
     samples (the engine's own AGC input): stronger than 1e-9;
  3. end to end through the AGC the deviation is MEASURED, not bounded by a loose constant.  process_agc is
     ill-conditioned by construction: it ends a gain ramp on `gain <= target_gain` after subtracting
     (gain-target)/buf_size exactly buf_size times (quisk.c:2079, 2120-2121), an equality decided by the last bit, so
     inputs that differ by one ulp can end the ramp one sample apart and shift the following release window
     (tests/test_oracle_rx.py::test_agc_is_ill_conditioned shows it on the oracle alone).  agc_deviation_report()
     runs the oracle's state machine on both AGC inputs (the oracle's own and the engine's; the second run IS the
     engine's AGC, by 2.) and asserts:
       a. every output sample whose end-to-end error exceeds 1e-9 lies inside an EPISODE, an interval in which the two
          state machines are in different discrete states (is_clipping / index_start) or their gains differ by more than
          1e-10 relative;
       b. every episode starts at a discrete decision that went the other way (a ramp end or a clip start), never by the
          gains drifting apart on their own;
       c. inside an episode the error of every sample is explained by the gain difference: err <= TOL + |dg| / g;
       d. outside the episodes (before, between and after them: the streams RE-CONVERGE when a later ramp ends in both
          machines) the error is <= 1e-9.
     The report carries the number of samples above 1e-9, the worst of them, the episodes and their lengths; smoke() and
     the tests print it.
"""
import ctypes as C
import os

import numpy as np

import _bind as B
from quisk_amd import rxfilters
from quisk_amd.engine import TAP_PLAY

TOL = 1e-9
GAIN_TOL = 1e-10        # two gains count as equal below this relative difference


rx_filter = rxfilters.rx_filter     # (fi, fq, bw, start_offset) as quisk.py OnBtnFilter builds them (quisk.py:4837-4868)


def make_oracle_rx(sample_rate, mode, tune, playback_rate=48000, agc_gain=80.0, bw=None):
    fi, fq, bw, _ = rx_filter(sample_rate, mode, bw)
    return B.OracleRx(sample_rate, playback_rate, mode=mode, tune=int(tune), fi=fi, fq=fq, bandwidth=bw, agc_gain=agc_gain)


def configure_engine(eng, modes, tunes, agc_gain=80.0, bws=None):
    for r, (m, t) in enumerate(zip(modes, tunes)):
        fi, fq, bw, off = rx_filter(eng.sample_rate, int(m), bws[r] if bws else None)
        eng.set_mode(r, int(m))
        eng.set_tune(r, int(t))
        eng.set_filters(r, fi, fq, bw, off)
        eng.set_agc(r, agc_gain)


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    scale = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0))


class AgcTwin:
    """The oracle's process_agc run on the ENGINE's own AGC input, to check the AGC kernels bit for bit."""

    def __init__(self, playback_rate, mode, agc_gain=80.0):
        self.L = B.oracle()
        self.st = B.OAgc()
        self.L.qko_agc_init(C.byref(self.st), 0.7, 0)
        self.rate, self.gain, self.fm = playback_rate, agc_gain, int(mode in (5, 13))
        self.cpx = int(mode == 9)       # DGT-IQ: stereo, process_agc with is_cpx (quisk.c:2556-2558)

    def run(self, pre):
        s = np.array(pre, dtype=np.complex128) if self.cpx else (pre + 1j * pre).astype(np.complex128)
        if len(s):
            self.L.qko_agc_process(C.byref(self.st), s, len(s), self.cpx, self.rate, self.gain, 1.0, self.fm)
        return s


def agc_deviation_report(blocks_o, blocks_g, fin_o, fin_g, playback_rate, mode, agc_gain=80.0, release_time=1.0):
    """blocks_o / blocks_g: the AGC inputs of the oracle and of the engine, one array per process_agc call (same
    lengths); fin_o / fin_g: the concatenated outputs.  Returns the report dict described in this module's docstring
    (asserting a-d)."""
    L = B.oracle()
    fm, cpx = int(mode in (5, 13)), int(mode == 9)
    traces = []
    for blocks in (blocks_o, blocks_g):
        st = B.OAgc()
        L.qko_agc_init(C.byref(st), 0.7, 0)
        gs, cs, iss = [], [], []
        for b in blocks:
            n = len(b)
            if not n:
                continue
            x = np.array(b, dtype=np.complex128) if cpx else (np.asarray(b).real + 1j * np.asarray(b).real).astype(np.complex128)
            g = np.zeros(n); c = np.zeros(n, np.int32); i0 = np.zeros(n, np.int32)
            L.qko_agc_trace(C.byref(st), x, n, cpx, playback_rate, agc_gain, release_time, fm, g, c, i0)
            gs.append(g); cs.append(c); iss.append(i0)
        L.qko_agc_free(C.byref(st))
        traces.append((np.concatenate(gs), np.concatenate(cs), np.concatenate(iss)))
    (ga, ca, ia), (gb, cb, ib) = traces
    n = len(ga)
    assert len(fin_o) == len(fin_g) == n
    scale = max(float(np.max(np.abs(fin_o))), 1.0)
    err = np.abs(np.asarray(fin_g) - np.asarray(fin_o)) / scale
    # state AFTER sample i decides output i + 1: shift by one so that index i speaks about output i
    desync = np.concatenate([[False], (ca != cb) | (ia != ib)])[:n]
    grel = np.concatenate([[0.0], np.abs(ga - gb) / np.maximum(np.abs(ga), 1e-300)])[:n]
    inside = desync | (grel > GAIN_TOL)
    bad = err > TOL
    # episodes: maximal runs of `inside`
    edges = np.flatnonzero(np.diff(np.concatenate([[0], inside.astype(np.int8), [0]])))
    starts, ends = edges[0::2], edges[1::2]
    rep = {"samples": int(n), "above_1e-9": int(bad.sum()), "worst": float(err[bad].max()) if bad.any() else float(err.max() if n else 0.0),
           "worst_outside": float(err[~inside].max()) if (~inside).any() else 0.0,
           "episodes": int(len(starts)), "episode_samples": int(inside.sum()),
           "longest_episode": int((ends - starts).max()) if len(starts) else 0,
           "closed_episodes": int(np.sum(ends < n)), "reconverged": bool(len(starts) == 0 or ends[-1] < n)}
    assert not np.any(bad & ~inside), "end-to-end error above 1e-9 outside an AGC divergence episode: %r" % rep       # a, d
    for s0 in starts:
        assert desync[s0], "an episode began without a discrete decision (sample %d)" % s0                            # b
    # c: |d_g g_g - d_o g_o| <= |d| |g_g - g_o| + TOL.  Where the limiter acted (|out| > CLIP32 -> out / |out|, unit
    # magnitude, quisk.c:2063-2066) the product is not the output: those samples are left out of this check.
    mo, mg = np.abs(np.asarray(fin_o)) if cpx else np.abs(np.asarray(fin_o).real), np.abs(np.asarray(fin_g)) if cpx else np.abs(np.asarray(fin_g).real)
    lim = (np.abs(mo - 1.0) < 1e-6) | (np.abs(mg - 1.0) < 1e-6)
    expl = TOL + grel * (np.abs(np.asarray(fin_o)) / scale) * (1 + 1e-6)
    chk = inside & ~lim
    rep["limited"] = int(lim.sum())
    assert np.all(err[chk] <= expl[chk]), "an error inside an episode exceeds what the gain difference explains: %r" % rep
    return rep


class PairRunner:
    """Engine and per-receiver oracles stepped together; accumulates the three parity measures."""

    def __init__(self, eng, modes, tunes, agc_gain=80.0, pre_tol=None, bws=None):
        self.eng = eng
        self.modes = [int(m) for m in modes]
        configure_engine(eng, modes, tunes, agc_gain, bws)
        self.oras = [make_oracle_rx(eng.sample_rate, int(m), int(t), eng.playback_rate, agc_gain, bws[r] if bws else None)
                     for r, (m, t) in enumerate(zip(modes, tunes))]
        self.twins = [AgcTwin(eng.playback_rate, int(m), agc_gain) for m in modes]
        self.pre_err = 0.0
        self.agc_exact = True
        self.final_err = 0.0
        self.samples = 0
        self.peak = [0.0] * len(self.oras)
        self.pre_tol = pre_tol
        self.stereo_seen = False
        self.agc_gain = agc_gain
        self.pool = None
        self.log = [dict(po=[], pg=[], fo=[], fg=[]) for _ in self.oras]     # per receiver: AGC inputs and outputs, block by block
        self.reports = []

    def step(self, blk):
        self.eng.process(blk)
        if len(self.oras) > 8:      # wide configurations: the oracle's receivers run on the host's cores side by side
            if self.pool is None:   # (ctypes releases the GIL around the C call)
                from concurrent.futures import ThreadPoolExecutor
                self.pool = ThreadPoolExecutor(min(16, os.cpu_count() or 1))
            parts = list(self.pool.map(lambda o: o.process_parts(blk), self.oras))
            fin_all, counts = self.eng.audio_all()
        else:
            parts = fin_all = None
        for r, o in enumerate(self.oras):
            pre_o, fin_o = parts[r] if parts else o.process_parts(blk)
            fin_g = fin_all[r, :counts[r]].copy() if parts else self.eng.audio(r)
            if self.modes[r] == 9:      # DGT-IQ: the play stream is complex
                pg = self.eng.tap(TAP_PLAY, r, planes=2)
                pre_g = pg[0] + 1j * pg[1]
            else:
                pre_g = self.eng.tap(TAP_PLAY, r, planes=1)[0]
            assert len(fin_g) == len(fin_o) == len(pre_g), (len(fin_g), len(fin_o), len(pre_g))
            if not len(fin_o):
                continue
            lg = self.log[r]
            lg["po"].append(pre_o if self.modes[r] == 9 else pre_o.real.copy()); lg["pg"].append(pre_g)
            lg["fo"].append(fin_o); lg["fg"].append(fin_g)
            self.samples += len(fin_o)
            # errors are taken relative to the receiver's running audio peak, so that an almost silent
            # start-up block is not judged against its own tiny amplitude
            ref = pre_o if self.modes[r] == 9 else pre_o.real
            self.peak[r] = max(self.peak[r], float(np.max(np.abs(ref))))
            scale = self.peak[r] if self.peak[r] > 0 else 1.0
            self.pre_err = max(self.pre_err, float(np.max(np.abs(pre_g - ref))) / scale)
            twin = self.twins[r].run(pre_g)
            if self.modes[r] == 9 and np.any(fin_g.real != fin_g.imag):
                self.stereo_seen = True
            if not np.array_equal(twin.view(np.float64), fin_g.view(np.float64)):
                self.agc_exact = False
            fscale = max(float(np.max(np.abs(fin_o))), 1.0)
            self.final_err = max(self.final_err, float(np.max(np.abs(fin_g - fin_o))) / fscale)

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
            self.pool = None
        for o in self.oras:
            o.close()
        self.eng.close()

    def check(self, pre_tol=TOL):
        """1. pre-AGC <= 1e-9; 2. AGC bit-exact on identical input; 3. the measured end-to-end statement (module docstring)."""
        assert self.samples > 0
        assert self.pre_err < (self.pre_tol or pre_tol), "pre-AGC error %g" % self.pre_err
        assert self.agc_exact, "AGC kernels are not bit-exact against the oracle on identical input"
        self.reports = []
        for r, lg in enumerate(self.log):
            if not lg["fo"]:
                continue
            self.reports.append(agc_deviation_report(lg["po"], lg["pg"], np.concatenate(lg["fo"]), np.concatenate(lg["fg"]),
                                                     self.eng.playback_rate, self.modes[r], self.agc_gain))
        return self.summary()

    def summary(self):
        tot = {"receivers": len(self.reports), "samples": sum(r["samples"] for r in self.reports),
               "above_1e-9": sum(r["above_1e-9"] for r in self.reports),
               "worst": max([r["worst"] for r in self.reports] or [0.0]),
               "worst_outside_episodes": max([r["worst_outside"] for r in self.reports] or [0.0]),
               "episodes": sum(r["episodes"] for r in self.reports),
               "closed_episodes": sum(r["closed_episodes"] for r in self.reports),
               "longest_episode": max([r["longest_episode"] for r in self.reports] or [0])}
        return tot
