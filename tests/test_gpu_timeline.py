"""QK_TIMELINE (a measuring aid of the engine: eight stream marks per block, written as csv when the engine is destroyed):
the marks of a block are in pipeline order and the audio is what it is without them."""
import csv
import os

import numpy as np
import pytest

from _rxsetup import configure_engine
from quisk_amd import Engine
from quisk_amd.synth import make_iq, tune_offsets

pytestmark = pytest.mark.gpu


def _run(fs, nrx, block, nblk, x, modes, tunes):
    eng = Engine(fs, n_receivers=nrx, max_block=block)
    configure_engine(eng, modes, tunes)
    for b in range(nblk):
        eng.process(x[b * block:(b + 1) * block])
    audio, counts = eng.audio_all()
    audio, counts = audio.copy(), counts.copy()
    eng.close()
    return audio, counts


def test_stream_marks_are_in_pipeline_order(tmp_path):
    fs, nrx, block, nblk = 1536000, 4, 1 << 16, 6
    tunes, modes = tune_offsets(nrx, fs), [3, 2, 4, 5]
    x = make_iq(block * nblk, fs, tunes, modes, 77)
    plain = _run(fs, nrx, block, nblk, x, modes, tunes)
    path = tmp_path / "marks.csv"
    os.environ.update(QK_TIMELINE=str(path), QK_TIMELINE_FIRST="1", QK_TIMELINE_BLOCKS="4")
    try:
        marked = _run(fs, nrx, block, nblk, x, modes, tunes)
    finally:
        for k in ("QK_TIMELINE", "QK_TIMELINE_FIRST", "QK_TIMELINE_BLOCKS"):
            os.environ.pop(k, None)
    assert np.array_equal(plain[1], marked[1]) and np.array_equal(plain[0].view(np.float64), marked[0].view(np.float64))
    rows = list(csv.DictReader(open(path)))
    assert [int(r["block"]) for r in rows] == [1, 2, 3, 4]
    for r in rows:
        t = {k: float(v) for k, v in r.items() if k != "block"}
        assert min(t.values()) >= 0.0
        assert t["fe_start"] <= t["fe_end"] <= t["chain_start"] <= t["chain_end"] <= t["agc_start"] <= t["gain_end"] <= t["agc_end"], t
        assert t["chain_end"] <= t["block_end"], t
    starts = [float(r["fe_start"]) for r in rows]
    assert starts == sorted(starts)
