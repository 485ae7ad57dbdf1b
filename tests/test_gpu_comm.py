"""The RCCL gather inside libqkengine (qk_comm_* / qk_gather_*), as far as one GPU can show it: a communicator of one rank, whose
gather is a send to itself through ncclSend / ncclRecv -- the same group of calls that N ranks issue; the gathered rows and
counts must be the engine's own audio, the gathered graph row its released row.  (The N-rank layout is exercised on CPU by
tests/test_shard_gloo.py; a node with more GPUs runs it through bench.py --gpus N.)"""
import numpy as np
import pytest

from _hip import DevBuf
from _rxsetup import configure_engine
from quisk_amd import Engine
from quisk_amd.synth import make_iq, tune_offsets, mode_mix

pytestmark = pytest.mark.gpu


def test_gather_of_one_rank_returns_the_engines_own_outputs():
    fs, nrx, blk = 1536000, 6, 65536
    tunes, modes = tune_offsets(nrx, fs), mode_mix(nrx, "mixed")
    x = make_iq(3 * blk, fs, tunes, modes, 2)
    eng = Engine(fs, n_receivers=nrx, max_block=blk, fft_size=4096, data_width=512)
    configure_engine(eng, modes, tunes)
    eng.comm_init(Engine.comm_unique_id(), 0, 1)
    for b in range(3):
        eng.process(x[b * blk:(b + 1) * blk])
    audio, counts = eng.audio_all()
    _, stride = eng.audio_device()
    d_audio = DevBuf(nbytes=16 * nrx * stride)
    d_counts = DevBuf(nbytes=4 * nrx)
    eng.gather_audio(0, d_audio.ptr, d_counts.ptr)
    got_c = d_counts.to_host(np.int32)
    got = d_audio.to_host(np.complex128).reshape(nrx, stride)
    assert np.array_equal(got_c, counts) and int(counts.min()) > 0
    for r in range(nrx):
        assert np.array_equal(got[r, :counts[r]].view(np.float64), audio[r, :counts[r]].view(np.float64))
    cnt, row, _ = eng.get_graph(1.0, 0.0)
    assert cnt > 0
    d_row = DevBuf(nbytes=8 * 512)
    eng.gather_graph(0, d_row.ptr)
    assert np.array_equal(d_row.to_host(np.float64), row)
    eng.comm_destroy()
    eng.close()
